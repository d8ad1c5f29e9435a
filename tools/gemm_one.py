#!/usr/bin/env python3
"""Launch ONE decoder GEMM shape a few times (for rocprofv3 --pmc passes).

    python tools/gemm_one.py --N 768 --K 3072 --precision bf16x3 [--M 16384] [--iters 5] [--epi planes|f32res|gelu]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from didi_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=16384)
ap.add_argument("--N", type=int, default=768)
ap.add_argument("--K", type=int, default=3072)
ap.add_argument("--precision", default="fp16")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--epi", default="planes")
a = ap.parse_args()
g = torch.Generator(device="cuda").manual_seed(0)
pd = ops.plane_dtype(a.precision)
x_hi, x_lo = ops.split(torch.randn(a.M, a.K, device="cuda", generator=g), dtype=pd)
w_hi, w_lo = ops.split(torch.randn(a.N, a.K, device="cuda", generator=g) * 0.02, dtype=pd)
bias = torch.randn(a.N, device="cuda", generator=g)
r = ops.split(torch.randn(a.M, a.N, device="cuda", generator=g), dtype=pd) if a.epi == "f32res" else None
for _ in range(a.iters):
    ops.gemm(x_hi, x_lo, w_hi, w_lo, bias=bias, act="gelu" if a.epi == "gelu" else None, precision=a.precision,
             residual_planes=r, out_f32=bool(r), out_planes=not r, out_lo=a.precision == "bf16x3")
torch.cuda.synchronize()
print("done")
