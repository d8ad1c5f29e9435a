#!/usr/bin/env python3
"""What the fused projection + LayerNorm launch pays for its residual / output planes: the same launch (M = 16384, N = 768) with the
remainder planes full (16 bit), as bytes (lo8), or absent, on the read side, the write side, or both.  usage: [--K 768]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from didi_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--K", type=int, default=768)
ap.add_argument("--precision", default="fp16")
a = ap.parse_args()
M, N, K = 16384, 768, a.K
g = torch.Generator(device="cuda").manual_seed(0)
L = 6  # rotate over several operand sets, as the layers of a call do
xs = [ops.split(torch.randn(M, K, device="cuda", generator=g), dtype=torch.float16) for _ in range(L)]
ws = [ops.split(torch.randn(N, K, device="cuda", generator=g) * 0.03, dtype=torch.float16) for _ in range(L)]
bias = torch.randn(N, device="cuda", generator=g)
gam, bet = torch.rand(N, device="cuda", generator=g) + 0.5, torch.randn(N, device="cuda", generator=g)
h_hi, h_lo = ops.split(torch.randn(M, N, device="cuda", generator=g), dtype=torch.float16)
h_b = torch.zeros((M, N), dtype=torch.uint8, device="cuda")


def run(r_lo, y_lo, lo_bytes, n=600):
    def one(i):
        ops.gemm_ln(xs[i % L][0], xs[i % L][1], ws[i % L][0], ws[i % L][1], bias, (h_hi, r_lo), gam, bet, 1e-12, precision=a.precision,
                    out=(h_hi, y_lo), lo_bytes=lo_bytes)
    for i in range(50):
        one(i)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for i in range(n):
        one(i)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


for rep in range(2):
    print(f"K={K} {a.precision}: read16+write16 {run(h_lo, h_lo, 0):.1f} us | read8+write8 {run(h_b, h_b, 3):.1f} | read16 only {run(h_lo, None, 0):.1f} | "
          f"write16 only {run(None, h_lo, 0):.1f} | read8 only {run(h_b, None, 1):.1f} | write8 only {run(None, h_b, 2):.1f} | hi only {run(None, None, 0):.1f}", flush=True)
