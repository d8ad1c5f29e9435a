#!/usr/bin/env python3
"""What each GEMM epilogue variant costs on the decoder's shapes (fp16, M = 16384): same main loop, different stores."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from didi_amd import ops  # noqa: E402

M = 16384
gen = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=gen)  # noqa: E731


def timeit(call, iters=30):
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    a_hi, _ = ops.split(rnd(M, K), dtype=torch.float16)
    w_hi, w_lo = ops.split(rnd(N, K) * 0.02, dtype=torch.float16)
    bias = rnd(N)
    r = ops.split(rnd(M, N), dtype=torch.float16)
    cases = {
        "f32 out": dict(out_f32=True, out_planes=False),
        "f32 out + bias": dict(bias=bias, out_f32=True, out_planes=False),
        "hi plane + bias": dict(bias=bias, out_lo=False),
        "hi+lo planes + bias": dict(bias=bias),
        "hi plane + bias + gelu": dict(bias=bias, act="gelu", out_lo=False),
        "f32 out + bias + residual planes": dict(bias=bias, residual_planes=r, out_f32=True, out_planes=False),
    }
    for name, kw in cases.items():
        us = timeit(lambda: ops.gemm(a_hi, None, w_hi, w_lo, precision="fp16", **kw))
        print(f"N={N:5d} K={K:5d} {name:34s} {us:7.1f} us")
