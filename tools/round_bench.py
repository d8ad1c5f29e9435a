"""Rounding GEMM + argmax (didi_round_argmax) at cfg2 and config-4 vocabulary sizes, M = 16384 rows."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from didi_amd import ops  # noqa: E402

M, K = 16384, 768
g = torch.Generator(device="cuda").manual_seed(0)
for V in (8008, 8064, 50000):
    for prec in ("fp16", "bf16"):
        pd = ops.plane_dtype(prec)
        x_hi, _ = ops.split(torch.randn(M, K, device="cuda", generator=g), need_lo=False, dtype=pd)
        w_hi, w_lo = ops.split(torch.randn(V, K, device="cuda", generator=g) * 0.02, dtype=pd)
        for _ in range(3):
            ops.round_argmax(x_hi, None, w_hi, w_lo, prec)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.round_argmax(x_hi, None, w_hi, w_lo, prec)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 10
        print(f"V={V:6d} {prec:5s} {us:8.1f} us  {2.0 * M * V * K / us / 1e6:7.1f} TF (GEMM + reduce)", flush=True)
