"""Which GEMM form wins at which batch size?  Times the decoder's shapes at M = 64 ... 8192 rows in the form the
environment selects (run with DIDI_SMALL_MAX_M / DIDI_BIG_MIN_M / DIDI_GEMM_BIG to compare); `small` uses didi_gemm_small."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from didi_amd import ops  # noqa: E402
from didi_amd._lib import lib  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


g = torch.Generator(device="cuda").manual_seed(0)
shapes = [("qkv", 2304, 768), ("proj", 768, 768), ("ffn_up", 3072, 768), ("ffn_down", 768, 3072)]
print("form:", sys.argv[1], {k: v for k, v in os.environ.items() if k.startswith("DIDI_")})
for M in (64, 512, 1024, 2048, 4096, 8192):
    row = []
    for name, N, K in shapes:
        a = torch.randn(M, K, device="cuda", generator=g)
        w = torch.randn(N, K, device="cuda", generator=g) * 0.02
        a_hi, _ = ops.split(a, need_lo=False, dtype=torch.float16)
        w_hi, w_lo = ops.split(w, dtype=torch.float16)
        small = () if sys.argv[1] == "small" else None
        if small is not None and lib.didi_gemm_small_workspace_bytes(M, N, K, ops.PRECISIONS["fp16"]) == 0:
            row.append(f"{name} n/a")
            continue
        us = timeit(lambda: ops.gemm(a_hi, None, w_hi, w_lo, precision="fp16", out_lo=False, small=small))
        row.append(f"{name} {us:6.1f}")
    print(f"M={M:5d}  " + "  ".join(row), flush=True)
