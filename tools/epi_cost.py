"""What does each epilogue ingredient of the 256x192 kernel cost?  qkv / ffn_up shapes, fp16, M = 16384."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from didi_amd import ops  # noqa: E402

M = 16384
g = torch.Generator(device="cuda").manual_seed(0)


def t(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for name, N, K in (("qkv", 2304, 768), ("ffn_up", 3072, 768), ("proj", 768, 768)):
    a_hi, _ = ops.split(torch.randn(M, K, device="cuda", generator=g), need_lo=False, dtype=torch.float16)
    w_hi, w_lo = ops.split(torch.randn(N, K, device="cuda", generator=g) * 0.02, dtype=torch.float16)
    bias = torch.randn(N, device="cuda", generator=g)
    rows = []
    for label, kw in (("hi plane, no bias", dict(out_lo=False)), ("hi plane + bias", dict(bias=bias, out_lo=False)),
                      ("hi plane + bias + gelu", dict(bias=bias, act="gelu", out_lo=False)),
                      ("f32 out, no bias", dict(out_f32=True, out_planes=False)), ("hi + lo planes + bias", dict(bias=bias))):
        rows.append(f"{label}: {t(lambda: ops.gemm(a_hi, None, w_hi, w_lo, precision='fp16', **kw)):6.1f}")
    print(f"{name:7s} " + " | ".join(rows), flush=True)
