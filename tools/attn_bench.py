#!/usr/bin/env python3
"""Time didi_attention on the decoder's shapes (B=256, 12 heads, S=C=64, D=64): causal self and masked cross."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from didi_amd import ops  # noqa: E402

B, heads, S, D = 256, 12, 64, 64
H = heads * D
gen = torch.Generator(device="cuda").manual_seed(0)
for prec in ("fp16", "bf16x3"):
    pd = ops.plane_dtype(prec)
    x3 = prec == "bf16x3"
    qkv_hi, qkv_lo = ops.split(torch.randn(B, S, 3 * H, device="cuda", generator=gen), need_lo=x3, dtype=pd)
    kv_hi, kv_lo = ops.split(torch.randn(B, S, 2 * H, device="cuda", generator=gen), need_lo=x3, dtype=pd)
    q_hi, q_lo = ops.split(torch.randn(B, S, H, device="cuda", generator=gen), need_lo=x3, dtype=pd)
    # CommonSense-shaped context lengths: clip(round(N(24, 12^2)), 2, S)
    lens = torch.clamp(torch.round(24 + 12 * torch.randn(B, 1, device="cuda", generator=gen)), 2, S)
    mask = (torch.arange(S, device="cuda")[None] < lens).int()
    counts = ops.key_counts(mask)
    v = lambda t, a, b: None if t is None else t[:, :, a:b]  # noqa: E731
    cases = {
        "causal self": lambda: ops.attention((v(qkv_hi, 0, H), v(qkv_lo, 0, H)), (v(qkv_hi, H, 2 * H), v(qkv_lo, H, 2 * H)),
                                             (v(qkv_hi, 2 * H, 3 * H), v(qkv_lo, 2 * H, 3 * H)), heads, causal=True, precision=prec),
        "masked cross": lambda: ops.attention((q_hi, q_lo), (v(kv_hi, 0, H), v(kv_lo, 0, H)), (v(kv_hi, H, 2 * H), v(kv_lo, H, 2 * H)),
                                              heads, causal=False, key_mask=mask, precision=prec),
        "masked cross, key counts": lambda: ops.attention((q_hi, q_lo), (v(kv_hi, 0, H), v(kv_lo, 0, H)), (v(kv_hi, H, 2 * H), v(kv_lo, H, 2 * H)),
                                                          heads, causal=False, key_mask=mask, precision=prec, key_count=counts),
    }
    for name, call in cases.items():
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            call()
        e1.record()
        torch.cuda.synchronize()
        print(f"{prec:7s} {name:25s} {e0.elapsed_time(e1) * 1e3 / 50:7.1f} us")
