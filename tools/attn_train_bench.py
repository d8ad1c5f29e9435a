#!/usr/bin/env python3
"""Training attention forward / backward at the bench shape (256 dialogs x 12 heads x 64 x 64 x 64): microseconds per launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from didi_amd import ops  # noqa: E402

B, heads, S, D = 256, 12, 64, 64
H = heads * D
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, S, 3 * H, device="cuda", generator=g)
q, k, v = qkv[:, :, :H], qkv[:, :, H:2 * H], qkv[:, :, 2 * H:]
dO = torch.randn(B, S, H, device="cuda", generator=g)
out = torch.zeros_like(qkv)
outs = (out[:, :, :H], out[:, :, H:2 * H], out[:, :, 2 * H:])


def timeit(fn, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for p in (0.0, 0.1):
    for causal in (True, False):
        kw = dict(causal=causal, drop_p=p, seed=7, offset=0)
        f = timeit(lambda: ops.attention_train(q, k, v, heads, **kw))
        b = timeit(lambda: ops.attention_backward(q, k, v, dO, heads, out=outs, **kw))
        print(f"drop {p} causal {causal}: forward {f:.1f} us, backward {b:.1f} us", flush=True)
