#!/usr/bin/env python3
"""Cross-attention block in isolation: didi_gemm (query projection) + didi_attention over the hoisted K / V against the one-launch
didi_q_cross_attention, at the benchmark shape (256 dialogs x 64 positions, 12 heads, contexts right-padded to 64 with the
synthetic length distribution of bench.py).  12 different K / V caches are walked in turn (as the 12 layers do: the cache is not
L2-resident).  usage: python tools/cross_bench.py [--precision fp16|fp16x2] [--full]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from didi_amd import _lib, ops  # noqa: E402
from didi_amd.synthetic import synthetic_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="fp16")
ap.add_argument("--B", type=int, default=256)
ap.add_argument("--full", action="store_true", help="contexts of 64 visible keys")
a = ap.parse_args()
B, S, Cn, heads, H, L = a.B, 64, 64, 12, 768, 12
M = B * S
g = torch.Generator(device="cuda").manual_seed(0)
x_hi, x_lo = ops.split(torch.randn(M, H, device="cuda", generator=g), dtype=torch.float16)
w_hi, w_lo = ops.split(torch.randn(H, H, device="cuda", generator=g) * 0.05, dtype=torch.float16)
bias = torch.randn(H, device="cuda", generator=g)
kv = [torch.randn(B * Cn, 2 * H, device="cuda", generator=g).to(torch.float16) for _ in range(L)]
_, mask = synthetic_batch(B, Cn, 8008, seed=0xBADB01)
mask = torch.ones_like(mask) if a.full else mask
mask = mask.to(torch.int32).cuda()
counts = torch.empty(B, dtype=torch.int32, device="cuda")
_lib.check(_lib.lib.didi_key_counts(mask.data_ptr(), B, Cn, counts.data_ptr(), _lib.current_stream()), "key_counts")
q = torch.empty((M, H), dtype=torch.float16, device="cuda")
o1, o2 = torch.empty_like(q), torch.empty_like(q)
gg = _lib.GemmArgs()
gg.M, gg.N, gg.K, gg.precision = M, H, H, ops.PRECISIONS[a.precision]
gg.A_hi, gg.A_lo, gg.lda = x_hi.data_ptr(), x_lo.data_ptr(), H
gg.W_hi, gg.W_lo, gg.ldw = w_hi.data_ptr(), w_lo.data_ptr(), H
gg.bias, gg.act = bias.data_ptr(), _lib.ACT_NONE
at = _lib.AttnArgs()
at.B, at.heads, at.Sq, at.Sk, at.D = B, heads, S, Cn, 64
at.precision, at.causal = ops.PRECISIONS["fp16"], 0
at.Q_hi, at.ldq, at.ldk, at.ldv = q.data_ptr(), H, 2 * H, 2 * H
at.key_mask, at.key_count = mask.data_ptr(), counts.data_ptr()
at.O_hi, at.ldo = o1.data_ptr(), H
st = _lib.current_stream()


def two(l):
    gg.C_hi, gg.ldc = q.data_ptr(), H
    _lib.check(_lib.lib.didi_gemm(C.byref(gg), st), "gemm")
    at.K_hi, at.V_hi = kv[l].data_ptr(), kv[l].data_ptr() + 2 * H
    _lib.check(_lib.lib.didi_attention(C.byref(at), st), "attention")


def gemm_only(l):
    gg.C_hi, gg.ldc = q.data_ptr(), H
    _lib.check(_lib.lib.didi_gemm(C.byref(gg), st), "gemm")


def one(l):
    gg.C_hi, gg.ldc = o2.data_ptr(), H
    _lib.check(_lib.lib.didi_q_cross_attention(C.byref(gg), kv[l].data_ptr(), kv[l].data_ptr() + 2 * H, 2 * H, mask.data_ptr(), counts.data_ptr(),
                                                heads, S, Cn, 0.0, st), "fused")


def timeit(fn, n=600):
    for i in range(60):
        fn(i % L)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for i in range(n):
        fn(i % L)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


two(0)
one(0)
torch.cuda.synchronize()
print("bit-identical:", torch.equal(o1, o2))
for rep in range(3):
    print(f"{a.precision} B={B} {'full' if a.full else 'synthetic lengths'}: gemm {timeit(gemm_only):.1f} us, gemm + attention {timeit(two):.1f} us, one launch {timeit(one):.1f} us", flush=True)
