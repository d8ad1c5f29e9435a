#!/usr/bin/env python3
"""Where a workgroup of the fused cross-attention launch (gemm_pp_kernel<*, ATTN = 2>) spends its time: s_memtime stamps.
Needs the DIAGNOSTIC build:  make -C didi_amd/csrc EXTRA=-DDIDI_STAMPS BUILD=build_stamps LIB=../../tools/ab/libdidi_hip_stamps.so
    DIDI_HIP_LIB=tools/ab/libdidi_hip_stamps.so python tools/cross_stamps.py [--precision fp16] [--full]
Stamps (waves 0 and 4): 0 start, 1 first K-tile landed, 2 K-loop done, 3 queries converted, 8 + ph after phase ph's barrier,
4 / 5 end of phase 0's / 5's arithmetic, 6 stores issued, 7 drained."""
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
ap.add_argument("--full", action="store_true")
a = ap.parse_args()
B, S, Cn, heads, H, L = 256, 64, 64, 12, 768, 12
M = B * S
set_stamps = _lib.lib.didi_debug_set_stamps
set_stamps.restype, set_stamps.argtypes = None, [C.c_void_p]
g = torch.Generator(device="cuda").manual_seed(0)
x_hi, x_lo = ops.split(torch.randn(M, H, device="cuda", generator=g), dtype=torch.float16)
w_hi, w_lo = ops.split(torch.randn(H, H, device="cuda", generator=g) * 0.05, dtype=torch.float16)
bias = torch.randn(H, device="cuda", generator=g)
kv = [torch.randn(B * Cn, 2 * H, device="cuda", generator=g).to(torch.float16) for _ in range(L)]
_, mask = synthetic_batch(B, Cn, 8008, seed=0xBADB01)
mask = (torch.ones_like(mask) if a.full else mask).to(torch.int32).cuda()
counts = torch.empty(B, dtype=torch.int32, device="cuda")
_lib.check(_lib.lib.didi_key_counts(mask.data_ptr(), B, Cn, counts.data_ptr(), _lib.current_stream()), "key_counts")
out = torch.empty((M, H), dtype=torch.float16, device="cuda")
gg = _lib.GemmArgs()
gg.M, gg.N, gg.K, gg.precision = M, H, H, ops.PRECISIONS[a.precision]
gg.A_hi, gg.A_lo, gg.lda = x_hi.data_ptr(), x_lo.data_ptr(), H
gg.W_hi, gg.W_lo, gg.ldw = w_hi.data_ptr(), w_lo.data_ptr(), H
gg.bias, gg.act = bias.data_ptr(), _lib.ACT_NONE
gg.C_hi, gg.ldc = out.data_ptr(), H
n_wg = (M // 256) * 4
stamps = torch.zeros(n_wg * 2 * 16, dtype=torch.int64, device="cuda")


def launch(l):
    _lib.check(_lib.lib.didi_q_cross_attention(C.byref(gg), kv[l].data_ptr(), kv[l].data_ptr() + 2 * H, 2 * H, mask.data_ptr(), counts.data_ptr(),
                                                heads, S, Cn, 0.0, _lib.current_stream()), "fused")


set_stamps(None)
for i in range(2000):
    launch(i % L)
set_stamps(stamps.data_ptr())
launch(0)
set_stamps(None)
torch.cuda.synchronize()
s = stamps.view(n_wg, 2, 16).cpu().double()
clk = ((s[:, :, 7] - s[:, :, 0]) / ((s[:, :, 15] - s[:, :, 14]) / 100.0)).median().item()  # cycles per microsecond
print(f"in-kernel clock ~{clk:.0f} MHz; medians over {n_wg} workgroups, microseconds since the kernel's start (group X = wave 0, Y = wave 4)")
names = [(0, "start"), (1, "first K-tile"), (2, "K-loop done"), (3, "queries converted"), (8, "phase 0 barrier"), (4, "phase 0 arithmetic done"),
         (9, "phase 1 barrier"), (10, "phase 2 barrier"), (11, "phase 3 barrier"), (12, "phase 4 barrier"), (13, "phase 5 barrier"),
         (5, "phase 5 arithmetic done"), (6, "stores issued"), (7, "drained")]
for grp in range(2):
    t0 = s[:, grp, 0]
    print("group", "XY"[grp], "  ".join(f"{n} {((s[:, grp, i] - t0) / clk).median().item():.2f}" for i, n in names))
