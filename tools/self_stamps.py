#!/usr/bin/env python3
"""Stamps of the fused QKV projection + causal self-attention launch (gemm_pp_kernel<*, ATTN = 1>); diagnostic build, see tools/ln_stamps.py.
Stamps: 0 start, 1 first K-tile, 2 K-loop done, 3 Q/K/V images in LDS, 6 outputs issued, 7 drained."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from didi_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="fp16")
a = ap.parse_args()
B, S, heads, H = 256, 64, 12, 768
M = B * S
set_stamps = _lib.lib.didi_debug_set_stamps
set_stamps.restype, set_stamps.argtypes = None, [C.c_void_p]
g = torch.Generator(device="cuda").manual_seed(0)
x_hi, x_lo = ops.split(torch.randn(M, H, device="cuda", generator=g), dtype=torch.float16)
w_hi, w_lo = ops.split(torch.randn(3 * H, H, device="cuda", generator=g) * 0.05, dtype=torch.float16)
bias = torch.randn(3 * H, device="cuda", generator=g)
out = torch.empty((M, H), dtype=torch.float16, device="cuda")
gg = _lib.GemmArgs()
gg.M, gg.N, gg.K, gg.precision = M, 3 * H, H, ops.PRECISIONS[a.precision]
gg.A_hi, gg.A_lo, gg.lda = x_hi.data_ptr(), x_lo.data_ptr(), H
gg.W_hi, gg.W_lo, gg.ldw = w_hi.data_ptr(), w_lo.data_ptr(), H
gg.bias, gg.act = bias.data_ptr(), _lib.ACT_NONE
gg.C_hi, gg.ldc = out.data_ptr(), H
n_wg = (M // 256) * heads
stamps = torch.zeros(n_wg * 2 * 16, dtype=torch.int64, device="cuda")


def launch():
    _lib.check(_lib.lib.didi_qkv_self_attention(C.byref(gg), heads, S, 0.0, _lib.current_stream()), "fused")


set_stamps(None)
for i in range(1500):
    launch()
set_stamps(stamps.data_ptr())
launch()
set_stamps(None)
torch.cuda.synchronize()
s = stamps.view(n_wg, 2, 16).cpu().double()
clk = ((s[:, :, 7] - s[:, :, 0]) / ((s[:, :, 15] - s[:, :, 14]) / 100.0)).median().item()
t00 = s[:, :, 14].min().item()
print(f"in-kernel clock ~{clk:.0f} MHz; {n_wg} workgroups; launch span (first stamp of the first workgroup -> last drained) {(s[:, :, 15].max().item() - t00) / 100:.2f} us")
names = [(0, "start"), (1, "first K-tile"), (2, "K-loop done"), (3, "Q/K/V in LDS"), (6, "outputs issued"), (7, "drained")]
for grp in range(2):
    t0 = s[:, grp, 0]
    print("group", "XY"[grp], "  ".join(f"{n} {((s[:, grp, i] - t0) / clk).median().item():.2f}" for i, n in names))
st = (s[:, 0, 14] - t00) / 100
print("workgroup start times (us), sorted deciles:", [round(x, 1) for x in torch.quantile(st, torch.linspace(0, 1, 11, dtype=torch.float64)).tolist()])
