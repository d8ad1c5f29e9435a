#!/usr/bin/env python3
"""What lies between two dependent launches of gemm_pp_kernel: s_memrealtime (100 MHz, chip-wide) at the first and last stamp
of every workgroup of two CONSECUTIVE launches of the same shape (diagnostic build, see tools/ln_stamps.py).
    DIDI_HIP_LIB=tools/ab/libdidi_hip_stamps.so python tools/launch_gap.py [--N 768] [--K 768] [--ln]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from didi_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=16384)
ap.add_argument("--N", type=int, default=768)
ap.add_argument("--K", type=int, default=768)
ap.add_argument("--ln", action="store_true")
ap.add_argument("--precision", default="fp16")
a = ap.parse_args()
set_stamps = _lib.lib.didi_debug_set_stamps
set_stamps.restype, set_stamps.argtypes = None, [C.c_void_p]
g = torch.Generator(device="cuda").manual_seed(0)
pd = ops.plane_dtype(a.precision)
x_hi, x_lo = ops.split(torch.randn(a.M, a.K, device="cuda", generator=g), dtype=pd)
w_hi, w_lo = ops.split(torch.randn(a.N, a.K, device="cuda", generator=g) * 0.02, dtype=pd)
bias = torch.randn(a.N, device="cuda", generator=g)
gamma, beta = torch.rand(a.N, device="cuda", generator=g) + 0.5, torch.randn(a.N, device="cuda", generator=g)
r = ops.split(torch.randn(a.M, a.N, device="cuda", generator=g), dtype=pd)
out = (torch.empty_like(r[0]), torch.empty_like(r[1]))
n_wg = (a.M // 256) * ((a.N + 191) // 192)
bufs = [torch.zeros(n_wg * 2 * 16, dtype=torch.int64, device="cuda") for _ in range(3)]


def launch():
    if a.ln:
        ops.gemm_ln(x_hi, x_lo, w_hi, w_lo, bias, r, gamma, beta, 1e-12, precision=a.precision, out=out)
    else:
        ops.gemm(x_hi, x_lo, w_hi, w_lo, bias=bias, precision=a.precision, out_f32=False, out_planes=True, out_lo=False)


set_stamps(None)
for _ in range(3000):
    launch()
for b in bufs:
    set_stamps(b.data_ptr())
    launch()
set_stamps(None)
torch.cuda.synchronize()
s = [b.view(n_wg, 2, 16).cpu().double() for b in bufs]
t0 = s[0][:, :, 14].min().item()
us = lambda t: (t - t0) / 100.0  # noqa: E731
for i, b in enumerate(s):
    st, en = b[:, :, 14], b[:, :, 15]
    print(f"launch {i}: first stamp of the earliest / median / latest workgroup {us(st.min().item()):7.2f} {us(st.median().item()):7.2f} {us(st.max().item()):7.2f} us;"
          f" last stamp (stores drained) earliest / median / latest {us(en.min().item()):7.2f} {us(en.median().item()):7.2f} {us(en.max().item()):7.2f} us;"
          f" median workgroup lifetime {((en - st).median().item()) / 100.0:.2f} us")
print(f"entry (the wave's first instructions) -> first stamp, median over workgroups: {((s[1][:, :, 14] - s[1][:, :, 12]).median().item()) / 100.0:.2f} us")
for i in range(1, len(s)):
    print(f"launch {i - 1} -> {i}: last workgroup drained -> first workgroup's first stamp {(s[i][:, :, 14].min().item() - s[i - 1][:, :, 15].max().item()) / 100.0:.2f} us;"
          f" -> first workgroup's entry {(s[i][:, :, 12].min().item() - s[i - 1][:, :, 15].max().item()) / 100.0:.2f} us;"
          f" launch period {(s[i][:, :, 14].min().item() - s[i - 1][:, :, 14].min().item()) / 100.0:.2f} us")
