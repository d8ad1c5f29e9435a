#!/usr/bin/env python3
"""Where a gemm_pp_kernel workgroup's time goes, from in-kernel s_memtime stamps.

Needs the DIAGNOSTIC build of the library (never the product one):
    make -C didi_amd/csrc EXTRA=-DDIDI_STAMPS BUILD=build_stamps LIB=../../tools/ab/libdidi_hip_stamps.so
    DIDI_HIP_LIB=tools/ab/libdidi_hip_stamps.so python tools/ln_stamps.py [--K 768] [--plain] [--M 16384]
Launches the shape back to back for ~1 s (the clock settles under load), then reads the stamps of the last launch: per wave group
(X = waves 0-3, the early group; Y = waves 4-7) the median over workgroups of every segment, in cycles and in microseconds at
the in-kernel clock (s_memtime / s_memrealtime).  Stamps: 0 kernel start, 1 first K-tile landed, 2 K-loop done, LayerNorm form:
3 bias + residual added (both halves' residual planes arrived), 4 local statistics, 5 band statistics collected (exchange),
6 stores issued, 7 stores drained (vmcnt(0)).
"""
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
ap.add_argument("--plain", action="store_true", help="the plain GEMM (hi plane out, no residual / LayerNorm) instead of didi_gemm_ln")
ap.add_argument("--gelu", action="store_true")
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
stamps = torch.zeros(n_wg * 2 * 16, dtype=torch.int64, device="cuda")


def launch():
    if a.plain:
        ops.gemm(x_hi, x_lo, w_hi, w_lo, bias=bias, act="gelu" if a.gelu else None, precision=a.precision, out_f32=False, out_planes=True, out_lo=False)
    else:
        ops.gemm_ln(x_hi, x_lo, w_hi, w_lo, bias, r, gamma, beta, 1e-12, precision=a.precision, out=out)


set_stamps(None)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(50):
    launch()
torch.cuda.synchronize()
n = 2000 if a.K <= 768 else 800
ev[0].record()
for _ in range(n):
    launch()
ev[1].record()
set_stamps(stamps.data_ptr())
launch()
set_stamps(None)
torch.cuda.synchronize()
us = 1e3 * ev[0].elapsed_time(ev[1]) / n
s = stamps.view(n_wg, 2, 16).cpu().double()
clk = ((s[:, :, 7] - s[:, :, 0]) / ((s[:, :, 15] - s[:, :, 14]).clamp_min(1))).median().item() * 100.0  # MHz
names = {0: "start->first tile landed", 1: "K-loop", 2: "bias+residual add (residual planes arrive)", 3: "local statistics", 4: "exchange with band mates",
         5: "normalise + issue stores", 6: "stores drain"} if not a.plain else {0: "start->first tile landed", 1: "K-loop", 2: "epilogue: bias/act + issue stores", 6: "stores drain"}
print(f"shape M={a.M} N={a.N} K={a.K} {'plain' if a.plain else 'LayerNorm'} {a.precision}: {us:.1f} us per launch (events over {n} launches), in-kernel clock {clk:.0f} MHz, "
      f"{n_wg} workgroups")
t0 = s[:, :, 0].min()
for grp, gname in ((0, "X (waves 0-3)"), (1, "Y (waves 4-7)")):
    print(f"  group {gname}: workgroup start spread {(s[:, grp, 0].max() - t0).item():.0f} cycles; total {((s[:, grp, 7] - s[:, grp, 0]).median()).item():.0f} cycles "
          f"= {((s[:, grp, 7] - s[:, grp, 0]).median()).item() / clk:.1f} us")
    order = [0, 1, 2, 3, 4, 5, 6, 7] if not a.plain else [0, 1, 2, 6, 7]
    for i, j in zip(order, order[1:]):
        d = (s[:, grp, j] - s[:, grp, i])
        print(f"    {names[i]:<46s} median {d.median().item():8.0f} cyc = {d.median().item() / clk:5.2f} us   (p10 {d.quantile(0.1).item():.0f}, p90 {d.quantile(0.9).item():.0f})")
last = (s[:, :, 7].max() - t0).item()
print(f"  first workgroup start -> last stamp anywhere: {last:.0f} cycles = {last / clk:.1f} us")
