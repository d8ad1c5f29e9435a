#!/usr/bin/env python3
"""Phase stamps of the training attention backward (diagnostic build: make -C didi_amd/csrc EXTRA=-DDIDI_STAMPS BUILD=build_stamps
LIB=../../tools/ab/libdidi_hip_stamps.so; DIDI_HIP_LIB=tools/ab/libdidi_hip_stamps.so python tools/attn_train_stamps.py)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from didi_amd import _lib  # noqa: E402

B, heads, S, D = 256, 12, 64, 64
H = heads * D
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, S, 3 * H, device="cuda", generator=g)
dO = torch.randn(B, S, H, device="cuda", generator=g)
out = torch.zeros_like(qkv)
stamps = torch.zeros(B * heads * 16, dtype=torch.int64, device="cuda")
a = _lib.AttnTrainArgs()
a.Q, a.K, a.V, a.dO = qkv.data_ptr(), qkv.data_ptr() + 4 * H, qkv.data_ptr() + 8 * H, dO.data_ptr()
a.B, a.heads, a.Sq, a.Sk, a.D = B, heads, S, S, D
a.ldq = a.ldk = a.ldv = 3 * H
a.ldo = H
a.causal, a.scale, a.drop_p, a.seed, a.offset = 1, 0.0, 0.1, 7, 0
a.dQ, a.dK, a.dV = out.data_ptr(), out.data_ptr() + 4 * H, out.data_ptr() + 8 * H
for _ in range(20):
    _lib.check(_lib.lib.didi_attention_train_backward(C.byref(a), _lib.current_stream()), "bwd")
a.O = stamps.data_ptr()
_lib.check(_lib.lib.didi_attention_train_backward(C.byref(a), _lib.current_stream()), "bwd")
torch.cuda.synchronize()
s = stamps.view(B * heads, 16).cpu().double()
names = ["start", "operands in LDS", "scores", "softmax", "dropout", "dV", "dP -> dS", "dQ", "dK issued", "drained"]
d = (s[:, 1:10] - s[:, 0:9])
print("median shader cycles per phase (s_memtime; ~2.1 cycles per ns under this load):")
for i in range(9):
    print(f"  {names[i]:>16s} -> {names[i + 1]:<16s} {d[:, i].median().item():8.0f} cycles  ~{d[:, i].median().item() / 2100:.2f} us")
print(f"  workgroup lifetime {(s[:, 9] - s[:, 0]).median().item():.0f} cycles ~{(s[:, 9] - s[:, 0]).median().item() / 2100:.1f} us")
