"""Step kernel (sampler update + Philox noise + time embedding + BertEmbeddings LayerNorm) at cfg2 size: in-kernel noise vs injected noise."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from didi_amd import _lib, ops  # noqa: E402

M, S, H = 16384, 64, 768
g = torch.Generator(device="cuda").manual_seed(0)
x0 = torch.randn(M, H, device="cuda", generator=g)
e = torch.randn(M, H, device="cuda", generator=g)
te, pos, tok = torch.randn(2001, H, device="cuda", generator=g), torch.randn(512, H, device="cuda", generator=g), torch.randn(H, device="cuda", generator=g)
gam, bet = torch.ones(H, device="cuda"), torch.zeros(H, device="cuda")
hi = torch.empty(M, H, dtype=torch.float16, device="cuda")
lo = torch.empty_like(hi)


def run(E):
    ops.step_embed_ln(_lib.STEP_DDPM, M, S, H, X0=x0, E=E, cx0=0.9, ce=0.1, cin=1.0, seed=7, offset=11, time_embeds=te, t=5, pos_emb=pos,
                      tok_type0=tok, ln_gamma=gam, ln_beta=bet, ln_eps=1e-12, h_hi=hi, h_lo=lo)


for name, E in (("in-kernel Philox", None), ("injected noise", e)):
    for _ in range(3):
        run(E)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run(E)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    nbytes = M * H * (4 + 4) + (M * H * 4 if E is not None else 0)
    print(f"{name:18s} {us:6.1f} us   {nbytes / us / 1e3:7.0f} GB/s of algorithmic traffic ({nbytes / 1e6:.0f} MB)")
