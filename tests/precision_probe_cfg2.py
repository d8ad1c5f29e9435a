#!/usr/bin/env python3
"""CPU probe at the BENCHMARKED model size: which operand roundings cost the parity of the one-product fp16 mode?

cfg2-sized model (12 x 768 decoder), T = 2000, step_freq = 10 (201 calls), B dialogs, injected noise.  The fp32 CPU oracle is
the reference; the variants round the DECODER's matmul operands to fp16 (products and accumulation stay fp32, as on the MFMA):
    A+W   activations and weights (= the HIP "fp16" mode: one product hi*hi)
    W     weights only  (= two products A_hi*W_hi + A_lo*W_hi: the activation remainder is multiplied too)
    A     activations only (= two products A_hi*W_hi + A_hi*W_lo)
    attn  as A+W but the attention matmuls (QK^T, PV) exact
Test infrastructure only (imports oracle/).   python tests/precision_probe_cfg2.py [B] [mode] [step_freq] [variants,comma-separated]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from golden_util import ListNoise, config_to_oracle_config  # noqa: E402
import oracle.model as om  # noqa: E402
import oracle.transformer as tr  # noqa: E402
from didi_amd.config import DiDiConfig  # noqa: E402
from didi_amd.synthetic import synthetic_batch, synthetic_state_dict  # noqa: E402
from oracle.model import OracleDiDi  # noqa: E402
from oracle.sampling import Context, sample  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    mode = sys.argv[2] if len(sys.argv) > 2 else "ddpm"
    f = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    only = sys.argv[4].split(",") if len(sys.argv) > 4 else None
    tails = [int(t) for t in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0]  # exact arithmetic in the last `tail` calls
    cfg = DiDiConfig(diffusion_steps=2000)
    sd = synthetic_state_dict(cfg, seed=1234)
    ids, mask = synthetic_batch(40, 64, cfg.vocab, seed=11)
    ids, mask = ids[:B], mask[:B]
    n_calls = len(range(cfg.diffusion_steps, 1, -f)) + 1
    n_draws = n_calls if mode == "ddpm" else n_calls + 1
    g = torch.Generator().manual_seed(21)
    noise = torch.randn(n_draws, 40, 64, cfg.dec_dim, generator=g)[:, :B].contiguous()
    model = OracleDiDi(sd, config_to_oracle_config(cfg))
    ctx_in = Context(ids, mask)

    h16 = lambda x: x.to(torch.float16).float()  # noqa: E731
    wcache = {}
    state = {"on": False, "A": False, "W": False, "attn": False, "call": 0, "tail": 0}

    def lin(sd_, prefix, x):
        w = sd_[prefix + ".weight"]
        if state["on"] and state["W"]:
            if prefix not in wcache:
                wcache[prefix] = h16(w)
            w = wcache[prefix]
        if state["on"] and state["A"]:
            x = h16(x)
        return F.linear(x, w, sd_.get(prefix + ".bias"))

    def attend(q, k, v, add_mask):
        r = h16 if (state["on"] and state["attn"]) else (lambda t: t)
        d = q.shape[-1]
        scores = torch.matmul(r(q), r(k).transpose(2, 3)) * (d**-0.5)
        if add_mask is not None:
            scores = scores + add_mask
        p = torch.softmax(scores, dim=-1)
        o = torch.matmul(r(p), r(v)).transpose(1, 2)
        return o.reshape(o.shape[0], o.shape[1], -1)

    dec0 = om.bert_decoder

    def dec(*a, **k):
        state["on"] = state["call"] < n_calls - state["tail"]
        state["call"] += 1
        try:
            return dec0(*a, **k)
        finally:
            state["on"] = False

    tr._lin, tr._attend, om.bert_decoder = lin, attend, dec
    cls_w = model.sd["classifier.weight"]
    ref = None
    variants = [("fp32", 0, 0, 0, 0)] + [(n, a, w, at, tl) for tl in tails
                                          for n, a, w, at in (("A+W", 1, 1, 1), ("W", 0, 1, 0), ("A", 1, 0, 1), ("attn-exact", 1, 1, 0))]
    for name, A, W, attn, tail in variants:
        if only and name not in only and name != "fp32":
            continue
        state.update(A=bool(A), W=bool(W), attn=bool(attn), call=0, tail=tail)
        wcache.clear()
        if A or W:
            model.classifier = lambda x: F.linear(h16(x) if A else x, h16(cls_w) if W else cls_w)
        t0 = time.time()
        trace = []
        logits = sample(ctx_in, model, mode, f, max_len=64, raw_output=True, noise_source=ListNoise(list(noise)), trace=trace)
        got = logits.argmax(-1)
        if ref is None:
            ref = (trace, got)
            top2 = logits.topk(2, -1).values
            print(f"fp32 reference: {n_calls} calls x {B} dialogs, {time.time() - t0:.0f}s; median top-2 gap {float((top2[..., 0] - top2[..., 1]).median()):.3f}", flush=True)
            continue
        errs = [((a - b).norm() / b.norm()).item() for a, b in zip(trace, ref[0])]
        prof = " ".join(f"{errs[i]:.1e}" for i in range(0, len(errs), max(1, len(errs) // 10)))
        print(f"{name:10s} tail {tail:4d}  every {max(1, len(errs) // 10)}th call: {prof}", flush=True)
        print(f"{name:10s} max x_t rel-L2 {max(errs):.3e} (call {errs.index(max(errs))}), last call {errs[-1]:.3e}, median {sorted(errs)[len(errs) // 2]:.3e}; "
              f"ids differ {int((got != ref[1]).sum())} of {got.numel()}  ({time.time() - t0:.0f}s)", flush=True)


if __name__ == "__main__":
    main()
