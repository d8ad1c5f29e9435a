#!/usr/bin/env python3
"""CPU probe: what would a single-product 16-bit operand format cost in parity?

Runs the CPU oracle on BASELINE config 1 (expected outputs = the reference run, tests/golden) with the
DECODER's matmul operands (Linear inputs/weights, Q/K/V, softmax probabilities, classifier) rounded to
bf16 or fp16 -- products and accumulation stay fp32, as on the MFMA -- and reports the per-call x_t
rel-L2 and the token-id agreement.  Test infrastructure only (imports oracle/).

    python tests/precision_probe.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from golden_util import ListNoise, config1, config1_inputs, config1_state_dict, config_to_oracle_config, load_case  # noqa: E402
import oracle.transformer as tr  # noqa: E402
from oracle.model import OracleDiDi  # noqa: E402
from oracle.sampling import Context, sample  # noqa: E402

ROUND = {"dtype": None}


def rnd(x):
    return x if ROUND["dtype"] is None else x.to(ROUND["dtype"]).float()


def main():
    t, meta = load_case("config1_ddpm")
    model = OracleDiDi(config1_state_dict(), config_to_oracle_config(config1()))
    ids, mask = config1_inputs()
    ctx = model.encode_context(ids, mask)  # encoder + adapter stay fp32-class in the product too

    import oracle.model as om

    lin0, att0, dec0 = tr._lin, tr._attend, om.bert_decoder
    state = {"on": False}

    def r(x):
        return rnd(x) if state["on"] else x

    def lin(sd, prefix, x):
        return F.linear(r(x), r(sd[prefix + ".weight"]), sd.get(prefix + ".bias"))

    def attend(q, k, v, add_mask):
        d = q.shape[-1]
        scores = torch.matmul(r(q), r(k).transpose(2, 3)) * (d**-0.5)
        if add_mask is not None:
            scores = scores + add_mask
        p = torch.softmax(scores, dim=-1)
        o = torch.matmul(r(p), r(v)).transpose(1, 2)
        return o.reshape(o.shape[0], o.shape[1], -1)

    def dec(*a, **k):  # only the decoder is rounded
        state["on"] = True
        try:
            return dec0(*a, **k)
        finally:
            state["on"] = False

    tr._lin, tr._attend, om.bert_decoder = lin, attend, dec
    cls0 = model.classifier
    cls_w = model.sd["classifier.weight"]

    shape = tuple(t["noise.0"].shape)
    for name, dt in (("fp32", None), ("bf16", torch.bfloat16), ("fp16", torch.float16)):
        ROUND["dtype"] = dt
        model.classifier = (lambda x: F.linear(rnd(x), rnd(cls_w))) if dt is not None else cls0
        torch.manual_seed(0xBADB01)
        draws = [torch.randn(shape)] + [torch.empty(shape).normal_(0, 1) for _ in range(int(meta["n_noise"]) - 1)]
        trace = []

        logits = sample(Context(ids, mask), model, "ddpm", 1, raw_output=True, noise_source=ListNoise(draws), trace=trace)
        worst = max(((trace[i] - t[f"xt.{i}"]).norm() / t[f"xt.{i}"].norm()).item() for i in meta["kept_calls"])
        got = logits.argmax(-1)
        match = (got == t["out.ids"]).float().mean().item()
        top2 = logits.topk(2, -1).values
        print(f"{name}: max x_t rel-L2 {worst:.3e}  id match {match:.5f} ({int((got != t['out.ids']).sum())} of {got.numel()} differ)"
              f"  median top-2 logit gap {float((top2[..., 0] - top2[..., 1]).median()):.3e}")


if __name__ == "__main__":
    main()
