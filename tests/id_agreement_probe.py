#!/usr/bin/env python3
"""Token-id agreement of the HIP sampler with the fp32 CPU oracle over a larger sample than the fixtures hold.

BASELINE config-1 model (random-init, seed of the fixture), 96 synthetic dialogs x 23 positions = 2208 ids,
DDPM T=50, identical injected noise.  Prints the agreement per precision (DESIGN.md section 2 quotes it).
Test infrastructure (imports oracle/): run by hand on a GPU box, `python tests/id_agreement_probe.py`.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from golden_util import ListNoise, config1, config1_state_dict, config_to_oracle_config  # noqa: E402
from oracle.model import OracleDiDi  # noqa: E402
from oracle.sampling import Context, sample as oracle_sample  # noqa: E402


def main():
    from didi_amd.model import DiDiSampler
    from didi_amd.sampling import sample
    from didi_amd.synthetic import RawContext

    cfg, sd = config1(), config1_state_dict()
    B, C = 96, 24
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(3, cfg.vocab, (B, C), generator=g)
    mask = torch.ones(B, C, dtype=torch.long)
    lens = torch.randint(8, C + 1, (B,), generator=g)
    for b in range(B):
        mask[b, lens[b]:] = 0
    torch.manual_seed(5)
    draws = [torch.randn(B, C - 1, cfg.dec_dim) for _ in range(cfg.diffusion_steps + 1)]
    trace = []
    want = oracle_sample(Context(ids, mask), OracleDiDi(sd, config_to_oracle_config(cfg)), "ddpm", 1, raw_output=True,
                         noise_source=ListNoise(draws), trace=trace).argmax(-1)
    n_calls = len(trace)
    ctx = RawContext(ids.cuda(), mask.cuda())
    for precision in ("bf16x3", "fp16", "bf16"):
        model = DiDiSampler(sd, cfg, device="cuda", precision=precision)
        tr = []
        got, _ = sample(ctx, model, "ddpm", 1, noise=torch.stack(draws[:n_calls]), return_ids=True, trace=tr)
        diff = int((got.cpu() != want).sum())
        worst = max(((a.cpu() - b).norm() / b.norm()).item() for a, b in zip(tr, trace))
        print(f"{precision:7s} ids differ {diff} of {want.numel()} ({100 - 100.0 * diff / want.numel():.3f} % agree)  max x_t rel-L2 {worst:.3e}")


if __name__ == "__main__":
    main()
