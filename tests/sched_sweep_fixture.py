#!/usr/bin/env python3
"""Precision-schedule candidates against the FULL-SIZE REFERENCE fixtures (tests/golden/cfg2_*: the imported reference sampler on
HF modules, oracle/make_golden.py fullsize) -- no CPU oracle runs here, a candidate costs one GPU loop (~6 s).
    python tests/sched_sweep_fixture.py [case] [filter]
Per candidate: x_t rel-L2 at every stored call, final x0 rel-L2, ids differing of 1024, loop seconds.  Not collected by pytest;
its output is committed under profiles/."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from golden_util import load_case, sd_hash, trained_like  # noqa: E402


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def main():
    from didi_amd.config import DiDiConfig
    from didi_amd.model import DiDiSampler
    from didi_amd.sampling import sample
    from didi_amd.synthetic import RawContext, synthetic_batch, synthetic_state_dict

    case = sys.argv[1] if len(sys.argv) > 1 else "cfg2_ddpm_f1"
    t, meta = load_case(case)
    run, keep = meta["run"], meta["kept_calls"]
    cfg = DiDiConfig(diffusion_steps=meta["spec"]["T"])
    sd = synthetic_state_dict(cfg, seed=run["weight_seed"])
    if run["weights"] == "trained_like":
        sd = trained_like(sd)
    sd.update({k[2:]: (v.reshape(()) if k == "w.std_0" else v) for k, v in t.items() if k.startswith("w.")})
    assert sd_hash(sd) == meta["weights_sha256"]
    Bg, Bo, Bx, S = run["B_gpu"], run["B_ref"], run["B_xt"], run["max_len"]
    ids, mask = synthetic_batch(Bg, 64, cfg.vocab, seed=run["batch_seed"])
    ctx = RawContext(ids.cuda(), mask.cuda())
    T = cfg.diffusion_steps

    def sched(pairs, tail=64, wd=4, ad=True):
        name = " / ".join(f"{p} above {t_}" for t_, p in pairs) + f"; x3 tail {tail}; weight dither {wd}; act dither {int(ad)}"
        return name, dict(precision="fp16x2", schedule=pairs, tail_calls=((tail, "fp16x3"),) if tail else (), weight_dither=wd, act_dither=ad)

    cands = [
        ("default fp16-sched", dict(precision="fp16-sched")),
        ("default fp16-sched + act dither", dict(precision="fp16-sched", act_dither=True)),
        sched(((128, "fp16"), (0, "fp16x3"))),
        sched(((128, "fp16"), (0, "fp16x3")), ad=False),
        sched(((64, "fp16"), (0, "fp16x3"))),
        sched(((0, "fp16"),), tail=64),
        sched(((0, "fp16"),), tail=32),
        sched(((0, "fp16"),), tail=64, ad=False),
        sched(((128, "fp16"), (0, "fp16x2")), tail=32),
        sched(((300, "fp16"), (128, "fp16x2"), (0, "fp16x3")), tail=64),
    ]
    if len(sys.argv) > 2:
        cands = [c for c in cands if any(k in c[0] for k in sys.argv[2].split(","))]
    want_ids = t["out.ids"]
    for name, kw in cands:
        model = DiDiSampler(sd, cfg, device="cuda", **kw)
        trace = []
        torch.cuda.synchronize()
        t0 = time.time()
        got, _ = sample(ctx, model, run["mode"], run["step_freq"], max_len=S, return_ids=True, trace=trace, philox_seed=run["philox_seed"])
        model.check_fault(name)
        torch.cuda.synchronize()
        dt = time.time() - t0
        errs = {k: rel_l2(trace[k][:Bx], t[f"xt.{k}"]) for k in keep}
        x0 = next(iter(model._bufs.values()))["x0"].view(Bg, S, -1)
        flips = int((got[:Bo].cpu() != want_ids).sum())
        print(f"[{case}] {name}\n    loop {dt:.2f} s; ids differ {flips} of {want_ids.numel()}; final x0 {rel_l2(x0[:Bo], t['out.x0']):.2e}; max x_t "
              f"{max(errs.values()):.2e}; " + " ".join(f"{k}:{e:.1e}" for k, e in errs.items()), flush=True)
        del model, trace
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
