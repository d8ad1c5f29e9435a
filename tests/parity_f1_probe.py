#!/usr/bin/env python3
"""Parity of the HEADLINE configuration (BASELINE configs[1]: T = 2000, step_freq = 1, 2000 denoiser calls) against the CPU oracle.

cfg2-sized model, the GPU samples 32 dialogs (M = 2048 rows: the 256x192 GEMM tile path bench.py runs) with injected noise, the
fp32 CPU oracle the first `Bo` of them (default 16 = 1024 token positions): per precision / schedule the max x_t rel-L2 over all
2000 calls, its profile every 200 calls, and the number of differing token ids.  The oracle's 2000 calls take ~6 minutes of host
CPU, which is why this is a script (its output is committed under profiles/) and not a collected test; the collected tests run the
same model over the 201 calls of step_freq = 10 (tests/test_gpu_fullsize.py).   python tests/parity_f1_probe.py [Bo] [mode] [f] [filter]
Environment PROBE_SEEDS="weights,batch,noise" (default 1234,11,21: the seeds the default schedule was chosen on) runs the same
probe on other random-init weights / batch / noise: the held-out check that the schedule is not fitted to one draw.
Test infrastructure (imports oracle/).
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from golden_util import ListNoise, config_to_oracle_config  # noqa: E402
from oracle.model import OracleDiDi  # noqa: E402
from oracle.sampling import Context, sample as oracle_sample  # noqa: E402


class Tick(ListNoise):
    def __call__(self, shape):
        if self.i % 100 == 0:
            print(f"  oracle draw {self.i}/{len(self.draws)}  ({time.time() - T0:.0f}s)", flush=True)
        return super().__call__(shape)


T0 = time.time()


def main():
    from didi_amd.config import DiDiConfig
    from didi_amd.model import DEFAULT_SCHEDULE, DEFAULT_SCHEDULE_EULER, DiDiSampler
    from didi_amd.sampling import sample
    from didi_amd.synthetic import RawContext, synthetic_batch, synthetic_state_dict

    Bo = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    mode = sys.argv[2] if len(sys.argv) > 2 else "ddpm"
    f = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    B, C, S = 32, 64, 64
    cfg = DiDiConfig(diffusion_steps=2000)
    sw, sb, sn = (int(x) for x in os.environ.get("PROBE_SEEDS", "1234,11,21").split(","))
    print(f"seeds: weights {sw}, batch {sb}, noise {sn}", flush=True)
    sd = synthetic_state_dict(cfg, seed=sw)
    if os.environ.get("PROBE_TRAINED_LIKE"):  # outlier features, spread LayerNorm gammas, x4 attention / FFN weights (tests/test_gpu_precision.py)
        from test_gpu_precision import _trained_like
        sd = _trained_like(sd)
        print("weights: trained-like statistics", flush=True)
    ids, mask = synthetic_batch(B, C, cfg.vocab, seed=sb)
    n_calls = len(range(cfg.diffusion_steps, 1, -f)) + 1
    n_draws = n_calls if mode == "ddpm" else n_calls + 1
    g = torch.Generator().manual_seed(sn)
    noise = torch.randn(n_draws, B, S, cfg.dec_dim, generator=g)
    want_trace = []
    want = oracle_sample(Context(ids[:Bo], mask[:Bo]), OracleDiDi(sd, config_to_oracle_config(cfg)), mode, f, max_len=S, raw_output=True,
                         noise_source=Tick([noise[i, :Bo] for i in range(n_draws)]), trace=want_trace)
    want_ids = want.argmax(-1)
    print(f"oracle: {n_calls} calls x {Bo} dialogs in {time.time() - T0:.0f}s", flush=True)
    noise_dev = noise.cuda()
    ctx = RawContext(ids.cuda(), mask.cuda())
    def sched(*pairs):
        return ("fp16 planes, " + " / ".join(f"{p} above t={t}" for t, p in pairs), dict(precision="fp16x2", schedule=pairs))

    configs = [
        ("bf16x3", dict(precision="bf16x3")),
        sched((0, "fp16x3")),
        ("fp16-sched (default) " + str(DEFAULT_SCHEDULE_EULER if mode == "euler" else DEFAULT_SCHEDULE), dict(precision="fp16-sched")),
        sched((700, "fp16"), (128, "fp16x2"), (0, "fp16x3")),
        sched((600, "fp16"), (0, "fp16x3")),
        ("fp16x2", dict(precision="fp16x2")),
    ]
    if len(sys.argv) > 4:
        configs = [c for c in configs if any(k in c[0] for k in sys.argv[4].split(","))]
    for name, kw in configs:
        model = DiDiSampler(sd, cfg, device="cuda", **kw)
        trace = []
        t1 = time.time()
        got, _ = sample(ctx, model, mode, f, max_len=S, noise=noise_dev, return_ids=True, trace=trace)
        model.check_fault(name)
        torch.cuda.synchronize()
        dt = time.time() - t1
        errs = [((a[:Bo].cpu().double() - b.double()).norm() / b.double().norm()).item() for a, b in zip(trace, want_trace)]
        flips = int((got[:Bo].cpu() != want_ids).sum())
        step = max(1, n_calls // 10)
        prof = " ".join(f"{errs[i]:.1e}" for i in range(0, n_calls, step))
        print(f"[{mode} f={f}] {name}: max x_t rel-L2 {max(errs):.3e} (call {errs.index(max(errs))}), last {errs[-1]:.2e}; ids differ {flips} of "
              f"{want_ids.numel()} ({100 - 100.0 * flips / want_ids.numel():.3f} % agree); GPU loop {dt:.1f}s\n    every {step}th call: {prof}", flush=True)
        del model, trace
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
