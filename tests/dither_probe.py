"""Script (not collected): does cycling differently rounded fp16 weight planes over the one-product calls (weight_dither = P) cut
the coherent part of the weight-rounding error?  cfg2-sized model, step_freq 10 (201 calls), against the CPU oracle on 32 dialogs.
Usage: python tests/dither_probe.py [ddpm|euler] [f]"""
import os
import sys
import time

import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
from golden_util import ListNoise, config_to_oracle_config  # noqa: E402


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def main():
    from didi_amd.config import DiDiConfig
    from didi_amd.model import DiDiSampler
    from didi_amd.sampling import sample
    from didi_amd.synthetic import RawContext, synthetic_batch, synthetic_state_dict
    from oracle.model import OracleDiDi
    from oracle.sampling import Context, sample as oracle_sample

    mode = sys.argv[1] if len(sys.argv) > 1 else "ddpm"
    f = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    sw, sb, sn = (int(x) for x in os.environ.get("PROBE_SEEDS", "1234,11,21").split(","))
    cfg = DiDiConfig(diffusion_steps=2000)
    sd = synthetic_state_dict(cfg, seed=sw)
    B, Bo, C, S = 40, int(os.environ.get("PROBE_BO", "32")), 64, 64
    ids, mask = synthetic_batch(B, C, cfg.vocab, seed=sb)
    n_calls = len(range(cfg.diffusion_steps, 1, -f)) + 1
    n_draws = n_calls if mode == "ddpm" else n_calls + 1
    noise = torch.randn(n_draws, B, S, cfg.dec_dim, generator=torch.Generator().manual_seed(sn))

    class Tick(ListNoise):
        def __call__(self, shape):
            if self.i % 100 == 0:
                print(f"  oracle draw {self.i}", flush=True)
            return super().__call__(shape)

    t0 = time.time()
    want_trace = []
    want = oracle_sample(Context(ids[:Bo], mask[:Bo]), OracleDiDi(sd, config_to_oracle_config(cfg)), mode, f, max_len=S, raw_output=True,
                         noise_source=Tick([noise[i, :Bo] for i in range(n_draws)]), trace=want_trace)
    want_ids = want.argmax(-1)
    print(f"[{mode} f={f}] oracle {time.time() - t0:.0f}s", flush=True)
    ctx = RawContext(ids.cuda(), mask.cuda())
    nd = noise.cuda()
    T = cfg.diffusion_steps
    x3tail = ((64, "fp16x3"),)
    cands = []
    for P in (0, 2, 4, 8):
        cands.append((f"fp16 everywhere, dither {P}", dict(precision="fp16", weight_dither=P)))
    for P in (0, 4, 8):
        cands.append((f"default schedule, dither {P}", dict(precision="fp16-sched", weight_dither=P)))
        for thr in (0.25, 0.15):
            cands.append((f"fp16 above {thr} T, dither {P}", dict(precision="fp16x2", weight_dither=P, tail_calls=x3tail,
                                                                  schedule=((thr, "fp16"), (0.064, "fp16x2"), (0, "fp16x3")))))
    if os.environ.get("PROBE_CANDS"):  # "thr:P,thr:P,..."
        cands = []
        for item in os.environ["PROBE_CANDS"].split(","):
            thr, P = item.split(":")
            cands.append((f"fp16 above {thr} T, dither {P}", dict(precision="fp16x2", weight_dither=int(P), tail_calls=x3tail,
                                                                  schedule=((float(thr), "fp16"), (0.064, "fp16x2"), (0, "fp16x3")))))
    for name, kw in cands:
        model = DiDiSampler(sd, cfg, device="cuda", **kw)
        trace = []
        got, _ = sample(ctx, model, mode, f, max_len=S, noise=nd, return_ids=True, trace=trace)
        model.check_fault("probe")
        errs = [rel_l2(a[:Bo], b) for a, b in zip(trace, want_trace)]
        flips = int((got[:Bo].cpu() != want_ids).sum())
        step = max(1, n_calls // 10)
        print(f"[{mode} f={f}] {name:36s}: max x_t {max(errs):.2e} (call {errs.index(max(errs))}) last {errs[-1]:.2e} flips {flips}/{want_ids.numel()}  "
              + " ".join(f"{errs[i]:.1e}" for i in range(0, n_calls, step)), flush=True)
        del model, trace
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
