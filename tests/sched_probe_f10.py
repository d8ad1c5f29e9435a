"""Script (not collected): precision-schedule candidates on the cfg2-sized model at step_freq 10 (201 calls), ddpm and euler,
against the CPU oracle on 32 dialogs (2048 positions).  Usage: python tests/sched_probe_f10.py [ddpm|euler]"""
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
    cfg = DiDiConfig(diffusion_steps=2000)
    sd = synthetic_state_dict(cfg, seed=1234)
    B, Bo, C, S, f = 40, 32, 64, 64, 10
    ids, mask = synthetic_batch(B, C, cfg.vocab, seed=11)
    n_calls = len(range(cfg.diffusion_steps, 1, -f)) + 1
    n_draws = n_calls if mode == "ddpm" else n_calls + 1
    noise = torch.randn(n_draws, B, S, cfg.dec_dim, generator=torch.Generator().manual_seed(21))

    class Tick(ListNoise):
        def __call__(self, shape):
            if self.i % 20 == 0:
                print(f"  oracle draw {self.i}", flush=True)
            return super().__call__(shape)

    t0 = time.time()
    want_trace = []
    want = oracle_sample(Context(ids[:Bo], mask[:Bo]), OracleDiDi(sd, config_to_oracle_config(cfg)), mode, f, max_len=S, raw_output=True,
                         noise_source=Tick([noise[i, :Bo] for i in range(n_draws)]), trace=want_trace)
    want_ids = want.argmax(-1)
    top2 = want.topk(2, -1).values
    gap = top2[..., 0] - top2[..., 1]
    print(f"[{mode}] oracle {time.time() - t0:.0f}s; median gap {float(gap.median()):.3f}", flush=True)
    ctx = RawContext(ids.cuda(), mask.cuda())
    nd = noise.cuda()
    cands = [("fp16x2", None, None), ("bf16x3", None, None)]
    for sched in (((0.35, "fp16"), (0.064, "fp16x2"), (0, "fp16x3")), ((0.5, "fp16"), (0.064, "fp16x2"), (0, "fp16x3"))):
        for tail in (((0, "fp16x3"), (0, "fp16x2")), ((32, "fp16x3"), (100, "fp16x2")), ((64, "fp16x3"), (150, "fp16x2")),
                     ((128, "fp16x3"), (201, "fp16x2")), ((32, "fp16x3"), (0, "fp16x2")), ((64, "fp16x3"), (0, "fp16x2"))):
            cands.append(("fp16-sched", sched, tail))
    t64 = ((64, "fp16x3"),)
    cands += [("fp16-sched", ((0, "fp16x2"),), t64), ("fp16-sched", ((0.7, "fp16"), (0, "fp16x2")), t64),
              ("fp16-sched", ((0.5, "fp16"), (0, "fp16x2")), t64), ("fp16-sched", ((0.35, "fp16"), (0, "fp16x2")), t64), ("fp16", None, None)]
    for precision, sched, tail in cands:
        model = DiDiSampler(sd, cfg, device="cuda", precision=precision, schedule=sched, tail_calls=tail)
        trace = []
        torch.cuda.synchronize()
        t0 = time.time()
        got, _ = sample(ctx, model, mode, f, max_len=S, noise=nd, return_ids=True, trace=trace)
        torch.cuda.synchronize()
        dt = time.time() - t0
        model.check_fault("probe")
        diff = got[:Bo].cpu() != want_ids
        flips = int(diff.sum())
        worst = max(rel_l2(a[:Bo], b) for a, b in zip(trace, want_trace))
        last = rel_l2(trace[-1][:Bo], want_trace[-1])
        gaps = [round(float(x), 5) for x in gap[diff]]
        print(f"[{mode}] {precision} sched={sched} tail={tail}: flips {flips}/{want_ids.numel()} max x_t {worst:.2e} last {last:.2e} "
              f"gaps at flips {gaps}  ({dt:.2f}s with trace)", flush=True)
        del model, trace


if __name__ == "__main__":
    main()
