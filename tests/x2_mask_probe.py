"""Script (not collected): which decoder GEMMs need the weights' remainder plane in an fp16x2 call?  cfg2-sized model, ddpm,
step_freq 10 (201 calls: fp16x2 everywhere + the default 64-call fp16x3 tail), DIDI_OPT_X2_GEMMS masks, against the CPU oracle on
32 dialogs.  Usage: python tests/x2_mask_probe.py [ddpm|euler]"""
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
    from didi_amd import _lib
    from didi_amd.config import DiDiConfig
    from didi_amd.model import DiDiSampler
    from didi_amd.sampling import sample
    from didi_amd.synthetic import RawContext, synthetic_batch, synthetic_state_dict
    from oracle.model import OracleDiDi
    from oracle.sampling import Context, sample as oracle_sample

    mode = sys.argv[1] if len(sys.argv) > 1 else "ddpm"
    cfg = DiDiConfig(diffusion_steps=2000)
    sd = synthetic_state_dict(cfg, seed=1234)
    B, Bo, C, S, f = 256, 32, 64, 64, 10
    ids, mask = synthetic_batch(B, C, cfg.vocab, seed=11)
    n_calls = len(range(cfg.diffusion_steps, 1, -f)) + 1
    n_draws = n_calls if mode == "ddpm" else n_calls + 1
    g = torch.Generator(device="cuda").manual_seed(21)
    noise = torch.empty(n_draws, B, S, cfg.dec_dim, device="cuda")
    for i in range(n_draws):
        noise[i].normal_(generator=g)
    host = noise[:, :Bo].cpu()

    class Tick(ListNoise):
        def __call__(self, shape):
            if self.i % 20 == 0:
                print(f"  oracle draw {self.i}", flush=True)
            return super().__call__(shape)

    t0 = time.time()
    want_trace = []
    want = oracle_sample(Context(ids[:Bo], mask[:Bo]), OracleDiDi(sd, config_to_oracle_config(cfg)), mode, f, max_len=S, raw_output=True,
                         noise_source=Tick([host[i] for i in range(n_draws)]), trace=want_trace)
    want_ids = want.argmax(-1)
    print(f"[{mode}] oracle {time.time() - t0:.0f}s", flush=True)
    ctx = RawContext(ids.cuda(), mask.cuda())
    names = ["qkv", "attn_out", "cross_q", "cross_out", "ffn_up", "ffn_down"]
    masks = [63, 0, 48, 15, 1, 2, 4, 8, 16, 32, 63 - 1, 63 - 16, 63 - 32, 63 - 4, 63 - 2 - 8, 1 + 16 + 32]
    for m in masks:
        _lib.set_option("X2_GEMMS", m)
        model = DiDiSampler(sd, cfg, device="cuda", precision="fp16x2", schedule=((0, "fp16x2"),), tail_calls=((64, "fp16x3"),))
        trace = []
        got, _ = sample(ctx, model, mode, f, max_len=S, noise=noise, return_ids=True, trace=trace)
        model.check_fault("probe")
        worst = max(rel_l2(a[:Bo], b) for a, b in zip(trace[:n_calls - 64], want_trace[:n_calls - 64]))
        del trace
        torch.cuda.synchronize()
        t1 = time.time()
        sample(ctx, model, mode, f, max_len=S, noise=noise, return_ids=True)
        torch.cuda.synchronize()
        dt = time.time() - t1
        flips = int((got[:Bo].cpu() != want_ids).sum())
        on = "+".join(n for i, n in enumerate(names) if m >> i & 1) or "none"
        print(f"[{mode}] x2 on {on:55s} (mask {m:2d}): max x_t over the fp16x2 calls {worst:.2e}  flips {flips}/{want_ids.numel()}  loop {dt:.3f}s", flush=True)
        del model
    _lib.set_option("X2_GEMMS", 63)


if __name__ == "__main__":
    main()
