"""GPU parity at the BENCHMARKED model size (BASELINE configs[1..3]) against the CPU oracle, through the C ABI.

The fixtures from the reference pin the oracle on small models (tests/test_oracle_golden.py); here the oracle is the
reference for the cfg2-sized model (12 x 768 decoder, 2 x 1280 encoder, V = 8008 / 50 000) that bench.py times:
  * token-id agreement >= 99.9 % and every denoiser input x_t within rel-L2 1e-3 (fp16-sched, bench.py's default: fp16 planes
    with the precision schedule; fp16x2) / 1e-4 (bf16x3) over a whole T = 2000, step_freq = 10 loop (201 denoiser calls), ddpm
    and euler (= the "DDIM 200-step" of BASELINE configs[2]).  The one-product fp16 mode is run too and REPORTED: round 2 found
    that it misses both bars in ddpm at this size (3 of 1024 ids, x_t 1.8e-3), so it only has to stay within 3e-3 / 99 % here
    and is not what bench.py times.  The 2000-call loop of BASELINE configs[1] (step_freq = 1) takes the CPU oracle six
    minutes: tests/parity_f1_probe.py, output under profiles/r02_parity_f1_ddpm.log;
  * one denoiser call + rounding at the exact bench shape B = 256, S = C = 64 (M = 16 384 rows: 64 row bands = 256
    workgroups = every CU in one didi_gemm_ln launch), finite everywhere, no device fault;
  * BASELINE configs[3]'s shape: V = 50 000, S = C = 84, B = 1024 (M = 86 016) with the per-step clamp extension.
Dialogs are independent (no cross-sample op on the path), so the oracle runs a slice of the batch the GPU runs.
"""
import time

import pytest
import torch

from golden_util import ListNoise, config1, config1_state_dict, config_to_oracle_config

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


class TickNoise(ListNoise):
    """ListNoise that prints a heartbeat: the CPU oracle's 201-call loop takes minutes and a silent GPU job reads as hung."""

    def __call__(self, shape):
        if self.i % 20 == 0:
            print(f"  oracle draw {self.i}/{len(self.draws)}", flush=True)
        return super().__call__(shape)


def _ctx(ids, mask):
    from didi_amd.synthetic import RawContext

    return RawContext(ids.cuda(), mask.cuda())


@pytest.mark.parametrize("mode,weights", [("ddpm", "random"), ("euler", "random"), ("ddpm", "trained_like")])
def test_cfg2_size_loop_ids_and_xt_vs_oracle(mode, weights):
    """cfg2-sized model, T = 2000, step_freq = 10 (201 calls).  The GPU samples 40 dialogs (M = 2560 rows: the 256x192 GEMM
    tile with the fused projection + LayerNorm, the path bench.py runs); the oracle samples the first 16 with the same
    injected noise: 1024 token positions, so the 99.9 % bar allows ONE differing id.
    weights "trained_like": the same model with the statistics trained post-LN decoders show (outlier features +-40, LayerNorm
    gammas spread to 8, x4 attention / FFN weights: tests/test_gpu_precision.py) instead of plain N(0, 0.02) initialisation."""
    from didi_amd.config import DiDiConfig
    from didi_amd.model import DiDiSampler
    from didi_amd.sampling import sample
    from didi_amd.synthetic import synthetic_batch, synthetic_state_dict
    from oracle.model import OracleDiDi
    from oracle.sampling import Context, sample as oracle_sample

    cfg = DiDiConfig(diffusion_steps=2000)
    sd = synthetic_state_dict(cfg, seed=1234)
    if weights == "trained_like":
        from test_gpu_precision import _trained_like
        sd = _trained_like(sd)
    B, Bo, C, S, f = 40, 16, 64, 64, 10
    ids, mask = synthetic_batch(B, C, cfg.vocab, seed=11)
    n_calls = len(range(cfg.diffusion_steps, 1, -f)) + 1
    n_draws = n_calls if mode == "ddpm" else n_calls + 1
    g = torch.Generator().manual_seed(21)
    noise = torch.randn(n_draws, B, S, cfg.dec_dim, generator=g)
    t0 = time.time()
    want_trace = []
    want = oracle_sample(Context(ids[:Bo], mask[:Bo]), OracleDiDi(sd, config_to_oracle_config(cfg)), mode, f, max_len=S,
                         raw_output=True, noise_source=TickNoise([noise[i, :Bo] for i in range(n_draws)]), trace=want_trace)
    want_ids = want.argmax(-1)
    top2 = want.topk(2, -1).values
    print(f"\n[cfg2 {mode} f={f}] oracle: {n_calls} calls x {Bo} dialogs in {time.time() - t0:.0f} s; "
          f"median top-2 logit gap {float((top2[..., 0] - top2[..., 1]).median()):.3f}")
    assert len(want_trace) == n_calls
    noise_dev = noise.cuda()
    # "fp16x2" throughout is not a parity-holding mode of its own, and its bar here says so: it sits AT the 99.9 % line at f = 10 (1 or 2
    # of 1024 ids across builds whose arithmetic differs in the last bit -- an FMA contracted differently in the step kernel is enough),
    # and misses it at f = 1.  The modes that hold the bar are the default schedule and the three-product ones.
    modes = (("bf16x3", 1e-4, 0.999), ("fp16-sched", 1e-3, 0.999), ("fp16x2", 1e-3, 0.997), ("fp16", 3e-3, 0.99))
    for precision, xt_bar, id_bar in (modes if weights == "random" else modes[:2]):
        model = DiDiSampler(sd, cfg, device="cuda", precision=precision)
        trace = []
        got, _ = sample(_ctx(ids, mask), model, mode, f, max_len=S, noise=noise_dev, return_ids=True, trace=trace)
        model.check_fault(f"cfg2 {mode} {precision}")
        flips = int((got[:Bo].cpu() != want_ids).sum())
        worst = max(rel_l2(a[:Bo], b) for a, b in zip(trace, want_trace))
        print(f"[cfg2 {mode} f={f} {precision}] ids differ on {flips} of {want_ids.numel()} positions "
              f"({100 - 100.0 * flips / want_ids.numel():.3f} % agree); max x_t rel-L2 over {n_calls} calls {worst:.3e}")
        assert worst <= xt_bar
        assert 1.0 - flips / want_ids.numel() >= id_bar
        del model, trace


FULLSIZE_CASES = ["cfg2_ddpm_f1", "cfg2_ddpm_f1_trained_like", "cfg2_euler_f10"]


@pytest.mark.parametrize("case", FULLSIZE_CASES)
def test_cfg2_size_vs_reference_fixture(case):
    """The BENCHMARKED configuration against the REFERENCE ITSELF: tests/golden/cfg2_*.safetensors hold what the imported
    ``src.sampling.sample`` (/root/reference/src/sampling.py:38-101) produced in the build container driving HF ``BertModel``
    (12 x 768, causal + cross-attention) and a 2 x 1280 ``BlenderbotEncoder`` loaded with the seed-1234 synthetic weights, on 16
    dialogs, with its noise drawn from the product's Philox stream (oracle/make_golden.py ``fullsize``).  cfg2_ddpm_f1 is BASELINE
    configs[1] as bench.py times it -- ddpm, T = 2000, step_freq = 1: 2000 denoiser calls, 1700 of them one-product fp16 under
    the 4-way weight dither --, cfg2_ddpm_f1_trained_like the same loop on weights with trained-decoder statistics, cfg2_euler_f10
    the "DDIM 200-step" loop of configs[2].  The GPU samples 48 dialogs (M = 3072 rows: 256x192 tiles with the fused
    projection + LayerNorm, bench.py's kernels) with IN-KERNEL noise from the fixture's seed: nothing but the seed travels.
    Bars (the default precision `fp16-sched`): ids >= 99.9 % of the 1024 stored positions, x_t rel-L2 <= 1e-3 at every stored
    call (the calls bracket the error's peak near call 1800 of 2000) and for the final x0.  bf16x3 rides along at 3e-4."""
    from didi_amd.config import DiDiConfig
    from didi_amd.model import DiDiSampler
    from didi_amd.sampling import sample
    from didi_amd.synthetic import synthetic_batch, synthetic_state_dict
    from golden_util import load_case, sd_hash, trained_like

    t, meta = load_case(case)
    run, keep = meta["run"], meta["kept_calls"]
    cfg = DiDiConfig(diffusion_steps=meta["spec"]["T"])
    sd = synthetic_state_dict(cfg, seed=run["weight_seed"])
    if run["weights"] == "trained_like":
        sd = trained_like(sd)
    # tensors whose last bits vary with the host CPU's libm (the schedule buffers; trained_like's exp() on the LayerNorm gammas)
    # travel in the fixture, as they would in a checkpoint
    sd.update({k[2:]: (v.reshape(()) if k == "w.std_0" else v) for k, v in t.items() if k.startswith("w.")})
    assert sd_hash(sd) == meta["weights_sha256"], "the regenerated weights are not the ones the reference run used"
    Bg, Bo, Bx, S = run["B_gpu"], run["B_ref"], run["B_xt"], run["max_len"]
    ids, mask = synthetic_batch(Bg, 64, cfg.vocab, seed=run["batch_seed"])
    assert torch.equal(ids, t["in.input_ids"]) and torch.equal(mask, t["in.attention_mask"])
    want_ids, gap = t["out.ids"], t["out.top2_gap"]
    for precision, bar in (("fp16-sched", 1e-3), ("bf16x3", 3e-4)):
        model = DiDiSampler(sd, cfg, device="cuda", precision=precision)
        trace = []
        t0 = time.time()
        got, _ = sample(_ctx(ids, mask), model, run["mode"], run["step_freq"], max_len=S, return_ids=True, trace=trace,
                        philox_seed=run["philox_seed"], philox_offset=0)
        model.check_fault(f"{case} {precision}")
        torch.cuda.synchronize()
        dt = time.time() - t0
        assert len(trace) == int(meta["n_calls"])
        errs = {k: rel_l2(trace[k][:Bx], t[f"xt.{k}"]) for k in keep}
        x0 = next(iter(model._bufs.values()))["x0"].view(Bg, S, -1)
        e_x0 = rel_l2(x0[:Bo], t["out.x0"])
        differ = (got[:Bo].cpu() != want_ids)
        flips = int(differ.sum())
        print(f"\n[{case} {precision}] {len(trace)} calls x {Bg} dialogs in {dt:.1f} s; ids differ {flips} of {want_ids.numel()} "
              f"(reference top-2 gaps there: {[f'{g:.1e}' for g in gap[differ].tolist()]}); final x0 rel-L2 {e_x0:.2e}; x_t rel-L2 "
              + " ".join(f"{k}:{e:.1e}" for k, e in errs.items()), flush=True)
        assert max(errs.values()) <= bar and e_x0 <= bar
        assert 1.0 - flips / want_ids.numel() >= 0.999
        del model, trace
        torch.cuda.empty_cache()


def test_bench_shape_single_call_full_chip():
    """B = 256, S = C = 64 (bench.py's shape): one denoiser call + rounding in fp16x2, fp16 (the regimes of bench.py's default schedule, one by one) and bf16x3.
    Every output finite, no device fault, and the first 4 dialogs agree with the oracle."""
    from didi_amd import ops
    from didi_amd.config import DiDiConfig
    from didi_amd.model import DiDiSampler
    from didi_amd.synthetic import synthetic_batch, synthetic_state_dict
    from oracle.model import OracleDiDi

    cfg = DiDiConfig(diffusion_steps=2000)
    sd = synthetic_state_dict(cfg, seed=1234)
    B, Bo, C, S = 256, 4, 64, 64
    ids, mask = synthetic_batch(B, C, cfg.vocab, seed=0xBADB01)
    g = torch.Generator().manual_seed(5)
    x_t = torch.randn(B, S, cfg.dec_dim, generator=g)
    tids = torch.randint(1, 2001, (B, 1), generator=g)
    oracle = OracleDiDi(sd, config_to_oracle_config(cfg))
    want, _ = oracle(encoder_input_ids=ids[:Bo], encoder_attention_mask=mask[:Bo], decoder_inputs_embeds=x_t[:Bo], time_ids=tids[:Bo])
    want_ids = oracle.classifier(want).argmax(-1)
    for precision, bar in (("fp16x2", 1e-3), ("fp16", 2e-3), ("bf16x3", 1e-4)):  # (a single call: the schedule's regimes one by one)
        model = DiDiSampler(sd, cfg, device="cuda", precision=precision)
        out, _ = model(encoder_input_ids=ids.cuda(), encoder_attention_mask=mask.cuda(), decoder_inputs_embeds=x_t.cuda(),
                       time_ids=tids.cuda())
        model.check_fault(f"bench shape {precision}")
        assert bool(torch.isfinite(out).all())
        hi, lo = ops.split(out.view(B * S, -1), dtype=model._pd)
        got_ids = ops.round_argmax(hi, lo if precision == "bf16x3" else None, model._cls.hi, model._cls.lo, precision).view(B, S)
        err = rel_l2(out[:Bo], want)
        match = (got_ids[:Bo].cpu() == want_ids).float().mean().item()
        # the ids of a random x_t are near-uniform: no token may dominate (a poisoned row would round to token 0)
        top_share = got_ids.flatten().bincount(minlength=cfg.vocab).max().item() / got_ids.numel()
        print(f"\n[bench shape {precision}] x0_hat rel-L2 {err:.3e}  id match {match:.4f} on {want_ids.numel()}  most frequent id share {top_share:.4f}")
        assert err <= bar and match >= 0.99 and top_share < 0.05
        del model


@pytest.mark.parametrize("precision", ["fp16", "fp16x2"])
def test_engine_options_at_the_bench_shape(precision):
    """One denoiser call at the bench shape under the engine's switches (didi_set_option): the attention fusions are exact, so
    DIDI_OPT_FUSED_ATTN = 0 / 1 / 2 / 3 give IDENTICAL x0_hat.  The residual stream's remainder plane as bytes
    (DIDI_OPT_LO8_RESIDUAL, default on) moves x0_hat by 4.7e-4 of its norm against the 16-bit plane, the hi-only stream
    (DIDI_OPT_HI_RESIDUAL, opt-in, one-product calls only) by 1.3e-3.  These are distances between two ROUNDED computations (any
    perturbation re-draws downstream fp16 roundings), not errors: against the reference the byte plane costs nothing measurable
    and the hi-only stream doubles the error (test_cfg2_size_vs_reference_fixture, profiles/r03_lo8_fixture.log,
    r03_hires_fixture.log) -- which is why the first is on and the second off."""
    from didi_amd import _lib
    from didi_amd.config import DiDiConfig
    from didi_amd.model import DiDiSampler
    from didi_amd.synthetic import synthetic_batch, synthetic_state_dict

    cfg = DiDiConfig(diffusion_steps=2000)
    sd = synthetic_state_dict(cfg, seed=1234)
    B, C, S = 64, 64, 64
    ids, mask = synthetic_batch(B, C, cfg.vocab, seed=0xBADB01)
    g = torch.Generator().manual_seed(5)
    x_t = torch.randn(B, S, cfg.dec_dim, generator=g)
    tids = torch.randint(1, 2001, (B, 1), generator=g)
    model = DiDiSampler(sd, cfg, device="cuda", precision=precision)

    def call():
        out, _ = model(encoder_input_ids=ids.cuda(), encoder_attention_mask=mask.cuda(), decoder_inputs_embeds=x_t.cuda(), time_ids=tids.cuda())
        model.check_fault("engine options")
        return out.clone()

    defaults = {k: _lib.get_option(k) for k in ("FUSED_ATTN", "LO8_RESIDUAL", "HI_RESIDUAL")}
    assert defaults == {"FUSED_ATTN": 3, "LO8_RESIDUAL": 1, "HI_RESIDUAL": 0}
    try:
        base = call()
        for bits in (0, 1, 2):
            _lib.set_option("FUSED_ATTN", bits)
            assert torch.equal(call(), base), f"FUSED_ATTN={bits}"
        _lib.set_option("FUSED_ATTN", 3)
        _lib.set_option("LO8_RESIDUAL", 0)
        full = call()
        d8 = rel_l2(base, full)
        _lib.set_option("HI_RESIDUAL", 1)
        dh = rel_l2(call(), full)
        print(f"\n[engine options {precision}] lo8 vs 16-bit remainder plane {d8:.2e}; hi-only vs 16-bit {dh:.2e}")
        if precision == "fp16":
            assert 0 < d8 <= 1e-3 and 2 * d8 < dh <= 1e-2
        else:
            assert d8 == 0.0 and dh == 0.0  # two-product calls keep their 16-bit remainder plane
        assert _lib.lib.didi_set_option(_lib.OPTIONS["FUSED_ATTN"], 4) == -1
    finally:
        for k, v in defaults.items():
            _lib.set_option(k, v)


def test_config2_shape_euler_f10_full_loop_and_batch_invariance():
    """BASELINE configs[2] at its per-GPU shape: euler (= DDIM, s_churn 0), step_freq 10 (201 calls), 512 dialogs x 64 (M = 32 768
    rows), the default precision schedule, the whole loop.  The noise is injected from device memory (202 draws x 101 MB: it fits
    in HBM, not in a test's host memory), so the first 40 dialogs can be re-run alone with the same draws: dialogs are
    independent and both batches take the fused 256x192 path, so their ids and final x0 must be IDENTICAL -- which ties this
    shape to test_cfg2_size_loop_ids_and_xt_vs_oracle[euler], where a 40-dialog batch is held to the oracle."""
    from didi_amd.config import DiDiConfig
    from didi_amd.model import DiDiSampler
    from didi_amd.sampling import sample
    from didi_amd.synthetic import synthetic_batch, synthetic_state_dict

    cfg = DiDiConfig(diffusion_steps=2000)
    sd = synthetic_state_dict(cfg, seed=1234)
    B, Bs, C, S, f = 512, 40, 64, 64, 10
    ids, mask = synthetic_batch(B, C, cfg.vocab, seed=0xBADB01)
    n_draws = len(range(cfg.diffusion_steps, 1, -f)) + 2
    g = torch.Generator(device="cuda").manual_seed(77)
    noise = torch.empty(n_draws, B, S, cfg.dec_dim, device="cuda")
    for i in range(n_draws):
        noise[i].normal_(generator=g)
    model = DiDiSampler(sd, cfg, device="cuda", precision="fp16-sched")
    got, lengths = sample(_ctx(ids, mask), model, "euler", f, max_len=S, noise=noise, return_ids=True)
    model.check_fault("config2 shape")
    x0 = next(iter(model._bufs.values()))["x0"].view(B, S, -1)
    assert tuple(got.shape) == (B, S) and bool(torch.isfinite(x0).all())
    assert int(got.min()) >= 0 and int(got.max()) < cfg.vocab and int(lengths.min()) >= 1
    top_share = got.flatten().bincount(minlength=cfg.vocab).max().item() / got.numel()
    assert top_share < 0.5
    got, x0 = got[:Bs].clone(), x0[:Bs].clone()
    small_noise = noise[:, :Bs].contiguous()
    del noise
    small = DiDiSampler(sd, cfg, device="cuda", precision="fp16-sched")
    got_s, _ = sample(_ctx(ids[:Bs], mask[:Bs]), small, "euler", f, max_len=S, noise=small_noise, return_ids=True)
    small.check_fault("config2 shape, 40-dialog slice")
    x0_s = next(iter(small._bufs.values()))["x0"].view(Bs, S, -1)
    print(f"\n[config2 shape] 512-dialog run vs its first 40 dialogs alone: x0 rel-L2 {rel_l2(x0, x0_s):.3e}, ids differing {int((got != got_s).sum())}")
    assert torch.equal(got, got_s) and torch.equal(x0, x0_s)


def test_config3_shape_large_vocab_clamp():
    """BASELINE configs[3] at length: V = 50 000, S = C = 84, B = 1024 (M = 86 016 rows), per-step clamp ON, T = 2000 with
    step_freq = 50: 41 denoiser calls, each followed by the rounding GEMM (6.6 TFLOP, the 17 GB of logits never written), the
    embedding gather and the sampler update.  The CPU oracle runs the first 2 dialogs (168 positions) with the same injected
    noise.  With the clamp every x0 is an exact embedding row, so a denoiser input x_t either agrees with the oracle's to
    rounding (same clamped id at the previous call) or is another token's embedding: the bars are, at EVERY call, >= 99 % of
    the rows within 1e-4 rel-L2 of the oracle's row, and final ids >= 99 %.  (An extension: the reference never clamps --
    parity unpinned by the reference, pinned on the oracle's restatement of the rule.)"""
    from didi_amd.config import DiDiConfig
    from didi_amd.model import DiDiSampler
    from didi_amd.sampling import sample
    from didi_amd.synthetic import synthetic_batch, synthetic_state_dict
    from oracle.model import OracleDiDi
    from oracle.sampling import Context, sample as oracle_sample

    cfg = DiDiConfig(vocab=50000, diffusion_steps=2000)
    sd = synthetic_state_dict(cfg, seed=4321)
    B, Bo, C, S, f = 1024, 2, 84, 84, 50
    ids, mask = synthetic_batch(B, C, cfg.vocab, seed=17)
    n_calls = len(range(cfg.diffusion_steps, 1, -f)) + 1
    assert n_calls == 41
    g = torch.Generator().manual_seed(33)
    small = torch.randn(n_calls, Bo, S, cfg.dec_dim, generator=g)  # the oracle's dialogs; the other 1022 draw on the device
    noise = torch.empty(n_calls, B, S, cfg.dec_dim, device="cuda")
    gd = torch.Generator(device="cuda").manual_seed(34)
    for i in range(n_calls):
        noise[i].normal_(generator=gd)
    noise[:, :Bo] = small.cuda()
    want_trace = []
    t0 = time.time()
    want = oracle_sample(Context(ids[:Bo], mask[:Bo]), OracleDiDi(sd, config_to_oracle_config(cfg)), "ddpm", f, max_len=S,
                         raw_output=True, noise_source=TickNoise([small[i] for i in range(n_calls)]), trace=want_trace, clamp=True)
    print(f"\n[config3 shape] oracle: {n_calls} calls x {Bo} dialogs, V = {cfg.vocab}, in {time.time() - t0:.0f} s")
    for precision in ("fp16-sched", "bf16x3"):
        model = DiDiSampler(sd, cfg, device="cuda", precision=precision)
        trace = []
        t0 = time.time()
        got, lengths = sample(_ctx(ids, mask), model, "ddpm", f, max_len=S, noise=noise, return_ids=True, trace=trace, clamp=True)
        model.check_fault("config3 shape")
        torch.cuda.synchronize()
        dt = time.time() - t0
        assert tuple(got.shape) == (B, S) and int(got.min()) >= 0 and int(got.max()) < cfg.vocab and len(trace) == n_calls
        match = (got[:Bo].cpu() == want.argmax(-1)).float().mean().item()
        share = []
        for a, b in zip(trace, want_trace):
            a, b = a[:Bo].cpu().double().view(Bo * S, -1), b.double().view(Bo * S, -1)
            share.append(float((((a - b).norm(dim=1) / b.norm(dim=1)) <= 1e-4).float().mean()))
        print(f"[config3 shape {precision}, clamp, {n_calls} calls x {B} dialogs in {dt:.1f} s] final id match {match:.4f} on {Bo * S} positions; "
              f"share of rows within 1e-4 of the oracle's x_t: min {min(share):.4f} (call {share.index(min(share))}), last {share[-1]:.4f}")
        assert share[0] == 1.0  # the initial noise, scaled
        assert min(share) >= 0.99 and match >= 0.99
        top_share = got.flatten().bincount(minlength=cfg.vocab).max().item() / got.numel()
        assert top_share < 0.05
        del model, trace
        torch.cuda.empty_cache()


def test_id_agreement_config1_2208_positions():
    """Token-id agreement over 2208 positions (config-1 model, 96 dialogs x 23, T = 50, identical injected noise, fp32 CPU
    oracle): bf16x3 and fp16 must hold >= 99.9 %; plain bf16 is reported (it does not: DESIGN.md section 2)."""
    from didi_amd.model import DiDiSampler
    from didi_amd.sampling import sample
    from oracle.model import OracleDiDi
    from oracle.sampling import Context, sample as oracle_sample

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
    for precision, bar in (("bf16x3", 0.999), ("fp16-sched", 0.999), ("fp16x2", 0.999), ("fp16", 0.999), ("bf16", 0.97)):
        model = DiDiSampler(sd, cfg, device="cuda", precision=precision)
        tr = []
        got, _ = sample(_ctx(ids, mask), model, "ddpm", 1, noise=torch.stack(draws[:n_calls]), return_ids=True, trace=tr)
        model.check_fault(precision)
        diff = int((got.cpu() != want).sum())
        worst = max(rel_l2(a, b) for a, b in zip(tr, trace))
        print(f"\n[id agreement {precision}] ids differ {diff} of {want.numel()} ({100 - 100.0 * diff / want.numel():.3f} % agree)  max x_t rel-L2 {worst:.3e}")
        assert 1.0 - diff / want.numel() >= bar


@pytest.mark.parametrize("precision", ["bf16x3", "fp16x2"])
def test_training_gradients_at_the_bench_shape_equal_the_transposing_route(precision, monkeypatch):
    """BASELINE configs[4]'s shape (12 x 768 decoder, 256 dialogs x 64 positions, dropout 0.1 / 0.1): every parameter gradient of one
    step through the transposed-source GEMM forms (didi_gemm_wgrad / didi_gemm_dgrad on the planes as they lie, the backward of
    dropout and GELU folded into the dC split) against the SAME step with those forms switched off (transposed copies through
    didi_gemm, separate elementwise passes): same products, different summation order.  The gradient fixtures from the reference are
    tiny models, which never take the transposed-source forms; this holds the composition at the shape bench.py times."""
    from didi_amd import ops
    from didi_amd.config import DiDiConfig
    from didi_amd.synthetic import synthetic_batch, synthetic_state_dict
    from didi_amd.trainer import DiDiTrainer

    cfg = DiDiConfig(diffusion_steps=2000)
    tr = DiDiTrainer(synthetic_state_dict(cfg, seed=1234), cfg, device="cuda", precision=precision, hidden_dropout=0.1, attention_dropout=0.1)
    ids, mask = synthetic_batch(256, 64, cfg.vocab, seed=0xBADB01)
    tids, tmask = synthetic_batch(256, 64, cfg.vocab, seed=0xBADB01 + 100)
    ctx, tgt = _ctx(ids, mask), _ctx(tids, tmask)
    used = {"wgrad": 0, "dgrad": 0}
    wgrad, dgrad = ops.gemm_wgrad, ops.gemm_dgrad

    def count(name, fn):
        def f(*a, **kw):
            used[name] += 1
            return fn(*a, **kw)
        return f

    monkeypatch.setattr(ops, "gemm_wgrad", count("wgrad", wgrad))
    monkeypatch.setattr(ops, "gemm_dgrad", count("dgrad", dgrad))
    torch.manual_seed(7)  # the step draws its noise seed and timesteps from torch's generator
    a = tr.forward_backward(ctx, tgt, dropout_seed=1234)
    assert used["wgrad"] >= 7 * cfg.dec_layers and used["dgrad"] >= 6 * cfg.dec_layers, used
    monkeypatch.setattr(ops, "wgrad_supported", lambda *a_: False)
    monkeypatch.setattr(ops, "dgrad_supported", lambda *a_: False)
    n_before = dict(used)
    torch.manual_seed(7)
    b = tr.forward_backward(ctx, tgt, dropout_seed=1234)
    assert used == n_before, "the reference pass must not touch the transposed-source forms"
    assert float(a["loss"]) == float(b["loss"])  # the forward is the same code either way
    worst = max((rel_l2(a["grads"][k], b["grads"][k]), k) for k in b["grads"])
    print(f"\n[configs4 gradients, {precision}] {len(b['grads'])} tensors, worst rel-L2 against the transposing route {worst[0]:.2e} ({worst[1]})")
    assert set(a["grads"]) == set(b["grads"]) and worst[0] <= 2e-5, worst
