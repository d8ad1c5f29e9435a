"""GPU parity of the sampler path (through the C ABI) against the golden fixtures and the CPU oracle.

Bars (BASELINE.json north_star): with identical weights and identical injected noise,
  * bf16x3 ("fp32-class") : every denoiser input x_t rel-L2 <= 1e-4, logits rel-L2 <= 1e-4,
                            token ids identical on >= 99.9 % of positions;
  * fp16x2 (bench default): fp16 planes, two products A_hi*(W_hi + W_lo): x_t rel-L2 <= 1e-3 (BASELINE.md bar "1e-3 in bf16"),
                            logits <= 2e-3, token ids >= 99.9 % -- the same id bar as bf16x3, held at the benchmarked model
                            size too (tests/test_gpu_fullsize.py);
  * fp16                  : one product on fp16 planes: the same bars on these SMALL models (measured ~1.2e-4); at the
                            benchmarked size it misses them (weight rounding adds up coherently over the loop), which is
                            why it is no longer the default;
  * bf16                  : x_t rel-L2 <= 1e-2 (north_star quotes 1e-3 "e.g."; measured values are
                            printed), ids reported, >= 97 % required on random-init weights whose
                            top-2 logit gaps are O(1e-1) (DESIGN.md "Precision").
"""
import pytest
import torch

from golden_util import (config1, config1_inputs, config1_state_dict, config_to_oracle_config, load_case, noise_list,
                         spec_to_config, weights_of, ListNoise)

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _ctx(ids, mask):
    from didi_amd.synthetic import RawContext

    return RawContext(ids.cuda(), mask.cuda())


def _model(weights, cfg, precision):
    from didi_amd.model import DiDiSampler

    return DiDiSampler(weights, cfg, device="cuda", precision=precision)


TINY_CASES = ["tiny_ddpm_f1", "tiny_ddpm_f3_len8", "tiny_euler_f2", "tiny_euler_churn_f4"]


@pytest.mark.parametrize("name", TINY_CASES + ["tiny_bertenc_ddpm_f2", "tiny_t5enc_ddpm_f2", "tiny_t5gated_euler_f3", "tiny_bbd80_ddpm_f2",
                                                "tiny_t5d128_ddpm_f2"])
@pytest.mark.parametrize("precision", ["bf16x3", "fp16-sched", "fp16x2", "fp16", "bf16"])
def test_golden_cases(name, precision):
    from didi_amd.sampling import sample

    t, meta = load_case(name)
    spec, run = meta["spec"], meta["run"]
    w = weights_of(t) if name.startswith(("tiny_bertenc", "tiny_t5", "tiny_bbd80")) else weights_of(load_case("tiny_weights")[0])
    model = _model(w, spec_to_config(spec), precision)
    noise = torch.stack(noise_list(t, int(meta["n_noise"])))
    ctx = _ctx(t["in.input_ids"], t["in.attention_mask"])
    # encoder + adapter
    context = model._encode_context(ctx.input_ids, ctx.attention_mask)
    assert rel_l2(context, t["out.context"]) <= 1e-4
    trace = []
    logits = sample(ctx, model, run["mode"], run["step_freq"], max_len=run["max_len"], raw_output=True, noise=noise, trace=trace)
    assert len(trace) == int(meta["n_calls"])
    worst = max(rel_l2(x, t[f"xt.{i}"]) for i, x in enumerate(trace))
    lerr = rel_l2(logits, t["out.logits"])
    ids, _ = sample(ctx, model, run["mode"], run["step_freq"], max_len=run["max_len"], noise=noise, return_ids=True)
    match = (ids.cpu() == t["out.ids"]).float().mean().item()
    print(f"\n[{name} {precision}] max x_t rel-L2 {worst:.3e}  logits rel-L2 {lerr:.3e}  id match {match:.4f}")
    if precision == "bf16x3":
        assert worst <= 1e-4 and lerr <= 1e-4
        assert match >= 0.999
    elif precision in ("fp16", "fp16x2", "fp16-sched"):  # fp16-sched: the default of bench.py, smoke() and the checkpoint loader
        assert worst <= 1e-3 and lerr <= 2e-3
        # >= 99.9 % of positions; the T5 cases have 48 positions, where ONE near-tie is 2 %: there an id may differ only
        # where the reference's own top-2 logits are closer than twice our largest logit deviation
        ref_top2 = t["out.logits"].topk(2, -1).values
        near_tie = (ref_top2[..., 0] - ref_top2[..., 1]) <= 2 * (logits.cpu() - t["out.logits"]).abs().max()
        assert match >= 0.999 or bool(((ids.cpu() == t["out.ids"]) | near_tie).all())
    else:
        assert worst <= 1e-2 and lerr <= 2e-2
        assert match >= 0.97
    # fused argmax == argmax of the materialised logits wherever the top-2 gap is not a near-tie
    top2 = logits.topk(2, -1).values
    clear = ((top2[..., 0] - top2[..., 1]) > 1e-4).cpu()
    assert torch.equal(ids.cpu()[clear], logits.argmax(-1).cpu()[clear])


@pytest.mark.parametrize("precision", ["bf16x3", "fp16x2", "fp16", "bf16"])
def test_config1_against_reference_fixture(precision):
    """BASELINE config 1 (V=8008, T=50, B=8, C=24): expected outputs come from the reference run."""
    from didi_amd.sampling import sample

    t, meta = load_case("config1_ddpm")
    model = _model(config1_state_dict(), config1(), precision)
    ids_in, mask = config1_inputs()
    shape = tuple(t["noise.0"].shape)
    torch.manual_seed(0xBADB01)
    draws = [torch.randn(shape)] + [torch.empty(shape).normal_(0, 1) for _ in range(int(meta["n_noise"]) - 1)]
    assert torch.equal(draws[0], t["noise.0"])
    trace = []
    ids, lengths = sample(_ctx(ids_in, mask), model, "ddpm", 1, noise=torch.stack(draws), trace=trace, return_ids=True)
    worst = max(rel_l2(trace[i], t[f"xt.{i}"]) for i in meta["kept_calls"])
    match = (ids.cpu() == t["out.ids"]).float().mean().item()
    print(f"\n[config1 {precision}] max x_t rel-L2 {worst:.3e}  id match {match:.4f}")
    assert worst <= {"bf16x3": 1e-4, "fp16x2": 1e-3, "fp16": 1e-3, "bf16": 1e-2}[precision]
    assert match >= (0.97 if precision == "bf16" else 0.999)
    assert lengths.cpu().tolist() == [ids.shape[1]] * ids.shape[0]  # no tokenizer -> no eos -> full rows


def test_reference_shaped_loop_drives_the_hip_model():
    """Drop-in check: the oracle's restatement of the reference Python loop (src/sampling.py) calls the
    HIP model object through the reference's own call surface (model(...), .emb, .sigmas, .classifier)."""
    from oracle.sampling import Context, sample as loop

    t, meta = load_case("tiny_ddpm_f1")
    model = _model(weights_of(load_case("tiny_weights")[0]), spec_to_config(meta["spec"]), "bf16x3")
    draws = [d.cuda() for d in noise_list(t, int(meta["n_noise"]))]
    trace = []
    logits = loop(Context(t["in.input_ids"].cuda(), t["in.attention_mask"].cuda()), model, "ddpm", 1, raw_output=True,
                  noise_source=ListNoise(draws), trace=trace)
    assert max(rel_l2(x, t[f"xt.{i}"]) for i, x in enumerate(trace)) <= 1e-4
    assert rel_l2(logits, t["out.logits"]) <= 1e-4
    with pytest.raises(ValueError):
        model(decoder_inputs_embeds=trace[0], time_ids=torch.ones(4, 1, dtype=torch.long))


@pytest.mark.parametrize("name", ["tiny_ddpm_f1", "tiny_euler_f2"])
def test_per_step_clamp_extension_matches_oracle(name):
    """BASELINE config 4's per-step clamp x0 <- emb[argmax(classifier(x0))] (an extension: the reference never
    clamps, so this is checked against the oracle's restatement of the same rule -- parity unpinned by the reference)."""
    from didi_amd.sampling import sample
    from golden_util import spec_to_oracle_config
    from oracle.model import OracleDiDi
    from oracle.sampling import Context, sample as oracle_sample

    t, meta = load_case(name)
    run = meta["run"]
    w = weights_of(load_case("tiny_weights")[0])
    draws = noise_list(t, int(meta["n_noise"]))
    want_trace = []
    want = oracle_sample(Context(t["in.input_ids"], t["in.attention_mask"]), OracleDiDi(w, spec_to_oracle_config(meta["spec"])),
                         run["mode"], run["step_freq"], max_len=run["max_len"], raw_output=True, noise_source=ListNoise(draws),
                         trace=want_trace, clamp=True)
    ctx = _ctx(t["in.input_ids"], t["in.attention_mask"])
    for precision, bar in (("bf16x3", 0.999), ("fp16", 0.97)):
        model = _model(w, spec_to_config(meta["spec"]), precision)
        trace = []
        ids, _ = sample(ctx, model, run["mode"], run["step_freq"], max_len=run["max_len"], noise=torch.stack(draws),
                        return_ids=True, trace=trace, clamp=True)
        match = (ids.cpu() == want.argmax(-1)).float().mean().item()
        worst = max(rel_l2(a, b) for a, b in zip(trace, want_trace))
        print(f"\n[clamp {name} {precision}] id match {match:.4f}  max x_t rel-L2 {worst:.3e}")
        assert match >= bar
        if precision == "bf16x3":
            assert worst <= 1e-4
        logits = sample(ctx, model, run["mode"], run["step_freq"], max_len=run["max_len"], noise=torch.stack(draws),
                        raw_output=True, clamp=True)
        assert torch.equal(logits.argmax(-1).cpu(), ids.cpu()) or precision != "bf16x3"


@pytest.mark.parametrize("B,C,max_len", [(1, 2, -1), (1, 1, 1), (3, 96, 96), (2, 96, -1), (5, 33, 7)])
def test_edge_shapes_against_oracle(B, C, max_len):
    """Smallest (one dialog, one target position), largest supported (S = C = 96) and ragged shapes, contexts with a
    single visible token, against the CPU oracle with the same synthetic weights and injected noise."""
    from didi_amd.config import DiDiConfig
    from didi_amd.sampling import _target_len, sample
    from didi_amd.synthetic import synthetic_state_dict
    from oracle.model import OracleDiDi
    from oracle.sampling import Context, sample as oracle_sample

    cfg = DiDiConfig(vocab=300, dec_dim=128, dec_layers=2, dec_heads=4, dec_ffn=256, dec_max_pos=128, enc_dim=96, enc_layers=1,
                     enc_heads=4, enc_ffn=192, enc_max_pos=128, diffusion_steps=6)
    sd = synthetic_state_dict(cfg, seed=77)
    g = torch.Generator().manual_seed(B * 100 + C)
    ids = torch.randint(3, cfg.vocab, (B, C), generator=g)
    mask = torch.ones(B, C, dtype=torch.long)
    mask[0, 1:] = 0  # one dialog whose context is a single token
    S = _target_len(C, max_len)
    draws = [torch.randn(B, S, cfg.dec_dim, generator=g) for _ in range(cfg.diffusion_steps + 1)]
    want = oracle_sample(Context(ids, mask), OracleDiDi(sd, config_to_oracle_config(cfg)), "ddpm", 1, max_len=max_len,
                         raw_output=True, noise_source=ListNoise(draws))
    n_calls = len(range(cfg.diffusion_steps, 1, -1)) + 1
    model = _model(sd, cfg, "bf16x3")
    got = sample(_ctx(ids, mask), model, "ddpm", 1, max_len=max_len, raw_output=True, noise=torch.stack(draws[:n_calls]))
    assert tuple(got.shape) == (B, S, cfg.vocab)
    assert rel_l2(got, want) <= 1e-4
    assert (got.argmax(-1).cpu() == want.argmax(-1)).float().mean().item() >= 0.999


def test_unsupported_lengths_fail_loudly():
    from didi_amd._lib import DidiHipError
    from didi_amd.config import DiDiConfig
    from didi_amd.sampling import sample
    from didi_amd.synthetic import synthetic_state_dict

    cfg = DiDiConfig(vocab=300, dec_dim=128, dec_layers=1, dec_heads=4, dec_ffn=256, dec_max_pos=128, enc_dim=96, enc_layers=1,
                     enc_heads=4, enc_ffn=192, enc_max_pos=128, diffusion_steps=4)
    model = _model(synthetic_state_dict(cfg, seed=78), cfg, "fp16")
    ids = torch.randint(3, cfg.vocab, (2, 100))
    with pytest.raises(DidiHipError):  # context of 100 tokens: the attention kernels are built for <= 96
        sample(_ctx(ids, torch.ones_like(ids)), model, "ddpm", 1, raw_output=True)
    with pytest.raises(ValueError):  # max_len = 0 leaves no target position
        sample(_ctx(ids[:, :8], torch.ones_like(ids[:, :8])), model, "ddpm", 1, max_len=0, raw_output=True)


def test_sample_errors_and_decode():
    from didi_amd.sampling import sample, truncate_predictions

    t, meta = load_case("tiny_ddpm_f3_len8")
    model = _model(weights_of(load_case("tiny_weights")[0]), spec_to_config(meta["spec"]), "bf16")
    ctx = _ctx(t["in.input_ids"], t["in.attention_mask"])
    with pytest.raises(NotImplementedError):
        sample(ctx, model, "ddim", 1)
    with pytest.raises(ValueError):
        sample(ctx, model, "ddpm", 3)

    class Tok:
        eos_token_id, sep_token_id = None, 17

        def batch_decode(self, rows, skip_special_tokens=True):
            return [" ".join(str(int(x)) for x in r) for r in rows]

    torch.manual_seed(5)
    a = sample(ctx, model, "ddpm", 3, tokenizer=Tok(), max_len=8)
    torch.manual_seed(5)
    b = sample(ctx, model, "ddpm", 3, tokenizer=Tok(), max_len=8)
    assert a == b and len(a) == 4, "Philox noise is keyed from torch's global generator: same seed, same replies"
    torch.manual_seed(6)
    c = sample(ctx, model, "ddpm", 3, tokenizer=Tok(), max_len=8, noise="torch")
    assert len(c) == 4
    pred = torch.tensor([[5, 17, 9, 17], [1, 2, 3, 4]]).cuda()
    rows = truncate_predictions(pred, 17)
    assert [r.tolist() for r in rows] == [[5, 17], [1, 2, 3, 4]]


class _FakeDictConfig(dict):
    """Stand-in for omegaconf.DictConfig (scripts/train.py:98 stores one as `opt_kwargs`): a class the weights-only loader rejects."""


def _decode(tokenizer, ids, skip_special_tokens=True):
    raise AssertionError("a checkpoint loader must never call into pickled callables")


@pytest.mark.parametrize("precision", [None, "bf16x3"])
def test_lightning_checkpoint_loader(tmp_path, precision):
    """SURVEY 8f row 1 against the REFERENCE's outputs: a .ckpt laid out as Lightning writes the reference's (state_dict +
    hyper_parameters holding an OmegaConf-like mapping and a functools.partial of a tokenizer method, model.py:91,
    scripts/train.py:89-99 -- a file torch.load(weights_only=True) refuses) is read by the tensors-only loader, and the sampler
    built from it reproduces what the reference sampler produced with those weights: every x_t and the ids of fixture
    tiny_ddpm_f3_len8 (generated by the imported src.sampling.sample).  precision None = the loader's default (fp16-sched)."""
    import functools
    import pickle

    from didi_amd.checkpoint import load_lightning_checkpoint
    from didi_amd.sampling import sample

    t, meta = load_case("tiny_ddpm_f3_len8")
    run = meta["run"]
    w = weights_of(load_case("tiny_weights")[0])
    cfg = spec_to_config(meta["spec"])
    hp = {"diffusion_steps": cfg.diffusion_steps, "schedule": cfg.schedule, "step_freq": 3, "sampling_mode": "ddpm",
          "opt_kwargs": _FakeDictConfig(lr=1e-4, scheduler=_FakeDictConfig(name="linear")), "batch_decoder": functools.partial(_decode, "tokenizer")}
    path = tmp_path / "step_1.ckpt"
    torch.save({"state_dict": w, "hyper_parameters": hp, "epoch": 0, "global_step": 1,
                "optimizer_states": [{"state": {0: {"exp_avg": torch.ones(3)}}}], "callbacks": {"x": _FakeDictConfig(a=1)}}, path)
    with pytest.raises(pickle.UnpicklingError):
        torch.load(path, map_location="cpu", weights_only=True)
    kw = {} if precision is None else {"precision": precision}
    loaded = load_lightning_checkpoint(path, device="cuda", dec_heads=cfg.dec_heads, enc_heads=cfg.enc_heads, **kw)
    assert loaded.cfg.step_freq == 3 and loaded.cfg.dec_layers == cfg.dec_layers and loaded.cfg.enc_kind == cfg.enc_kind
    if precision is None:  # "fp16-sched" = fp16 planes + the default schedule, tail floor and weight dither
        assert loaded.precision == "fp16x2" and loaded.schedule and loaded.tail_calls and loaded.weight_dither > 1
    ctx = _ctx(t["in.input_ids"], t["in.attention_mask"])
    noise = torch.stack(noise_list(t, int(meta["n_noise"])))
    trace = []
    ids, _ = sample(ctx, loaded, run["mode"], loaded.cfg.step_freq, max_len=run["max_len"], noise=noise, return_ids=True, trace=trace)
    loaded.check_fault("checkpoint")
    worst = max(rel_l2(x, t[f"xt.{i}"]) for i, x in enumerate(trace))
    match = (ids.cpu() == t["out.ids"]).float().mean().item()
    print(f"\n[checkpoint -> sampler, {precision or 'fp16-sched'}] max x_t rel-L2 vs the reference run {worst:.3e}  id match {match:.4f}")
    assert len(trace) == int(meta["n_calls"]) and worst <= (1e-4 if precision == "bf16x3" else 1e-3)
    assert torch.equal(ids.cpu(), t["out.ids"])


@pytest.mark.parametrize("name", ["tiny_ddpm_f1", "tiny_euler_f2"])
@pytest.mark.parametrize("precision", ["bf16x3", "fp16"])
def test_validation_step_against_reference_logits(name, precision):
    """DiDi.validation_step (model.py:245-254): CE over non-pad target tokens and token accuracy.  Expected values are
    computed from the REFERENCE run's logits in the fixture (calculate_batch_ce, metrics.py:91-93); ours come from the
    classifier GEMM's cross-entropy epilogue (no logits in memory)."""
    from didi_amd.synthetic import RawContext
    from didi_amd.validation import validation_step

    t, meta = load_case(name)
    run = meta["run"]
    model = _model(weights_of(load_case("tiny_weights")[0]), spec_to_config(meta["spec"]), precision)
    noise = torch.stack(noise_list(t, int(meta["n_noise"])))
    logits = t["out.logits"]
    B, S, V = logits.shape
    g = torch.Generator().manual_seed(77)
    tgt = torch.where(torch.rand(B, S, generator=g) < 0.5, logits.argmax(-1), torch.randint(0, V, (B, S), generator=g))
    lens = torch.randint(1, S + 1, (B,), generator=g)
    tmask = (torch.arange(S)[None, :] < lens[:, None]).long()
    keep = tmask.bool().view(-1)
    want_ce = torch.nn.functional.cross_entropy(logits.view(-1, V)[keep], tgt.view(-1)[keep]).item()
    want_acc = (((logits.argmax(-1) == tgt) * tmask).sum() / tmask.sum()).item()
    ce, acc = validation_step(model, _ctx(t["in.input_ids"], t["in.attention_mask"]), RawContext(tgt, tmask), run["mode"],
                              run["step_freq"], noise=noise)
    print(f"\n[{name} {precision}] ce {ce:.6f} (reference {want_ce:.6f})  acc {acc:.4f} (reference {want_acc:.4f})")
    assert abs(ce - want_ce) <= (1e-4 if precision == "bf16x3" else 2e-3) * max(1.0, abs(want_ce))
    assert abs(acc - want_acc) <= 1.0 / tmask.sum().item() + 1e-6  # at most one near-tie position
    with pytest.raises(ValueError):  # a target longer than the sampled length cannot be compared
        validation_step(model, _ctx(t["in.input_ids"], t["in.attention_mask"]),
                        RawContext(torch.zeros(B, 40, dtype=torch.long), torch.ones(B, 40, dtype=torch.long)), run["mode"], run["step_freq"])


def test_torch_noise_loop_equals_native_loop():
    """The per-call Python loop and the native didi_sample loop run the same kernels: with the same
    injected draws they must agree bit-for-bit."""
    from didi_amd.sampling import sample

    t, meta = load_case("tiny_euler_churn_f4")
    model = _model(weights_of(load_case("tiny_weights")[0]), spec_to_config(meta["spec"]), "bf16x3")
    ctx = _ctx(t["in.input_ids"], t["in.attention_mask"])
    n = int(meta["n_noise"])
    shape = tuple(t["noise.0"].shape)
    torch.manual_seed(11)
    torch.cuda.manual_seed(11)
    l_torch = sample(ctx, model, "euler", 4, raw_output=True, noise="torch")
    torch.cuda.manual_seed(11)
    draws = [torch.randn(shape, device="cuda")] + [torch.empty(shape, device="cuda").normal_(0, 1) for _ in range(n - 1)]
    l_native = sample(ctx, model, "euler", 4, raw_output=True, noise=torch.stack(draws))
    assert torch.equal(l_torch, l_native)


@pytest.mark.parametrize("name", ["tiny_ddpm_f1", "tiny_euler_churn_f4"])
def test_precision_schedule(name):
    """Precision schedule over the timestep of each denoiser call (fp16 planes: fp16 | fp16x2 | fp16x3 per call).
    (a) fp16x3 for every call is an fp32-class sampler: x_t within 1e-4 of the reference fixture, ids identical;
    (b) a schedule that names the plain precision everywhere is bit-identical to no schedule; (c) a schedule that switches at
    timestep t only changes calls from the first one at or below t on; (d) the per-call Python loop honours the same schedule
    bit for bit; (e) malformed schedules are rejected."""
    from didi_amd.model import DiDiSampler
    from didi_amd.sampling import sample
    from didi_amd.schedule import ddpm_plan, euler_plan

    t, meta = load_case(name)
    run, n_calls = meta["run"], int(meta["n_calls"])
    w = weights_of(load_case("tiny_weights")[0])
    cfg = spec_to_config(meta["spec"])
    noise = torch.stack(noise_list(t, int(meta["n_noise"])))
    ctx = _ctx(t["in.input_ids"], t["in.attention_mask"])

    def go(model, **kw):
        tr = []
        ids, _ = sample(ctx, model, run["mode"], run["step_freq"], max_len=run["max_len"], noise=noise, return_ids=True, trace=tr, **kw)
        model.check_fault(name)
        return ids.cpu(), tr

    full = DiDiSampler(w, cfg, device="cuda", precision="fp16", schedule=((0, "fp16x3"),))
    ids, tr = go(full)
    worst = max(rel_l2(x, t[f"xt.{i}"]) for i, x in enumerate(tr))
    print(f"\n[{name} fp16 planes, fp16x3 in all {n_calls} calls] max x_t rel-L2 {worst:.3e}  id match {(ids == t['out.ids']).float().mean().item():.4f}")
    assert worst <= 1e-4 and torch.equal(ids, t["out.ids"])
    plain_ids, plain_tr = go(DiDiSampler(w, cfg, device="cuda", precision="fp16x2"))
    same_ids, same_tr = go(DiDiSampler(w, cfg, device="cuda", precision="fp16x2", schedule=((0, "fp16x2"),)))
    assert all(torch.equal(a, b) for a, b in zip(plain_tr, same_tr))  # (the final rounding of a scheduled loop is fp16x3: ids may differ)
    m0 = DiDiSampler(w, cfg, device="cuda", precision="fp16x2")
    plan = (ddpm_plan(m0._sigmas_host, cfg.diffusion_steps, run["step_freq"]) if run["mode"] == "ddpm" else
            euler_plan(m0._sigmas_host, cfg.diffusion_steps, run["step_freq"], 1, cfg.s_churn, cfg.s_tmin, cfg.s_tmax))
    t_switch = sorted(plan.time_row)[1]  # the two lowest timesteps run in fp16x3
    part = DiDiSampler(w, cfg, device="cuda", precision="fp16x2", schedule=((t_switch, "fp16x2"), (0, "fp16x3")))
    precs = part.call_precisions(plan.time_row)
    k0 = next(k for k, p in enumerate(precs) if p == 5)
    assert 0 < k0 < n_calls and sum(p == 5 for p in precs) == 2
    _, part_tr = go(part)
    assert all(torch.equal(a, b) for a, b in zip(plain_tr[: k0 + 1], part_tr[: k0 + 1])), "calls before the switch are untouched"
    if k0 + 1 < n_calls:
        assert not torch.equal(plain_tr[k0 + 1], part_tr[k0 + 1]), "the first fp16x3 call changes the next input"
    # tail floors: the last n calls of a loop run at least the named precision, whatever their timestep ("fp16-sched" default: 64)
    from didi_amd.model import DEFAULT_TAIL_CALLS
    sched = DiDiSampler(w, cfg, device="cuda", precision="fp16-sched")
    assert sched.tail_calls == DEFAULT_TAIL_CALLS and part.tail_calls == ()
    rows = list(range(2000, 1, -10)) + [1]
    sched.schedule = ((700, "fp16"), (128, "fp16x2"), (0, "fp16x3"))  # thresholds at 0.35 T / 0.064 T of T = 2000
    precs = sched.call_precisions(rows)
    assert precs.count(5) == 64 and all(p == 5 for p in precs[-64:]) and precs.count(2) == 130 and precs.count(4) == 7
    # euler integrates an ODE (errors persist): its default schedule has no one-product regime; an explicit schedule serves both
    sched.schedule_euler = ((128, "fp16x2"), (0, "fp16x3"))
    precs = sched.call_precisions(rows, "euler")
    assert precs.count(2) == 0 and precs.count(5) == 64 and precs.count(4) == 137
    assert part.schedule_euler == part.schedule and part.call_precisions(plan.time_row, "euler") == part.call_precisions(plan.time_row)
    floor = DiDiSampler(w, cfg, device="cuda", precision="fp16x2", schedule=((0, "fp16x2"),), tail_calls=((1, "fp16x3"),))
    precs = floor.call_precisions(plan.time_row)
    assert precs[-1] == 5 and all(p == 4 for p in precs[:-1])
    with pytest.raises(ValueError):
        DiDiSampler(w, cfg, device="cuda", precision="fp16x2", tail_calls=((4, "fp16x3"),))
    for bad in (dict(precision="bf16x3", schedule=((0, "fp16x3"),)), dict(precision="fp16", schedule=((5, "fp16"),)),
                dict(precision="fp16", schedule=((5, "fp16"), (9, "fp16x2"), (0, "fp16x3")))):
        with pytest.raises(ValueError):
            DiDiSampler(w, cfg, device="cuda", **bad)
    # the Python loop (torch-drawn noise) follows the same schedule as the native loop
    shape = tuple(t["noise.0"].shape)
    torch.cuda.manual_seed(3)
    a = sample(ctx, part, run["mode"], run["step_freq"], max_len=run["max_len"], raw_output=True, noise="torch")
    torch.cuda.manual_seed(3)
    draws = [torch.randn(shape, device="cuda")] + [torch.empty(shape, device="cuda").normal_(0, 1) for _ in range(int(meta["n_noise"]) - 1)]
    b = sample(ctx, part, run["mode"], run["step_freq"], max_len=run["max_len"], raw_output=True, noise=torch.stack(draws))
    assert torch.equal(a, b)


def test_full_size_decoder_properties():
    """cfg2-sized model (H=768, 12 layers, 12 heads, FFN 3072; enc 1280/32 heads/5120), S=C=64.
    (a) a B=2 slice agrees with the CPU oracle; (b) dialogs are independent: the same dialogs give
    bit-identical output inside batches of 40 and 36 (both on the 256x192 GEMM tile with the fused
    projection+LayerNorm) and the same output to the split-plane rounding floor in a batch of 2 (128x128 tile,
    GEMM then LayerNorm: the statistics are summed in another order, which moves a few lo-plane roundings of
    2^-17 relative; measured 1.5e-5 after 12 layers): batch-split invariance."""
    from didi_amd.config import DiDiConfig
    from didi_amd.synthetic import synthetic_batch, synthetic_state_dict
    from oracle.model import OracleDiDi

    cfg = DiDiConfig(diffusion_steps=2000)
    sd = synthetic_state_dict(cfg, seed=1234)
    ids, mask = synthetic_batch(40, 64, cfg.vocab, seed=3)
    g = torch.Generator().manual_seed(9)
    x_t = torch.randn(40, 64, cfg.dec_dim, generator=g)
    tids = torch.randint(1, 2001, (40, 1), generator=g)
    model = _model(sd, cfg, "bf16x3")
    out, ctx = model(encoder_input_ids=ids.cuda(), encoder_attention_mask=mask.cuda(), decoder_inputs_embeds=x_t.cuda(), time_ids=tids.cuda())
    out2, ctx2 = model(encoder_input_ids=ids[:2].cuda(), encoder_attention_mask=mask[:2].cuda(),
                       decoder_inputs_embeds=x_t[:2].cuda(), time_ids=tids[:2].cuda())
    assert rel_l2(out[:2], out2) <= 5e-5 and rel_l2(ctx[:2], ctx2) <= 1e-6
    out36, ctx36 = model(encoder_input_ids=ids[:36].cuda(), encoder_attention_mask=mask[:36].cuda(),
                         decoder_inputs_embeds=x_t[:36].cuda(), time_ids=tids[:36].cuda())
    assert torch.equal(out[:36], out36) and torch.equal(ctx[:36], ctx36)
    out, ctx = out.clone(), ctx.clone()
    oracle = OracleDiDi(sd, config_to_oracle_config(cfg))
    want, want_ctx = oracle(encoder_input_ids=ids[:2], encoder_attention_mask=mask[:2], decoder_inputs_embeds=x_t[:2], time_ids=tids[:2])
    e_ctx, e_out = rel_l2(ctx2, want_ctx), rel_l2(out2, want)
    mb = _model(sd, cfg, "bf16")
    out_b, _ = mb(encoder_attention_mask=mask[:2].cuda(), decoder_inputs_embeds=x_t[:2].cuda(), time_ids=tids[:2].cuda(), context=ctx2)
    e_b = rel_l2(out_b, want)
    del mb
    mh = _model(sd, cfg, "fp16")
    out_h, _ = mh(encoder_attention_mask=mask[:2].cuda(), decoder_inputs_embeds=x_t[:2].cuda(), time_ids=tids[:2].cuda(), context=ctx2)
    e_h = rel_l2(out_h, want)
    out_h40, _ = mh(encoder_attention_mask=mask.cuda(), decoder_inputs_embeds=x_t.cuda(), time_ids=tids.cuda(), context=ctx)
    e_h40 = rel_l2(out_h40[:2], want)  # 256x192 tile path
    print(f"\n[full-size forward] context rel-L2 {e_ctx:.3e}  x0_hat rel-L2 bf16x3 {e_out:.3e}  fp16 {e_h:.3e} / {e_h40:.3e}  bf16 {e_b:.3e}")
    assert e_ctx <= 1e-4 and e_out <= 1e-4
    assert e_h <= 2e-3 and e_h40 <= 2e-3
    assert e_b <= 2e-2


def test_weight_dither_variants_and_loops():
    """``weight_dither`` = P: P fp16 roundings of every per-call GEMM weight whose MEAN is within ulp / 2P of the fp32 weight
    (plain rounding: ulp / 2), each within one ulp of it; only one-product calls use them (an fp16x3-only loop is bit-identical
    with and without), and the per-call Python loop picks the same variant per call as the native loop."""
    from didi_amd import ops
    from didi_amd.model import DiDiSampler
    from didi_amd.sampling import sample

    name = "tiny_ddpm_f1"
    t, meta = load_case(name)
    run = meta["run"]
    w = weights_of(load_case("tiny_weights")[0])
    cfg = spec_to_config(meta["spec"])
    noise = torch.stack(noise_list(t, int(meta["n_noise"])))
    ctx = _ctx(t["in.input_ids"], t["in.attention_mask"])
    P = 4
    m = DiDiSampler(w, cfg, device="cuda", precision="fp16", weight_dither=P)
    assert len(m._dec_variants) == P
    key = "decoder.encoder.layer.0.intermediate.dense.weight"
    W = w[key].double()
    planes = []
    for d, layers in m._dec_variants:
        n = W.numel()
        buf = (torch.empty(0, dtype=torch.float16, device="cuda"))
        # read the variant's ffn_up hi plane back through its raw pointer
        ptr = layers[0].ffn_up.w_hi
        hit = [x for x in m._keep if torch.is_tensor(x) and x.data_ptr() == ptr]
        assert len(hit) == 1 and hit[0].dtype == torch.float16 and hit[0].numel() == n
        planes.append(hit[0].double().cpu().view_as(W))
    _, e = torch.frexp(W.float())
    ulp = torch.ldexp(torch.ones_like(W.float()), (e - 11).clamp_min(-24)).double()
    mean_err = (torch.stack(planes).mean(0) - W).abs() / ulp
    rne_err = (W.half().double() - W).abs() / ulp
    assert float(mean_err.max()) <= 0.5 / P + 0.26 and float(mean_err.mean()) < 0.5 * float(rne_err.mean())
    assert all(float(((q - W).abs() / ulp).max()) <= 1.0 for q in planes)
    assert len({q.numpy().tobytes() for q in planes}) == P
    # native loop == Python loop, call by call, with variants in play
    tr_a, tr_b = [], []
    a, _ = sample(ctx, m, run["mode"], run["step_freq"], max_len=run["max_len"], noise=noise, return_ids=True, trace=tr_a)
    worst = max(rel_l2(x, t[f"xt.{i}"]) for i, x in enumerate(tr_a))
    assert worst <= 1e-3
    torch.cuda.manual_seed(3)
    shape = tuple(t["noise.0"].shape)
    x_native = sample(ctx, m, run["mode"], run["step_freq"], max_len=run["max_len"], raw_output=True,
                      noise=torch.stack([torch.randn(shape, device="cuda")] + [torch.empty(shape, device="cuda").normal_(0, 1) for _ in range(int(meta["n_noise"]) - 1)]))
    torch.cuda.manual_seed(3)
    x_python = sample(ctx, m, run["mode"], run["step_freq"], max_len=run["max_len"], raw_output=True, noise="torch")
    assert torch.equal(x_native, x_python)
    # variants are for one-product calls only
    full = DiDiSampler(w, cfg, device="cuda", precision="fp16", schedule=((0, "fp16x3"),))
    full_d = DiDiSampler(w, cfg, device="cuda", precision="fp16", schedule=((0, "fp16x3"),), weight_dither=P)
    ia, _ = sample(ctx, full, run["mode"], run["step_freq"], max_len=run["max_len"], noise=noise, return_ids=True, trace=tr_b)
    tr_c = []
    ib, _ = sample(ctx, full_d, run["mode"], run["step_freq"], max_len=run["max_len"], noise=noise, return_ids=True, trace=tr_c)
    assert torch.equal(ia, ib) and all(torch.equal(x, y) for x, y in zip(tr_b, tr_c))
    assert DiDiSampler(w, cfg, device="cuda", precision="bf16x3", weight_dither=P)._dec_variants == []
