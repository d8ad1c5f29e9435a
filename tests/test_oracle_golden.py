"""Pin the CPU oracle to the fixtures generated from the reference sampler (oracle/make_golden.py).

Bars (fp32 CPU vs fp32 CPU, different but equivalent op order): every denoiser input x_t of every
call rel-L2 <= 1e-5, final logits rel-L2 <= 1e-5, token ids identical, schedules bit-identical.
"""
import pytest
import torch

from golden_util import (ListNoise, config1_inputs, config1_state_dict, config_to_oracle_config, config1, load_case,
                         noise_list, sd_hash, spec_to_oracle_config, weights_of)
from oracle import diffusion_math
from oracle.model import OracleDiDi
from oracle.sampling import Context, sample


def rel_l2(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("sched", ["sqrt", "cosine", "linear"])
@pytest.mark.parametrize("T", [50, 2000])
def test_schedule_bit_exact(sched, T):
    t, _ = load_case("schedules")
    sigmas, std0 = diffusion_math.configure_schedule(T, sched)
    assert torch.equal(sigmas, t[f"{sched}.{T}.sigmas"])
    assert torch.equal(std0.reshape(1), t[f"{sched}.{T}.std_0"])


def test_schedule_known_values():
    # values observed from the reference in the build container
    sigmas, std0 = diffusion_math.configure_schedule(2000, "sqrt")
    assert abs(sigmas[1].item() - 0.121897) < 1e-6 and abs(sigmas[2].item() - 0.154793) < 1e-6
    assert abs(sigmas[-1].item() - 7034.453) < 1e-2 and abs(std0.item() - 0.121001) < 1e-6
    assert sigmas[0].item() == 0.0


def _run_case(name, weights=None):
    t, meta = load_case(name)
    spec, run = meta["spec"], meta["run"]
    w = weights if weights is not None else weights_of(t)
    model = OracleDiDi(w, spec_to_oracle_config(spec))
    trace = []
    logits = sample(Context(t["in.input_ids"], t["in.attention_mask"]), model, run["mode"], run["step_freq"],
                    max_len=run["max_len"], raw_output=True,
                    noise_source=ListNoise(noise_list(t, int(meta["n_noise"]))), trace=trace)
    return t, meta, model, trace, logits


@pytest.mark.parametrize("name", ["tiny_ddpm_f1", "tiny_ddpm_f3_len8", "tiny_euler_f2", "tiny_euler_churn_f4"])
def test_tiny_cases(name):
    w, _ = load_case("tiny_weights")
    t, meta, model, trace, logits = _run_case(name, weights_of(w))
    assert len(trace) == int(meta["n_calls"])
    for i, x in enumerate(trace):
        assert rel_l2(x, t[f"xt.{i}"]) <= 1e-5, f"x_t of call {i}"
    assert rel_l2(logits, t["out.logits"]) <= 1e-5
    assert torch.equal(logits.argmax(-1), t["out.ids"])
    ctx = model.encode_context(t["in.input_ids"], t["in.attention_mask"])
    assert rel_l2(ctx, t["out.context"]) <= 1e-5


def test_bert_encoder_case():
    t, meta, model, trace, logits = _run_case("tiny_bertenc_ddpm_f2")
    assert not model.has_adapter
    for i, x in enumerate(trace):
        assert rel_l2(x, t[f"xt.{i}"]) <= 1e-5
    assert rel_l2(logits, t["out.logits"]) <= 1e-5
    assert torch.equal(logits.argmax(-1), t["out.ids"])


def test_wide_head_blenderbot_encoder_case():
    """A Blenderbot context encoder with 80-wide heads (facebook/blenderbot-3B's head dim; configs/train.yaml:3): fixture from
    HF's BlenderbotEncoder under the reference sampler (oracle/make_golden.py wide)."""
    t, meta, model, trace, logits = _run_case("tiny_bbd80_ddpm_f2")
    assert meta["spec"]["enc_dim"] // meta["spec"]["enc_heads"] == 80 and model.has_adapter
    ctx = model.encode_context(t["in.input_ids"], t["in.attention_mask"])
    assert rel_l2(ctx, t["out.context"]) <= 1e-5
    for i, x in enumerate(trace):
        assert rel_l2(x, t[f"xt.{i}"]) <= 1e-5
    assert rel_l2(logits, t["out.logits"]) <= 1e-5
    assert torch.equal(logits.argmax(-1), t["out.ids"])


@pytest.mark.parametrize("name,adapter", [("tiny_t5enc_ddpm_f2", True), ("tiny_t5gated_euler_f3", False), ("tiny_t5d128_ddpm_f2", True)])
def test_t5_encoder_cases(name, adapter):
    """T5EncoderModel contexts (model.py:39-42), ReLU and gated-GELU FFN, 40-token contexts (log-spaced position buckets):
    the fixture comes from HF's T5EncoderModel under the reference sampler."""
    t, meta, model, trace, logits = _run_case(name)
    assert model.has_adapter == adapter
    ctx = model.encode_context(t["in.input_ids"], t["in.attention_mask"])
    assert rel_l2(ctx, t["out.context"]) <= 1e-5
    for i, x in enumerate(trace):
        assert rel_l2(x, t[f"xt.{i}"]) <= 1e-5
    assert rel_l2(logits, t["out.logits"]) <= 1e-5
    assert torch.equal(logits.argmax(-1), t["out.ids"])


def test_config1_regenerated_weights():
    """BASELINE config 1: weights/noise regenerated from seeds, expected outputs from the reference run."""
    t, meta = load_case("config1_ddpm")
    w = config1_state_dict()
    assert sd_hash(w) == meta["weights_sha256"], "seeded weight generator drifted: regenerate the fixture"
    ids, mask = config1_inputs()
    assert torch.equal(ids, t["in.input_ids"]) and torch.equal(mask, t["in.attention_mask"])
    model = OracleDiDi(w, config_to_oracle_config(config1()))
    torch.manual_seed(0xBADB01)
    shape = tuple(t["noise.0"].shape)
    draws = [torch.randn(shape)] + [torch.empty(shape).normal_(0, 1) for _ in range(int(meta["n_noise"]) - 1)]
    assert torch.equal(draws[0], t["noise.0"])
    trace = []
    logits = sample(Context(ids, mask), model, "ddpm", 1, raw_output=True, noise_source=ListNoise(draws), trace=trace)
    for i in meta["kept_calls"]:
        assert rel_l2(trace[i], t[f"xt.{i}"]) <= 1e-5
    assert rel_l2(logits[0], t["out.logits_b0"]) <= 1e-5
    assert torch.equal(logits.argmax(-1), t["out.ids"])


def test_sample_errors_match_reference():
    w, _ = load_case("tiny_weights")
    t, meta = load_case("tiny_ddpm_f1")
    model = OracleDiDi(weights_of(w), spec_to_oracle_config(meta["spec"]))
    ctx = Context(t["in.input_ids"], t["in.attention_mask"])
    with pytest.raises(NotImplementedError):  # src/sampling.py:24
        sample(ctx, model, "ddim", 1, raw_output=True)
    with pytest.raises(ValueError):  # src/sampling.py:30
        sample(ctx, model, "ddpm", 5)
    with pytest.raises(ValueError):  # src/diffusion/model.py:180-181
        model(decoder_inputs_embeds=t["xt.0"], time_ids=torch.ones(4, 1, dtype=torch.long))


@pytest.mark.parametrize("name", ["train_tiny_plain", "train_tiny_anchor_nf2"])
def test_training_step_oracle_matches_reference_fixture(name):
    """oracle/training.py against DiDi.training_step run on the HF modules (oracle/make_golden.py train): x_t, x_0_hat, the five
    logged scalars and EVERY trainable parameter's gradient (63 tensors), with the captured draws and t injected."""
    from oracle.training import training_step_with_grads

    t, meta = load_case(name)
    run = meta["run"]
    w = weights_of(load_case("tiny_weights")[0])
    draws = noise_list(t, int(meta["n_noise"]))
    out, grads = training_step_with_grads(w, spec_to_oracle_config(meta["spec"]), Context(t["in.input_ids"], t["in.attention_mask"]),
                                          Context(t["in.target_ids"], t["in.target_mask"]), draws, t["t"],
                                          anchor_loss=run["anchor_loss"], noise_factor=run["noise_factor"])
    assert bool((t["t"] == 1).any()) and not bool((t["t"] == 1).all()), "the fixture exercises both branches of the mse `where`"
    assert rel_l2(out["x_t"], t["out.x_t"]) <= 1e-6 and rel_l2(out["x_0_hat"].detach(), t["out.x_0_hat"]) <= 1e-5
    got = torch.stack([out[k].detach() for k in ("mse", "ce", "t_T", "loss", "ce_hat")])
    assert torch.allclose(got, t["out.scalars"], rtol=1e-5, atol=1e-9), (got, t["out.scalars"])
    want = {k[2:]: v for k, v in t.items() if k.startswith("g.")}
    assert len(want) == int(meta["n_grads"]) == 63
    # HF keeps `classifier` / `emb` etc. under the same names; the decoder's unused word embeddings and pooler get no gradient
    for k, g in want.items():
        assert k in grads, k
        assert rel_l2(grads[k], g) <= 2e-5 or (g.norm() < 1e-12 and grads[k].norm() < 1e-9), (k, rel_l2(grads[k], g))


@pytest.mark.parametrize("case", ["cfg2_ddpm_f1", "cfg2_ddpm_f1_trained_like", "cfg2_euler_f10"])
def test_oracle_against_the_full_size_reference_fixture(case):
    """The oracle at the BENCHMARKED model size (12 x 768 decoder, 2 x 1280 encoder): its first three denoiser inputs for two
    dialogs against what the reference sampler + HF modules produced (tests/golden/cfg2_*, oracle/make_golden.py fullsize), on
    the same Philox-fed noise.  x_t of call 2 has been through the encoder, the adapter, the hoisted cross-attention and two
    whole decoder passes; the fixture's remaining ~2000 calls are the GPU test's (tests/test_gpu_fullsize.py)."""
    import numpy as np

    from didi_amd.config import DiDiConfig
    from didi_amd.synthetic import synthetic_batch, synthetic_state_dict
    from golden_util import sd_hash, trained_like
    from oracle.philox import normal

    t, meta = load_case(case)
    run = meta["run"]
    cfg = DiDiConfig(diffusion_steps=meta["spec"]["T"])
    sd = synthetic_state_dict(cfg, seed=run["weight_seed"])
    if run["weights"] == "trained_like":
        sd = trained_like(sd)
    # tensors whose last bits vary with the host CPU's libm (the schedule buffers; trained_like's exp() on the LayerNorm gammas)
    # travel in the fixture, as they would in a checkpoint
    sd.update({k[2:]: (v.reshape(()) if k == "w.std_0" else v) for k, v in t.items() if k.startswith("w.")})
    assert sd_hash(sd) == meta["weights_sha256"]
    ids, mask = synthetic_batch(run["B_gpu"], 64, cfg.vocab, seed=run["batch_seed"])
    assert torch.equal(ids, t["in.input_ids"]) and torch.equal(mask, t["in.attention_mask"])
    Bo, S, H = 2, run["max_len"], cfg.dec_dim
    per_draw = run["B_gpu"] * S * H // 4

    class Stop(Exception):
        pass

    class Feed:
        i = 0

        def __call__(self, shape):
            if self.i == 4:
                raise Stop
            z = torch.from_numpy(normal(int(np.prod(shape)), run["philox_seed"], self.i * per_draw)).view(*shape)
            self.i += 1
            return z

    trace = []
    with pytest.raises(Stop):
        sample(Context(ids[:Bo], mask[:Bo]), OracleDiDi(sd, config_to_oracle_config(cfg)), run["mode"], run["step_freq"], max_len=S,
               raw_output=True, noise_source=Feed(), trace=trace)
    assert len(trace) >= 3
    errs = [rel_l2(trace[k], t[f"xt.{k}"][:Bo]) for k in range(3)]
    print(f"\n[{case}] oracle vs reference x_t rel-L2, calls 0-2: " + " ".join(f"{e:.2e}" for e in errs))
    assert max(errs) <= 1e-5
