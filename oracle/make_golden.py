"""Generate tests/golden/*.safetensors by running the REFERENCE sampler in the build container.

Run here only (needs /root/reference and `transformers`; neither exists on the GPU box):

    python -m oracle.make_golden            # from the repo root

What runs, and whose code it is:
  * ``src.sampling.sample`` and ``src.diffusion.utils.configure_schedule`` are imported from
    /root/reference and run unmodified (they import only torch).
  * the decoder is ``transformers.BertModel(BertConfig(is_decoder=True, add_cross_attention=True))``
    and the encoder ``transformers`` ``BlenderbotEncoder`` / ``BertModel`` built from config, as
    reference ``get_components`` does minus the hub download (src/diffusion/model.py:29-54).
    ``attn_implementation="eager"`` is chosen because the reference pins transformers 4.30.2,
    whose only path was eager; the sdpa path is also run and its deviation printed.
  * ``src.diffusion.model`` itself is not importable here (ModuleNotFoundError: lightning), so
    ``RefShapedDiDi`` below restates DiDi.__init__'s sampler-visible attributes (model.py:104-127)
    and DiDi.forward (model.py:165-198) as a plain nn.Module around those HF modules.
  * the noise stream is CAPTURED from the reference run by recording what ``torch.randn_like``
    and ``Tensor.normal_`` return while ``sample`` runs.

Every fixture is data only: weights, inputs, the captured noise, the denoiser inputs x_t of every
call, the returned context, final logits and ids.  TEST INFRASTRUCTURE: see oracle/__init__.py.
"""
import hashlib
import json
import os
import sys

import torch
from safetensors.torch import save_file
from torch import nn

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class RefShapedDiDi(nn.Module):
    def __init__(self, encoder, decoder, enc_dim, dec_dim, vocab, *, diffusion_steps, schedule, pad_idx=0,
                 tie_weights=False, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf")):
        super().__init__()
        from src.diffusion.utils import configure_schedule  # the reference's own schedule code

        self.diffusion_steps = diffusion_steps
        self.encoder_dim, self.decoder_dim = enc_dim, dec_dim
        self.emb = nn.Embedding(vocab, dec_dim, padding_idx=pad_idx)
        self.time_embeds = nn.Embedding(diffusion_steps + 1, dec_dim)
        self.s_churn, self.s_tmin, self.s_tmax = s_churn, s_tmin, s_tmax
        self.encoder, self.decoder = encoder, decoder
        self.classifier = nn.Linear(dec_dim, vocab, bias=False)
        if tie_weights:
            self.classifier.weight = self.emb.weight
        if enc_dim != dec_dim:
            self.adapter = nn.Sequential(nn.Linear(enc_dim, dec_dim), nn.Tanh(), nn.Linear(dec_dim, dec_dim))
        sigmas, std_0 = configure_schedule(diffusion_steps, schedule)
        self.register_buffer("sigmas", sigmas)
        self.register_buffer("std_0", std_0)
        self.calls = []  # decoder_inputs_embeds of every call, in order
        self.keep_calls = None  # a set of call indices: record only those (full-size runs: 2000 calls x 3 MB otherwise)
        self.n_calls = 0

    def forward(self, encoder_input_ids=None, encoder_attention_mask=None, decoder_inputs_embeds=None,
                time_ids=None, context=None):
        if encoder_input_ids is None and context is None:
            raise ValueError("Either `encoder_input_ids` or `context` must be provided.")
        if context is None:
            context = self.encoder(input_ids=encoder_input_ids, attention_mask=encoder_attention_mask).last_hidden_state
            if self.encoder_dim != self.decoder_dim:
                context = self.adapter(context)
        if self.keep_calls is None or self.n_calls in self.keep_calls:
            self.calls.append(decoder_inputs_embeds.detach().clone())
        self.n_calls += 1
        x = decoder_inputs_embeds + self.time_embeds(time_ids)
        out = self.decoder(inputs_embeds=x, encoder_hidden_states=context,
                           encoder_attention_mask=encoder_attention_mask).last_hidden_state
        return out, context


class Ctx:
    def __init__(self, ids, mask):
        self.input_ids, self.attention_mask = ids, mask


class NoiseTap:
    """Record every tensor the global generator hands to the reference sampler."""

    def __enter__(self):
        self.draws = []
        self._randn_like, self._normal_ = torch.randn_like, torch.Tensor.normal_
        tap = self

        def randn_like(x, *a, **k):
            r = tap._randn_like(x, *a, **k)
            tap.draws.append(r.detach().clone())
            return r

        def normal_(self_t, *a, **k):
            r = tap._normal_(self_t, *a, **k)
            tap.draws.append(self_t.detach().clone())
            return r

        torch.randn_like, torch.Tensor.normal_ = randn_like, normal_
        return self

    def __exit__(self, *exc):
        torch.randn_like, torch.Tensor.normal_ = self._randn_like, self._normal_


def build_model(spec, seed, attn="eager"):
    from transformers import BertConfig, BertModel, BlenderbotConfig
    from transformers.models.blenderbot.modeling_blenderbot import BlenderbotEncoder

    torch.manual_seed(seed)
    V = spec["vocab"]
    if spec["enc_kind"] == "blenderbot":
        ecfg = BlenderbotConfig(vocab_size=V, d_model=spec["enc_dim"], encoder_layers=spec["enc_layers"],
                                encoder_attention_heads=spec["enc_heads"], encoder_ffn_dim=spec["enc_ffn"],
                                decoder_layers=1, decoder_attention_heads=spec["enc_heads"], decoder_ffn_dim=8,
                                max_position_embeddings=128, scale_embedding=True, activation_function="gelu",
                                dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, pad_token_id=0,
                                attn_implementation=attn)
        encoder = BlenderbotEncoder(ecfg)
    elif spec["enc_kind"] == "t5":
        from transformers import T5Config, T5EncoderModel

        ecfg = T5Config(vocab_size=V, d_model=spec["enc_dim"], d_kv=spec["enc_dkv"], d_ff=spec["enc_ffn"],
                        num_layers=spec["enc_layers"], num_heads=spec["enc_heads"], dropout_rate=0.0,
                        feed_forward_proj=spec["enc_ffn_act"], relative_attention_num_buckets=32,
                        relative_attention_max_distance=128, is_encoder_decoder=False, use_cache=False, pad_token_id=0,
                        attn_implementation=attn)
        encoder = T5EncoderModel(ecfg)  # model.py:40-42
    else:
        ecfg = BertConfig(vocab_size=V, hidden_size=spec["enc_dim"], num_hidden_layers=spec["enc_layers"],
                          num_attention_heads=spec["enc_heads"], intermediate_size=spec["enc_ffn"],
                          hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, attn_implementation=attn)
        encoder = BertModel(ecfg)
    dcfg = BertConfig(vocab_size=V, is_decoder=True, add_cross_attention=True, hidden_size=spec["dec_dim"],
                      num_hidden_layers=spec["dec_layers"], num_attention_heads=spec["dec_heads"],
                      intermediate_size=spec["dec_ffn"], hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                      attn_implementation=attn)
    decoder = BertModel(dcfg)
    model = RefShapedDiDi(encoder, decoder, spec["enc_dim"], spec["dec_dim"], V,
                          diffusion_steps=spec["T"], schedule=spec["schedule"],
                          s_churn=spec.get("s_churn", 0.0), s_tmin=spec.get("s_tmin", 0.0),
                          s_tmax=spec.get("s_tmax", float("inf")))
    # LayerNorm/bias perturbation so that gamma/beta/bias paths are actually exercised
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("LayerNorm.weight") or n.endswith("layer_norm.weight"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif n.endswith(".bias"):
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    return model.eval()


def make_inputs(B, C, V, seed, min_len=2):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, V, (B, C), generator=g)
    lens = torch.randint(min_len, C + 1, (B,), generator=g)
    lens[0] = C  # one full row
    mask = (torch.arange(C)[None, :] < lens[:, None]).long()
    ids = ids * mask  # pad id 0, right padding (commonsense_dataset.py:27-32 pads to max_length)
    return ids, mask


def run_case(model, ids, mask, mode, step_freq, max_len, noise_seed, noise_factor=1):
    from src.sampling import sample  # the reference sampler, unmodified

    model.calls.clear()
    torch.manual_seed(noise_seed)
    with NoiseTap() as tap:
        logits = sample(Ctx(ids, mask), model, mode, step_freq, noise_factor=noise_factor,
                        max_len=max_len, raw_output=True)
    return logits, tap.draws, list(model.calls)


def sd_hash(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].contiguous().numpy().tobytes())
    return h.hexdigest()


TINY = dict(vocab=600, enc_kind="blenderbot", enc_dim=96, enc_layers=2, enc_heads=4, enc_ffn=256,
            dec_dim=128, dec_layers=2, dec_heads=4, dec_ffn=512, T=20, schedule="sqrt")
TINY_BERT = dict(vocab=600, enc_kind="bert", enc_dim=128, enc_layers=2, enc_heads=4, enc_ffn=256,
                 dec_dim=128, dec_layers=2, dec_heads=4, dec_ffn=512, T=10, schedule="cosine")
TINY_T5 = dict(vocab=600, enc_kind="t5", enc_dim=96, enc_dkv=16, enc_layers=2, enc_heads=4, enc_ffn=256, enc_ffn_act="relu",
               dec_dim=128, dec_layers=2, dec_heads=4, dec_ffn=512, T=10, schedule="sqrt")
TINY_T5_GATED = dict(vocab=600, enc_kind="t5", enc_dim=128, enc_dkv=32, enc_layers=2, enc_heads=4, enc_ffn=192,
                     enc_ffn_act="gated-gelu", dec_dim=128, dec_layers=2, dec_heads=4, dec_ffn=512, T=10, schedule="linear")
CONFIG1 = dict(vocab=8008, enc_kind="blenderbot", enc_dim=128, enc_layers=2, enc_heads=4, enc_ffn=512,
               dec_dim=128, dec_layers=2, dec_heads=4, dec_ffn=512, T=50, schedule="sqrt")


def state_dict_of(model):
    return {k: v.detach().clone().contiguous() for k, v in model.state_dict().items()}


def save_case(name, spec, run, logits, draws, calls, ctx, ids, mask, extra_meta=None, weights=None):
    t = {"in.input_ids": ids, "in.attention_mask": mask, "out.logits": logits.contiguous(),
         "out.ids": logits.argmax(-1), "out.context": ctx.contiguous()}
    for i, d in enumerate(draws):
        t[f"noise.{i}"] = d.contiguous()
    for i, c in enumerate(calls):
        t[f"xt.{i}"] = c.contiguous()
    if weights is not None:
        for k, v in weights.items():
            t["w." + k] = v
    meta = {"spec": json.dumps(spec), "run": json.dumps(run), "n_noise": str(len(draws)), "n_calls": str(len(calls))}
    meta.update(extra_meta or {})
    save_file(t, os.path.join(OUT, name + ".safetensors"), metadata=meta)
    print(f"wrote {name}: {len(draws)} draws, {len(calls)} calls, logits {tuple(logits.shape)}")


def t5_cases():
    """(H) T5 context encoders (model.py:39-42): v1.0 ReLU FFN with an adapter, v1.1 gated-GELU FFN without one.
    40-token contexts reach the logarithmic relative-position buckets."""
    for name, spec, seed in (("tiny_t5enc_ddpm_f2", TINY_T5, 13), ("tiny_t5gated_euler_f3", TINY_T5_GATED, 14)):
        m = build_model(spec, seed=seed)
        with torch.no_grad():  # HF zero-inits nothing here, but the bias table starts tiny: widen it so it matters
            m.encoder.encoder.block[0].layer[0].SelfAttention.relative_attention_bias.weight.mul_(4.0)
        ids, mask = make_inputs(3, 40, spec["vocab"], seed=seed + 20)
        run = dict(mode="ddpm" if "ddpm" in name else "euler", step_freq=2 if "ddpm" in name else 3, max_len=16, noise_seed=200 + seed)
        logits, draws, calls = run_case(m, ids, mask, run["mode"], run["step_freq"], run["max_len"], run["noise_seed"])
        ctx = m(encoder_input_ids=ids, encoder_attention_mask=mask, decoder_inputs_embeds=calls[0],
                time_ids=torch.ones(3, 1, dtype=torch.long))[1]
        save_case(name, spec, run, logits, draws, calls, ctx.detach(), ids, mask, weights=state_dict_of(m))


TINY_BB_D80 = dict(vocab=600, enc_kind="blenderbot", enc_dim=160, enc_layers=2, enc_heads=2, enc_ffn=320,
                   dec_dim=128, dec_layers=2, dec_heads=4, dec_ffn=512, T=10, schedule="sqrt")


TINY_T5_D128 = dict(vocab=600, enc_kind="t5", enc_dim=96, enc_dkv=128, enc_layers=2, enc_heads=2, enc_ffn=256, enc_ffn_act="relu",
                    dec_dim=128, dec_layers=2, dec_heads=4, dec_ffn=512, T=10, schedule="sqrt")


def wide_t5_case():
    """(W2) A T5 context encoder with d_kv = 128 (t5-3b / t5-11b), 40-token contexts, position bias widened as in t5_cases."""
    name, spec, seed = "tiny_t5d128_ddpm_f2", TINY_T5_D128, 16
    m = build_model(spec, seed=seed)
    with torch.no_grad():
        m.encoder.encoder.block[0].layer[0].SelfAttention.relative_attention_bias.weight.mul_(4.0)
    ids, mask = make_inputs(3, 40, spec["vocab"], seed=seed + 20)
    run = dict(mode="ddpm", step_freq=2, max_len=16, noise_seed=200 + seed)
    logits, draws, calls = run_case(m, ids, mask, run["mode"], run["step_freq"], run["max_len"], run["noise_seed"])
    ctx = m(encoder_input_ids=ids, encoder_attention_mask=mask, decoder_inputs_embeds=calls[0], time_ids=torch.ones(3, 1, dtype=torch.long))[1]
    save_case(name, spec, run, logits, draws, calls, ctx.detach(), ids, mask, weights=state_dict_of(m))


def wide_head_cases():
    """(W) A Blenderbot context encoder with 80-wide attention heads, the head dim of facebook/blenderbot-3B (d_model 2560 / 32
    heads), which configs/train.yaml:3 lists as a base model: past the MFMA attention kernel's head dim of 64."""
    name, spec, seed = "tiny_bbd80_ddpm_f2", TINY_BB_D80, 15
    m = build_model(spec, seed=seed)
    ids, mask = make_inputs(3, 20, spec["vocab"], seed=seed + 20)
    run = dict(mode="ddpm", step_freq=2, max_len=12, noise_seed=200 + seed)
    logits, draws, calls = run_case(m, ids, mask, run["mode"], run["step_freq"], run["max_len"], run["noise_seed"])
    ctx = m(encoder_input_ids=ids, encoder_attention_mask=mask, decoder_inputs_embeds=calls[0], time_ids=torch.ones(3, 1, dtype=torch.long))[1]
    save_case(name, spec, run, logits, draws, calls, ctx.detach(), ids, mask, weights=state_dict_of(m))


def ref_training_step(model, ctx, tgt, anchor_loss, noise_factor, tap):
    """DiDi.training_step (src/diffusion/model.py:200-243) around the HF modules, line for line, with the reference's own
    ``get_x0`` / ``get_diffusion_variables`` / ``scale_input`` imported from /root/reference (they import only torch).
    ``calculate_batch_ce`` (src/metrics.py:91-93) and ``flat_mean`` (model.py:62-63) are restated: their modules import
    loguru / lightning, which are not installed.  Returns everything the fixture stores."""
    import torch.nn.functional as F
    from src.diffusion.utils import get_diffusion_variables, get_x0, scale_input

    def calculate_batch_ce(logits, target, non_pad_mask):  # metrics.py:91-93
        non_pad_mask = non_pad_mask.bool().view(-1)
        return F.cross_entropy(logits.view(-1, logits.shape[-1])[non_pad_mask], target.view(-1)[non_pad_mask])

    def flat_mean(tensor):  # model.py:62-63
        return tensor.mean(dim=list(range(1, len(tensor.shape))))

    emb = model.emb(tgt.input_ids)
    x_0 = get_x0(emb, model.std_0)
    noise = torch.randn_like(x_0)
    x_t, t = get_diffusion_variables(model.diffusion_steps, x_0, model.sigmas, noise, noise_factor)
    x_0_hat, _ = model(encoder_input_ids=ctx.input_ids, encoder_attention_mask=ctx.attention_mask, decoder_inputs_embeds=x_t, time_ids=t)
    x_0_hat.retain_grad()
    logits = model.classifier(x_0_hat) if anchor_loss else model.classifier(x_0)
    ce = calculate_batch_ce(logits, tgt.input_ids, tgt.attention_mask)
    non_pad_mask = tgt.attention_mask.unsqueeze(-1)
    # t is [B, 1] and flat_mean gives [B]: torch.where broadcasts to [B, B] -- reproduced as written (model.py:222-227)
    mse = torch.where(t == 1, flat_mean((x_0_hat - emb) ** 2 * non_pad_mask), flat_mean((x_0_hat - x_0) ** 2 * non_pad_mask)).mean()
    noise2, sigma_T = torch.randn_like(x_0), model.sigmas[-1]  # model.py:229: drawn, never used
    x_T = scale_input(x_0, sigma_T)
    t_T_loss = (x_T**2 * non_pad_mask).mean()
    loss = mse + ce
    if not anchor_loss:
        loss = loss + t_T_loss
    with torch.no_grad():
        ce_hat = calculate_batch_ce(model.classifier(x_0_hat), tgt.input_ids, tgt.attention_mask)
    return dict(emb=emb, x_0=x_0, x_t=x_t, t=t, x_0_hat=x_0_hat, mse=mse, ce=ce, t_T=t_T_loss, loss=loss, ce_hat=ce_hat)


def training_cases():
    """(T) DiDi.training_step on the tiny model (same weights as tiny_weights.safetensors): the RNG draws in the reference's order
    (get_x0's randn_like, the noise randn_like, randint for t -- captured as `t` --, the dead randn_like of model.py:229), x_t,
    x_0_hat, the five logged scalars, and the gradients autograd gives: every trainable parameter's, plus the intermediates of
    the LAST decoder layer's feed-forward block (BertIntermediate output, BertOutput.dense output + residual, their gradients),
    which pin the GEMM and LayerNorm backward kernels one at a time.  The encoder is frozen (configs/train.yaml:19)."""
    model = build_model(TINY, seed=11)
    for p_ in model.encoder.parameters():
        p_.requires_grad_(False)  # freeze_params(self.encoder), model.py:114-115
    V = TINY["vocab"]
    ids, mask = make_inputs(4, 12, V, seed=5)
    tids, tmask = make_inputs(4, 10, V, seed=8, min_len=3)
    last = model.decoder.encoder.layer[-1]
    for name, anchor, nf in (("train_tiny_plain", False, 1.0), ("train_tiny_anchor_nf2", True, 2.0)):
        for seed in range(300, 400):  # a seed whose randint gives some (not all) rows t == 1: both branches of the mse `where`
            torch.manual_seed(seed)
            _ = torch.randn(4, 10, TINY["dec_dim"]), torch.randn(4, 10, TINY["dec_dim"])
            t_try = torch.randint(1, TINY["T"] + 1, size=(4, 1))
            if 0 < int((t_try == 1).sum()) < 4:
                break
        model.zero_grad(set_to_none=True)
        stash = {}
        def keep(key, pick, grad=True):  # a forward hook that returns None (a returned value would replace the output)
            def hook(m, i, o):
                v = pick(i, o)
                if grad:
                    v.retain_grad()
                stash[key] = v
            return hook

        hooks = [
            last.intermediate.register_forward_hook(keep("ffn", lambda i, o: o)),
            last.output.LayerNorm.register_forward_hook(keep("ln_in", lambda i, o: i[0])),
            last.output.register_forward_hook(keep("resid", lambda i, o: i[1], grad=False)),
        ]
        torch.manual_seed(seed)
        with NoiseTap() as tap:
            r = ref_training_step(model, Ctx(ids, mask), Ctx(tids, tmask), anchor, nf, tap)
        r["loss"].backward()
        for h in hooks:
            h.remove()
        assert torch.equal(r["t"], t_try), "the seed search must replay the reference's draw order"
        t = {"in.input_ids": ids, "in.attention_mask": mask, "in.target_ids": tids, "in.target_mask": tmask,
             "t": r["t"].contiguous(), "out.emb": r["emb"].detach().contiguous(), "out.x_0": r["x_0"].detach().contiguous(),
             "out.x_t": r["x_t"].detach().contiguous(), "out.x_0_hat": r["x_0_hat"].detach().contiguous(),
             "out.scalars": torch.stack([r[k].detach() for k in ("mse", "ce", "t_T", "loss", "ce_hat")]),
             "grad.x_0_hat": r["x_0_hat"].grad.contiguous(),
             "last.ffn": stash["ffn"].detach().contiguous(), "last.ffn.grad": stash["ffn"].grad.contiguous(),
             "last.resid": stash["resid"].detach().contiguous(),
             "last.ln_in": stash["ln_in"].detach().contiguous(), "last.ln_in.grad": stash["ln_in"].grad.contiguous()}
        for i, d in enumerate(tap.draws):
            t[f"noise.{i}"] = d.contiguous()
        n_grads = 0
        for n, p_ in model.named_parameters():
            if p_.grad is not None:
                t["g." + n] = p_.grad.detach().clone().contiguous()
                n_grads += 1
        meta = {"spec": json.dumps(TINY), "run": json.dumps(dict(anchor_loss=anchor, noise_factor=nf, seed=seed)),
                "n_noise": str(len(tap.draws)), "n_grads": str(n_grads)}
        save_file(t, os.path.join(OUT, name + ".safetensors"), metadata=meta)
        print(f"wrote {name}: seed {seed}, t = {r['t'].view(-1).tolist()}, scalars {[round(float(r[k]), 6) for k in ('mse', 'ce', 't_T', 'loss', 'ce_hat')]}, "
              f"{n_grads} parameter gradients, {len(tap.draws)} draws")


CFG2 = dict(vocab=8008, enc_kind="blenderbot", enc_dim=1280, enc_layers=2, enc_heads=32, enc_ffn=5120,
            dec_dim=768, dec_layers=12, dec_heads=12, dec_ffn=3072, T=2000, schedule="sqrt")
# (name, mode, step_freq, weights, stored calls): ddpm f=1 is BASELINE configs[1] (what bench.py times), euler f=10 is the
# "DDIM 200-step" loop of configs[2]; the stored calls bracket where the 16-bit error peaks (calls ~1700-1850 of 2000)
FULLSIZE = [
    ("cfg2_ddpm_f1", "ddpm", 1, "random", [0, 1, 2, 200, 600, 1000, 1400, 1700, 1800, 1870, 1950, 1999]),
    ("cfg2_ddpm_f1_trained_like", "ddpm", 1, "trained_like", [0, 1, 2, 200, 600, 1000, 1400, 1700, 1800, 1870, 1950, 1999]),
    ("cfg2_euler_f10", "euler", 10, "random", [0, 1, 2, 20, 60, 100, 140, 170, 185, 195, 200]),
]
FULLSIZE_B_GPU, FULLSIZE_B_REF, FULLSIZE_B_XT = 48, 16, 4
FULLSIZE_PHILOX_SEED = 0xBADB01  # configs/train.yaml:1


class PhiloxFeed:
    """Make the reference sampler draw its noise from the product's counter-based stream instead of torch's generator:
    draw i of a B_ref-dialog run = the first B_ref dialogs' share of draw i of a B_gpu-dialog GPU run (engine.hip: counter =
    base_offset + i * (B_gpu*S*H/4) + element/4), computed by the numpy restatement oracle/philox.py.  The GPU box can then
    reproduce the run from (seed, B_gpu) alone -- no noise tensor travels, and the product runs its in-kernel generator, as in
    bench.py.  Replaces ``torch.randn_like`` and ``Tensor.normal_`` while ``sample`` runs (src/sampling.py:13,54,74,79)."""

    def __init__(self, seed, ctr_per_draw):
        self.seed, self.ctr_per_draw, self.i = seed, ctr_per_draw, 0

    def _draw(self, shape):
        from oracle.philox import normal
        import numpy as np

        n = int(np.prod(shape))
        t = torch.from_numpy(normal(n, self.seed, self.i * self.ctr_per_draw)).view(*shape)
        self.i += 1
        return t

    def __enter__(self):
        self._randn_like, self._normal_ = torch.randn_like, torch.Tensor.normal_
        feed = self

        def randn_like(x, *a, **k):
            return feed._draw(tuple(x.shape)).to(x.dtype)

        def normal_(self_t, mean=0, std=1, **k):
            assert mean == 0 and std == 1
            return self_t.copy_(feed._draw(tuple(self_t.shape)))

        torch.randn_like, torch.Tensor.normal_ = randn_like, normal_
        return self

    def __exit__(self, *exc):
        torch.randn_like, torch.Tensor.normal_ = self._randn_like, self._normal_


def fullsize_cases(only=None):
    """(X) The BENCHMARKED model size under the reference sampler: HF ``BertModel`` 12 x 768 decoder + 2 x 1280 Blenderbot encoder
    holding the seed-1234 synthetic weights (didi_amd.synthetic: what bench.py and the GPU tests build), the imported
    ``src.sampling.sample``, 16 dialogs, T = 2000.  Stored: final ids and top-2 logit gaps of the 16 dialogs, the final x0
    (the classifier's input), and the denoiser input x_t of the first 4 dialogs at a dozen calls.  ~40 CPU-minutes per
    2000-call case on 8 cores."""
    import time

    sys.path.insert(0, os.path.dirname(OUT))
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from golden_util import trained_like
    from didi_amd.config import DiDiConfig
    from didi_amd.synthetic import synthetic_batch, synthetic_state_dict
    from src.sampling import sample

    cfg = DiDiConfig(diffusion_steps=CFG2["T"])
    Bg, Bo, Bx, C, S, H = FULLSIZE_B_GPU, FULLSIZE_B_REF, FULLSIZE_B_XT, 64, 64, CFG2["dec_dim"]
    ids, mask = synthetic_batch(Bg, C, cfg.vocab, seed=11)
    for name, mode, f, weights, keep in FULLSIZE:
        if only and name not in only:
            continue
        sd = synthetic_state_dict(cfg, seed=1234)
        if weights == "trained_like":
            sd = trained_like(sd)
        m = build_model(CFG2, seed=0)
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        assert all(("position_ids" in k or "token_type_ids" in k or "word_embeddings" in k or "pooler" in k) for k in missing), missing
        m.keep_calls, m.n_calls = set(keep), 0
        m.calls.clear()
        x0_seen = []
        hook = m.classifier.register_forward_pre_hook(lambda mod, inp: x0_seen.append(inp[0].detach().clone()))
        t0 = time.time()
        orig_forward = m.forward

        def ticking(*a, _orig=orig_forward, **k):
            if m.n_calls % 50 == 0:
                print(f"  {name}: call {m.n_calls}  ({time.time() - t0:.0f} s)", flush=True)
            return _orig(*a, **k)

        m.forward = ticking
        with PhiloxFeed(FULLSIZE_PHILOX_SEED, Bg * S * H // 4) as feed, torch.no_grad():
            logits = sample(Ctx(ids[:Bo], mask[:Bo]), m, mode, f, max_len=S, raw_output=True)
        hook.remove()
        n_calls = m.n_calls
        assert n_calls == len(range(cfg.diffusion_steps, 1, -f)) + 1 and len(m.calls) == len(keep) and len(x0_seen) == 1
        top2 = logits.topk(2, -1).values
        t = {"in.input_ids": ids, "in.attention_mask": mask, "out.ids": logits.argmax(-1).contiguous(),
             "out.top2_gap": (top2[..., 0] - top2[..., 1]).contiguous(), "out.x0": x0_seen[0].contiguous(),
             # the schedule buffers as a checkpoint would carry them: configure_schedule's last bits differ between host CPUs
             "w.sigmas": sd["sigmas"].contiguous(), "w.std_0": sd["std_0"].reshape(1).contiguous()}
        if weights == "trained_like":  # its exp() on the LayerNorm gammas is libm-dependent too
            t.update({"w." + k: v.contiguous() for k, v in sd.items() if k.startswith("decoder.") and k.endswith("LayerNorm.weight")})
        for k, c in zip(keep, m.calls):
            t[f"xt.{k}"] = c[:Bx].contiguous()
        run = dict(mode=mode, step_freq=f, max_len=S, weights=weights, weight_seed=1234, batch_seed=11,
                   philox_seed=FULLSIZE_PHILOX_SEED, B_gpu=Bg, B_ref=Bo, B_xt=Bx, flush_denormal=True)
        save_file(t, os.path.join(OUT, name + ".safetensors"),
                  metadata={"spec": json.dumps(CFG2), "run": json.dumps(run), "weights_sha256": sd_hash(sd), "n_calls": str(n_calls),
                            "n_noise": str(feed.i), "kept_calls": json.dumps(keep)})
        print(f"wrote {name}: {n_calls} calls, {feed.i} draws in {time.time() - t0:.0f} s; median top-2 gap "
              f"{float(t['out.top2_gap'].median()):.3f}, min {float(t['out.top2_gap'].min()):.2e}", flush=True)


def preprocessor_cases():
    """(P) ``Preprocessor`` of /root/reference/src/data/utils.py:6-23 for the three tokenizer families.  Its module cannot be imported
    here (it imports src.diffusion.model -> lightning, and its constructor fetches a tokenizer from the hub), so the class
    definition is taken from the reference file's syntax tree at run time -- together with ``Modes`` / ``get_mode`` from
    src/diffusion/model.py:16-26, which it calls -- and executed as it stands; the only stand-in is the ``AutoTokenizer`` name,
    whose ``from_pretrained`` returns an object carrying the special-token strings of the family's real tokenizer.  Nothing of
    the reference's text is stored: the fixture holds inputs and the strings the class returned."""
    import ast
    from enum import Enum

    def defs(path, names):
        tree = ast.parse(open(path).read())
        keep = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
        assert len(keep) == len(names), (path, [n.name for n in keep])
        return compile(ast.Module(body=keep, type_ignores=[]), path, "exec")

    specials = {  # the special-token strings of the hub tokenizers the reference names (configs/train.yaml:3-5)
        "bert-base-uncased": dict(cls_token="[CLS]", sep_token="[SEP]", bos_token=None, eos_token=None),
        "facebook/blenderbot-400M-distill": dict(cls_token=None, sep_token=None, bos_token="<s>", eos_token="</s>"),
        "t5-base": dict(cls_token=None, sep_token=None, bos_token=None, eos_token="</s>"),
    }

    class Tok:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    class AutoTokenizer:
        @staticmethod
        def from_pretrained(name, **kw):
            return Tok(**specials[name])

    ns = {"Enum": Enum, "AutoTokenizer": AutoTokenizer}
    exec(defs(os.path.join(REF, "src/diffusion/model.py"), ["Modes", "get_mode"]), ns)
    exec(defs(os.path.join(REF, "src/data/utils.py"), ["Preprocessor"]), ns)
    inputs = [("how are you ?", "fine , thanks ."), (["hi", "hello there", "what's up ?"], "not much"), ("", ""), (["only one"], "x"),
              ([], "reply to nothing")]
    out = []
    for name in specials:
        p = ns["Preprocessor"](name)
        out.append({"base_name": name, "family": ns["get_mode"](name).value, "tokens": specials[name],
                    "bos": p.bos, "eos": p.eos, "sep": p.sep,
                    "cases": [{"src": s_, "trg": t_, "out": list(p(s_, t_))} for s_, t_ in inputs]})
    bad = None
    try:
        ns["get_mode"]("gpt2")
    except ValueError as e:
        bad = str(e)
    with open(os.path.join(OUT, "preprocessor.json"), "w") as f:
        json.dump({"families": out, "unsupported": {"base_name": "gpt2", "error": bad}}, f, indent=1)
    print("wrote preprocessor.json:", [(o["family"], o["bos"], o["eos"], o["sep"]) for o in out])


def main():
    sys.path.insert(0, REF)
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "t5":  # add the T5 cases without regenerating the others
        t5_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "wide":  # add the wide-head encoder case without regenerating the others
        wide_head_cases()
        wide_t5_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize":  # the cfg2-sized 2000-call cases (~40 CPU-minutes each); optional names follow
        torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "8")))
        # x86 runs subnormal operands through microcode: from call ~600 of the ddpm loop on, softmax tails / GELU tails that
        # small made every decoder call 8x slower (27 s -> 224 s per 50 calls).  Flushing them moves values by < 1.2e-38 absolute.
        torch.set_flush_denormal(True)
        fullsize_cases(set(sys.argv[2:]) or None)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "preproc":
        preprocessor_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "train":  # add the training-step cases without regenerating the others
        training_cases()
        return
    from src.diffusion.utils import configure_schedule

    # (F) schedules, straight from the reference
    t = {}
    for sched in ("sqrt", "cosine", "linear"):
        for T in (50, 2000):
            s, std0 = configure_schedule(T, sched)
            t[f"{sched}.{T}.sigmas"] = s.contiguous()
            t[f"{sched}.{T}.std_0"] = std0.reshape(1).contiguous()
    save_file(t, os.path.join(OUT, "schedules.safetensors"))

    # (A-D) tiny blenderbot-encoder model; weights stored once
    model = build_model(TINY, seed=11)
    w = state_dict_of(model)
    save_file({"w." + k: v for k, v in w.items()}, os.path.join(OUT, "tiny_weights.safetensors"),
              metadata={"spec": json.dumps(TINY), "sha256": sd_hash(w)})
    ids, mask = make_inputs(4, 12, TINY["vocab"], seed=5)
    cases = [
        ("tiny_ddpm_f1", dict(mode="ddpm", step_freq=1, max_len=-1, noise_seed=101), {}),
        ("tiny_ddpm_f3_len8", dict(mode="ddpm", step_freq=3, max_len=8, noise_seed=102), {}),
        ("tiny_euler_f2", dict(mode="euler", step_freq=2, max_len=-1, noise_seed=103), {}),
        ("tiny_euler_churn_f4", dict(mode="euler", step_freq=4, max_len=-1, noise_seed=104),
         dict(s_churn=1.5, s_tmin=0.5, s_tmax=50.0)),
    ]
    for name, run, euler_kw in cases:
        for k in ("s_churn", "s_tmin", "s_tmax"):
            setattr(model, k, euler_kw.get(k, {"s_churn": 0.0, "s_tmin": 0.0, "s_tmax": float("inf")}[k]))
        logits, draws, calls = run_case(model, ids, mask, run["mode"], run["step_freq"], run["max_len"], run["noise_seed"])
        ctx = model(encoder_input_ids=ids, encoder_attention_mask=mask, decoder_inputs_embeds=calls[0],
                    time_ids=torch.ones(ids.shape[0], 1, dtype=torch.long))[1]
        save_case(name, dict(TINY, **euler_kw), run, logits, draws, calls, ctx.detach(), ids, mask)
    # eager vs sdpa deviation on the first case (information only)
    m2 = build_model(TINY, seed=11, attn="sdpa")
    l_e, _, _ = run_case(build_model(TINY, seed=11), ids, mask, "ddpm", 1, -1, 101)
    l_s, _, _ = run_case(m2, ids, mask, "ddpm", 1, -1, 101)
    print("eager vs sdpa max|dlogits| =", (l_e - l_s).abs().max().item(),
          "ids equal:", bool((l_e.argmax(-1) == l_s.argmax(-1)).all()))

    # (E) bert-encoder variant (enc_dim == dec_dim -> no adapter), weights inside the case file
    mb = build_model(TINY_BERT, seed=12)
    ids_b, mask_b = make_inputs(3, 9, TINY_BERT["vocab"], seed=6)
    run = dict(mode="ddpm", step_freq=2, max_len=-1, noise_seed=105)
    logits, draws, calls = run_case(mb, ids_b, mask_b, **{k: run[k] for k in ("mode", "step_freq", "max_len", "noise_seed")})
    ctx = mb(encoder_input_ids=ids_b, encoder_attention_mask=mask_b, decoder_inputs_embeds=calls[0],
             time_ids=torch.ones(3, 1, dtype=torch.long))[1]
    save_case("tiny_bertenc_ddpm_f2", TINY_BERT, run, logits, draws, calls, ctx.detach(), ids_b, mask_b,
              weights=state_dict_of(mb))

    # (G) BASELINE config 1 (V=8008, T=50, B=8, C=24): weights are NOT stored (13 MB); they are
    # regenerated from the seed by tests/golden_util.py:config1_state_dict and checked by sha256.
    sys.path.insert(0, os.path.dirname(OUT))
    from golden_util import config1_inputs, config1_state_dict

    w1 = config1_state_dict()
    m1 = build_model(CONFIG1, seed=0)
    missing, unexpected = m1.load_state_dict(w1, strict=False)
    assert not unexpected, unexpected
    assert all(("position_ids" in k or "token_type_ids" in k or "word_embeddings" in k or "pooler" in k) for k in missing), missing
    ids1, mask1 = config1_inputs()
    run = dict(mode="ddpm", step_freq=1, max_len=-1, noise_seed=0xBADB01)
    logits, draws, calls = run_case(m1, ids1, mask1, "ddpm", 1, -1, 0xBADB01)
    keep = [0, 1, 10, 25, 40, 49]
    t = {"in.input_ids": ids1, "in.attention_mask": mask1, "out.ids": logits.argmax(-1),
         "out.logits_b0": logits[0].contiguous(), "noise.0": draws[0].contiguous()}
    for i in keep:
        t[f"xt.{i}"] = calls[i].contiguous()
    # the noise stream is the plain global-generator sequence: prove it, so tests may regenerate it
    torch.manual_seed(0xBADB01)
    regen = [torch.randn(draws[0].shape)] + [torch.empty(draws[0].shape).normal_(0, 1) for _ in draws[1:]]
    assert all(torch.equal(a, b) for a, b in zip(regen, draws))
    save_file(t, os.path.join(OUT, "config1_ddpm.safetensors"),
              metadata={"spec": json.dumps(CONFIG1), "run": json.dumps(run), "weights_sha256": sd_hash(w1),
                        "n_noise": str(len(draws)), "n_calls": str(len(calls)), "kept_calls": json.dumps(keep)})
    print("wrote config1_ddpm; weights sha256", sd_hash(w1)[:16])
    t5_cases()
    training_cases()


if __name__ == "__main__":
    main()
