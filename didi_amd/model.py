"""DiDiSampler: the sampler-visible part of reference ``DiDi`` (src/diffusion/model.py:66-198) on HIP.

Same call surface as the reference module on this path:
  * attributes ``emb, time_embeds, classifier, sigmas, diffusion_steps, s_churn, s_tmin, s_tmax,
    sampling_mode, step_freq`` (model.py:92-127), read by ``sample`` (src/sampling.py:11-13,39,87-88);
  * ``model(encoder_input_ids=, encoder_attention_mask=, decoder_inputs_embeds=, time_ids=,
    context=) -> (x0_hat, context)`` with the reference's ``ValueError`` when neither ids nor
    context is given (model.py:172-198).  ``context`` is returned and accepted back by reference.
Weights come from a state dict in the reference's checkpoint key layout.

Everything numeric runs in libdidi_hip (didi_amd/csrc) through didi_amd/ops.py and the engine
entry points; torch only owns memory.  Weight packing at load time: q/k/v (and cross k/v) Linear
weights are row-concatenated, then every matrix is split into bf16 hi/lo planes on the device.
"""
import ctypes as C
import math

import torch

from . import _lib, ops
from ._lib import lib
from .config import DiDiConfig
from .schedule import configure_schedule


# Default precision schedule of "fp16-sched", as fractions of the number of diffusion steps T (measured at
# T = 2000, sqrt schedule: t > 300 / t > 128 / the rest): one product while the input is mostly fresh noise, the weights'
# remainder once x_t starts to carry the previous estimate, all three products for the last steps.  The one-product regime
# reaches down to 0.15 T because its calls cycle through DEFAULT_WEIGHT_DITHER differently rounded copies of the weights: plain
# rounding is the SAME perturbation of the model at every call and its effect adds up coherently over the loop (T = 2000, f = 1,
# fp16 above 0.15 T: max x_t 1.0e-3 with one rounding, 6.4e-4 with four; without the copies the threshold was 0.35 T:
# profiles/r02_dither_probe_*.log).
DEFAULT_SCHEDULE = ((0.15, "fp16"), (0.064, "fp16x2"), (0, "fp16x3"))
DEFAULT_WEIGHT_DITHER = 4
# The euler sampler integrates an ODE: an error made at a high timestep shifts the trajectory for good instead of being drowned
# by the next step's fresh noise as in ddpm (T = 2000, f = 10, final x_t: 2.7e-4 with the ddpm schedule whatever the tail, 4.2e-5
# without a one-product regime; ids 4 vs 1 of 2048 on held-out seeds: profiles/r02_sched_probe_f10_euler.log,
# r02_parity_heldout_f10_euler.log).  So euler loops get no one-product regime.
DEFAULT_SCHEDULE_EULER = ((0.064, "fp16x2"), (0, "fp16x3"))
# The loop forgets an earlier error per CALL, not per unit of t: with step_freq > 1 the timestep thresholds leave too few calls
# in the precise regime (T = 2000, f = 10: 13 calls of fp16x3 -> the final x_t is 4.6e-4 off and 4 of 2048 ids differ; with 32
# calls 6.5e-5 / 0, with 64 calls 8e-6 / 0: tests/sched_probe_f10.py, profiles/r02_sched_probe_f10_ddpm.log).  Floors applied
# after the timestep rule, as (calls_from_the_end, precision): the last n calls run at least that precision.
DEFAULT_TAIL_CALLS = ((64, "fp16x3"),)
_RANK = {"fp16": 0, "fp16x2": 1, "fp16x3": 2}


class _Planes:
    """A Linear's packed weights: split planes [N, K] (bf16, or fp16 for the "fp16" precision) + fp32 bias."""

    def __init__(self, weight, bias, device, dtype=torch.bfloat16):
        w = weight.detach().to(device=device, dtype=torch.float32).contiguous()
        self.hi, self.lo = ops.split(w, dtype=dtype)
        self.bias = None if bias is None else bias.detach().to(device=device, dtype=torch.float32).contiguous()
        self.N, self.K = w.shape

    def c(self):
        return _lib.Linear(self.hi.data_ptr(), self.lo.data_ptr(), _lib.ptr(self.bias), self.N, self.K)

    def __call__(self, a_hi, a_lo, precision, **kw):
        return ops.gemm(a_hi, a_lo, self.hi, self.lo, bias=self.bias, precision=precision, **kw)


class _Embedding:
    """``nn.Embedding`` stand-in (``.weight`` + gather); the gather is pure indexing."""

    def __init__(self, weight):
        self.weight = weight

    def __call__(self, ids):
        return ops.embed_gather(ids.reshape(-1, ids.shape[-1]), self.weight).reshape(*ids.shape, -1)


class _Classifier:
    """``nn.Linear(dec_dim, vocab, bias=False)`` stand-in (model.py:118): fp32 logits via the HIP GEMM."""

    def __init__(self, planes, weight, owner):
        self.planes, self.weight, self._owner = planes, weight, owner

    def __call__(self, x):
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).contiguous().float()
        hi, lo = ops.split(x2, dtype=self.planes.hi.dtype)
        _, _, y = self.planes(hi, lo, self._owner.round_precision, out_f32=True, out_planes=False)
        return y.reshape(*lead, -1)


class DiDiSampler:
    def __init__(self, state_dict, cfg: DiDiConfig, device="cuda", precision="bf16x3", schedule=None, tail_calls=None, weight_dither=None,
                 encoder_only=False, act_dither=None):
        """``precision``: bf16 | fp16 (one MFMA product), fp16x2 (weights with their remainder), bf16x3 (fp32-class), or
        "fp16-sched": fp16 planes with the default precision schedule ``DEFAULT_SCHEDULE``.
        ``schedule``: precision schedule of ``sample`` over the TIMESTEP of each denoiser call, a tuple of (t_above, precision)
        pairs in decreasing t: a call whose time-embedding row is > t_above runs in that precision (fp16 | fp16x2 | fp16x3);
        the last pair must have t_above = 0.  Needs an fp16-plane ``precision``.
        ``tail_calls``: (calls_from_the_end, precision) floors on top of the schedule: the last n calls of a loop run at least
        that precision ("fp16-sched" defaults to ``DEFAULT_TAIL_CALLS``; an explicit schedule to none).
        ``weight_dither``: P > 1 keeps P differently rounded fp16 copies of the per-call GEMM weights and cycles the one-product
        calls of a loop through them (``_build_weight_variants``); fp16 planes only ("fp16-sched" defaults to
        ``DEFAULT_WEIGHT_DITHER``, everything else to none).
        ``act_dither``: the one-product calls of a loop round their fp16 ACTIVATION planes with an offset that cycles over
        four values (didi_sample_args.act_dither: the activation-side twin of ``weight_dither``); fp16 planes only.
        ``encoder_only``: build the context encoder (+ adapter) and the schedule buffers only -- what the trainer's frozen part
        needs (didi_amd/trainer.py); the decoder, embedding and classifier planes are not allocated and sampling is refused."""
        schedule_euler = None
        if precision == "fp16-sched":
            schedule_euler = None if schedule else DEFAULT_SCHEDULE_EULER  # an explicit schedule is used for both samplers
            precision, schedule = "fp16x2", (schedule or DEFAULT_SCHEDULE)
            tail_calls = DEFAULT_TAIL_CALLS if tail_calls is None else tail_calls
            weight_dither = DEFAULT_WEIGHT_DITHER if weight_dither is None else weight_dither
        if precision not in ("bf16", "fp16", "fp16x2", "bf16x3"):
            raise ValueError("precision must be one of bf16, fp16, fp16x2, bf16x3, fp16-sched")
        def resolve(schedule):  # thresholds below 1 are fractions of T
            resolved = []
            for t, p in schedule:
                frac = 0 < t < 1
                t = int(round(t * cfg.diffusion_steps)) if frac else int(t)
                if resolved and resolved[-1][0] == t and (frac or resolved[-1][2]):
                    resolved.pop()  # a small T collapses fractional thresholds: the later (more precise) regime keeps the slot
                resolved.append((t, p, frac))
            resolved = [(t, p) for t, p, _ in resolved]
            schedule = tuple(resolved)
            ok = precision in ("fp16", "fp16x2") and all(p in ("fp16", "fp16x2", "fp16x3") for _, p in schedule)
            ok = ok and schedule[-1][0] == 0 and all(a[0] > b[0] for a, b in zip(schedule, schedule[1:]))
            if not ok:
                raise ValueError("a precision schedule needs fp16-plane precisions and (t_above, precision) pairs in decreasing t ending at 0")
            return schedule

        if schedule is not None:
            schedule = resolve(schedule)
        if schedule_euler is not None:
            schedule_euler = resolve(schedule_euler)
        self.cfg, self.device, self.precision = cfg, torch.device(device), precision
        self.weight_dither = int(weight_dither or 0) if precision in ("fp16", "fp16x2") else 0
        self.act_dither = bool(act_dither) and precision in ("fp16", "fp16x2")
        self.schedule = schedule
        self.schedule_euler = schedule_euler if schedule_euler is not None else schedule  # "fp16-sched": DEFAULT_SCHEDULE_EULER
        self.tail_calls = tuple(tail_calls or ())  # an explicit schedule is taken as written unless floors are given too
        if self.tail_calls and (not schedule or any(p not in _RANK or n < 0 for n, p in self.tail_calls)):
            raise ValueError("tail_calls = (calls_from_the_end, fp16|fp16x2|fp16x3) pairs, and only together with a schedule")
        sd = state_dict
        dev = self.device
        f = lambda k: sd[k].detach().to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
        # attributes the sampler reads (model.py:92-127)
        self.diffusion_steps = cfg.diffusion_steps
        self.sampling_mode, self.step_freq, self.noise_factor = cfg.sampling_mode, cfg.step_freq, cfg.noise_factor
        self.s_churn, self.s_tmin, self.s_tmax = cfg.s_churn, cfg.s_tmin, cfg.s_tmax
        self.encoder_dim, self.decoder_dim, self.pad_idx = cfg.enc_dim, cfg.dec_dim, cfg.pad_idx
        self.encoder_only = bool(encoder_only)
        self._pd = ops.plane_dtype(precision)  # decoder + classifier planes; the encoder side stays bf16x3
        if not self.encoder_only:
            self.emb = _Embedding(f("emb.weight"))
            self.time_embeds = _Embedding(f("time_embeds.weight"))
            cls_w = self.emb.weight if cfg.tie_weights else f("classifier.weight")
            self._cls = _Planes(cls_w, None, dev, self._pd)
            self.classifier = _Classifier(self._cls, cls_w, self)
        if "sigmas" in sd:
            self._sigmas_host = sd["sigmas"].detach().float().cpu()
        else:
            self._sigmas_host, _ = configure_schedule(cfg.diffusion_steps, cfg.schedule)
        self.sigmas = self._sigmas_host.to(dev)
        # std_0 = sqrt(beta_0) (configure_schedule's second value; model.py:125-127): read by the training step's get_x0
        self.std_0 = float(sd["std_0"]) if "std_0" in sd else float(configure_schedule(cfg.diffusion_steps, cfg.schedule)[1])
        if not self.encoder_only:
            self._build_decoder(sd, f)
        self._build_encoder(sd, f)
        self._bufs = {}
        self._ctx_key = None

    # ------------------------------------------------------------------ decoder (engine structs)
    def _build_decoder(self, sd, f):
        cfg, dev = self.cfg, self.device
        H = cfg.dec_dim
        self._keep = []  # tensors referenced by raw pointer from the ctypes structs
        layers = (_lib.DecoderLayer * cfg.dec_layers)()
        for i in range(cfg.dec_layers):
            p = f"decoder.encoder.layer.{i}."

            def cat(block, names):
                w = torch.cat([sd[f"{p}{block}.self.{n}.weight"] for n in names], 0)
                b = torch.cat([sd[f"{p}{block}.self.{n}.bias"] for n in names], 0)
                return _Planes(w, b, dev, self._pd)

            def lin(name):
                return _Planes(sd[p + name + ".weight"], sd[p + name + ".bias"], dev, self._pd)

            parts = dict(
                qkv=cat("attention", ("query", "key", "value")), attn_out=lin("attention.output.dense"),
                cq=cat("crossattention", ("query",)), ckv=cat("crossattention", ("key", "value")),
                cross_out=lin("crossattention.output.dense"), ffn_up=lin("intermediate.dense"), ffn_down=lin("output.dense"),
            )
            lns = dict(ln1=p + "attention.output.LayerNorm", ln2=p + "crossattention.output.LayerNorm", ln3=p + "output.LayerNorm")
            ly = layers[i]
            for k, v in parts.items():
                setattr(ly, k, v.c())
                self._keep.append(v)
            for k, name in lns.items():
                g, b = f(name + ".weight"), f(name + ".bias")
                setattr(ly, k + "_g", g.data_ptr())
                setattr(ly, k + "_b", b.data_ptr())
                self._keep += [g, b]
        self._pos = f("decoder.embeddings.position_embeddings.weight")
        self._tok0 = f("decoder.embeddings.token_type_embeddings.weight")[0].contiguous()
        self._eln_g, self._eln_b = f("decoder.embeddings.LayerNorm.weight"), f("decoder.embeddings.LayerNorm.bias")
        d = _lib.Decoder()
        d.H, d.heads, d.I, d.L, d.ln_eps = H, cfg.dec_heads, cfg.dec_ffn, cfg.dec_layers, cfg.dec_ln_eps
        d.time_embeds, d.pos_emb, d.tok_type0 = self.time_embeds.weight.data_ptr(), self._pos.data_ptr(), self._tok0.data_ptr()
        d.emb_ln_g, d.emb_ln_b = self._eln_g.data_ptr(), self._eln_b.data_ptr()
        d.layers_host = C.cast(layers, C.POINTER(_lib.DecoderLayer))
        self._layers, self._dec = layers, d
        self._dec_variants = []
        if self.weight_dither > 1:
            self._build_weight_variants(sd)

    def _build_weight_variants(self, sd):
        """``weight_dither`` = P > 1: P copies of the per-call GEMM weights' hi planes, copy j rounding W + ((j + 1/2) / P - 1/2) ulp
        to fp16 (ordered dither: the copies' mean is within ulp / 2P of W).  One-product (fp16) calls of a loop cycle through
        them (didi_sample_args.dec_variants_host), so the weight-rounding error is not the same perturbation at every call."""
        cfg, dev, P = self.cfg, self.device, self.weight_dither
        names = ("qkv", "attn_out", "cq", "cross_out", "ffn_up", "ffn_down")
        for j in range(P):
            off = (j + 0.5) / P - 0.5
            layers = (_lib.DecoderLayer * cfg.dec_layers)()
            for i in range(cfg.dec_layers):
                C.memmove(C.byref(layers[i]), C.byref(self._layers[i]), C.sizeof(_lib.DecoderLayer))
                p = f"decoder.encoder.layer.{i}."
                srcs = dict(
                    qkv=torch.cat([sd[f"{p}attention.self.{n}.weight"] for n in ("query", "key", "value")], 0),
                    attn_out=sd[p + "attention.output.dense.weight"], cq=sd[p + "crossattention.self.query.weight"],
                    cross_out=sd[p + "crossattention.output.dense.weight"], ffn_up=sd[p + "intermediate.dense.weight"],
                    ffn_down=sd[p + "output.dense.weight"])
                for n in names:
                    w = srcs[n].detach().to(device=dev, dtype=torch.float32)
                    _, e = torch.frexp(w)
                    ulp = torch.ldexp(torch.ones_like(w), (e - 11).clamp_min(-24))
                    hi = torch.where(w == 0, w, w + off * ulp).to(torch.float16).contiguous()
                    self._keep.append(hi)
                    lin = getattr(layers[i], n)
                    lin.w_hi = hi.data_ptr()
            d = _lib.Decoder()
            C.memmove(C.byref(d), C.byref(self._dec), C.sizeof(_lib.Decoder))
            d.layers_host = C.cast(layers, C.POINTER(_lib.DecoderLayer))
            self._dec_variants.append((d, layers))
        self._dec_variant_ptrs = (C.c_void_p * P)(*[C.addressof(d) for d, _ in self._dec_variants])

    def act_dither_for_call(self, k, prec):
        """didi_forward_args.act_dither of call k of a loop (Python-loop twin of the engine's rule, engine.hip dith_of)."""
        return 1024 + 2048 * ((k + k // 4) & 3) if self.act_dither and prec == _lib.PREC_FP16 else 0

    def dec_for_call(self, k, prec):
        """The decoder description call k of a loop runs with (Python-loop twin of the engine's choice)."""
        if self._dec_variants and prec == _lib.PREC_FP16:
            return self._dec_variants[k % len(self._dec_variants)][0]
        return self._dec

    # ------------------------------------------------------------------ context encoder (cold path: once per batch)
    def _build_encoder(self, sd, f):
        cfg, dev = self.cfg, self.device
        e = {}
        if cfg.enc_kind == "blenderbot":
            e["tok"], e["pos"] = f("encoder.embed_tokens.weight"), f("encoder.embed_positions.weight")
            e["layers"] = []
            for i in range(cfg.enc_layers):
                p = f"encoder.layers.{i}."
                qkv_w = torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
                qkv_b = torch.cat([sd[p + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
                e["layers"].append(dict(
                    ln1=(f(p + "self_attn_layer_norm.weight"), f(p + "self_attn_layer_norm.bias")),
                    qkv=_Planes(qkv_w, qkv_b, dev),
                    out=_Planes(sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], dev),
                    ln2=(f(p + "final_layer_norm.weight"), f(p + "final_layer_norm.bias")),
                    fc1=_Planes(sd[p + "fc1.weight"], sd[p + "fc1.bias"], dev),
                    fc2=_Planes(sd[p + "fc2.weight"], sd[p + "fc2.bias"], dev),
                ))
            e["ln_f"] = (f("encoder.layer_norm.weight"), f("encoder.layer_norm.bias"))
        elif cfg.enc_kind == "bert":
            q = "encoder.embeddings."
            e["tok"], e["pos"] = f(q + "word_embeddings.weight"), f(q + "position_embeddings.weight")
            e["tok0"] = f(q + "token_type_embeddings.weight")[0].contiguous()
            e["eln"] = (f(q + "LayerNorm.weight"), f(q + "LayerNorm.bias"))
            e["zero_row"] = torch.zeros(1, cfg.enc_dim, device=dev)
            e["layers"] = []
            for i in range(cfg.enc_layers):
                p = f"encoder.encoder.layer.{i}."
                qkv_w = torch.cat([sd[p + f"attention.self.{n}.weight"] for n in ("query", "key", "value")], 0)
                qkv_b = torch.cat([sd[p + f"attention.self.{n}.bias"] for n in ("query", "key", "value")], 0)
                e["layers"].append(dict(
                    qkv=_Planes(qkv_w, qkv_b, dev),
                    out=_Planes(sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"], dev),
                    ln1=(f(p + "attention.output.LayerNorm.weight"), f(p + "attention.output.LayerNorm.bias")),
                    fc1=_Planes(sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"], dev),
                    fc2=_Planes(sd[p + "output.dense.weight"], sd[p + "output.dense.bias"], dev),
                    ln2=(f(p + "output.LayerNorm.weight"), f(p + "output.LayerNorm.bias")),
                ))
        elif cfg.enc_kind == "t5":  # T5EncoderModel (model.py:39-42): keys under encoder.encoder.*
            q = "encoder.encoder."
            e["tok"] = f(q + "embed_tokens.weight") if q + "embed_tokens.weight" in sd else f("encoder.shared.weight")
            e["rel"] = f(q + "block.0.layer.0.SelfAttention.relative_attention_bias.weight")  # [buckets, heads]
            e["layers"] = []
            for i in range(cfg.enc_layers):
                p = f"{q}block.{i}.layer."
                qkv_w = torch.cat([sd[p + f"0.SelfAttention.{n}.weight"] for n in "qkv"], 0)
                ly = dict(ln1=f(p + "0.layer_norm.weight"), qkv=_Planes(qkv_w, None, dev),
                          out=_Planes(sd[p + "0.SelfAttention.o.weight"], None, dev), ln2=f(p + "1.layer_norm.weight"),
                          wo=_Planes(sd[p + "1.DenseReluDense.wo.weight"], None, dev))
                if cfg.enc_gated_ffn:
                    ly["wi0"] = _Planes(sd[p + "1.DenseReluDense.wi_0.weight"], None, dev)
                    ly["wi1"] = _Planes(sd[p + "1.DenseReluDense.wi_1.weight"], None, dev)
                else:
                    ly["wi"] = _Planes(sd[p + "1.DenseReluDense.wi.weight"], None, dev)
                e["layers"].append(ly)
            e["ln_f"] = f(q + "final_layer_norm.weight")
            e["bias_cache"] = {}
        else:
            raise ValueError(f"Unsupported encoder kind: {cfg.enc_kind}")
        if cfg.has_adapter:  # model.py:122-123
            e["ad0"] = _Planes(sd["adapter.0.weight"], sd["adapter.0.bias"], dev)
            e["ad2"] = _Planes(sd["adapter.2.weight"], sd["adapter.2.bias"], dev)
        self._enc = e

    def _encode_context(self, input_ids, attention_mask):
        """``DiDi._encode_context`` (model.py:165-170): encoder last_hidden_state (+ adapter) -> fp32 [B, C, H]."""
        cfg, e, P = self.cfg, self._enc, "bf16x3"
        B, Cn = input_ids.shape
        c_hi, c_lo, ctx = self._encode_raw(input_ids, attention_mask)
        if cfg.has_adapter:
            t_hi, t_lo, _ = e["ad0"](c_hi, c_lo, P, act="tanh")
            _, _, ctx = e["ad2"](t_hi, t_lo, P, out_f32=True, out_planes=False)
        return ctx.view(B, Cn, cfg.dec_dim)

    def _encode_raw(self, input_ids, attention_mask):
        """The encoder's last_hidden_state before the adapter -> (hi, lo bf16 planes, fp32), each [B*C, enc_dim].  The training
        step (didi_amd/trainer.py) starts its trainable graph here: the encoder is frozen (configs/train.yaml:19)."""
        cfg, e, P = self.cfg, self._enc, "bf16x3"  # cold path (1 of ~2000 calls): always fp32-class
        B, Cn = input_ids.shape
        E, A = cfg.enc_dim, cfg.enc_heads
        mask32 = attention_mask.to(torch.int32).contiguous()

        wide_heads = E // A > 64 or (E // A) % 8 != 0  # e.g. Blenderbot-3B's 80-wide heads (configs/train.yaml:3)

        def attn(qkv_hi, qkv_lo, qkv_f32=None):
            v = lambda t, i: None if t is None else t.view(B, Cn, 3 * E)[:, :, i * E:(i + 1) * E]  # noqa: E731
            if wide_heads:  # past the MFMA attention kernel's head dim: the fp32 LDS kernel of the training forward (cold path)
                o = ops.attention_train(v(qkv_f32, 0), v(qkv_f32, 1), v(qkv_f32, 2), A, causal=False, key_mask=mask32)
                return ops.split(o.view(B * Cn, E))
            q, k, vv = ((v(qkv_hi, i), v(qkv_lo, i)) for i in range(3))
            o_hi, o_lo = ops.attention(q, k, vv, A, causal=False, key_mask=mask32, precision=P)
            return o_hi.view(B * Cn, E), o_lo.view(B * Cn, E)

        if cfg.enc_kind == "blenderbot":
            scale = math.sqrt(E) if cfg.enc_scale_embedding else 1.0
            h = ops.embed_gather(input_ids, e["tok"], scale, e["pos"]).view(B * Cn, E)
            for ly in e["layers"]:
                y_hi, y_lo, _ = ops.layernorm(h, *ly["ln1"], cfg.enc_ln_eps)
                qkv_hi, qkv_lo, qkv32 = ly["qkv"](y_hi, y_lo, P, out_f32=wide_heads)
                a_hi, a_lo = attn(qkv_hi, qkv_lo, qkv32)
                _, _, h = ly["out"](a_hi, a_lo, P, residual=h, out_f32=True, out_planes=False)
                y_hi, y_lo, _ = ops.layernorm(h, *ly["ln2"], cfg.enc_ln_eps)
                f_hi, f_lo, _ = ly["fc1"](y_hi, y_lo, P, act="gelu")
                _, _, h = ly["fc2"](f_hi, f_lo, P, residual=h, out_f32=True, out_planes=False)
            c_hi, c_lo, ctx = ops.layernorm(h, *e["ln_f"], cfg.enc_ln_eps, f32=True)
        elif cfg.enc_kind == "t5":
            inner = A * cfg.enc_dkv
            bias = self._t5_position_bias(Cn)

            wide_t5 = cfg.enc_dkv > 64 or cfg.enc_dkv % 8 != 0  # t5-3b / t5-11b: d_kv = 128

            def attn_t5(qkv_hi, qkv_lo, qkv_f32=None):
                v = lambda t, i: t.view(B, Cn, 3 * inner)[:, :, i * inner:(i + 1) * inner]  # noqa: E731
                if wide_t5:
                    o = ops.attention_train(v(qkv_f32, 0), v(qkv_f32, 1), v(qkv_f32, 2), A, causal=False, key_mask=mask32, scale=1.0,
                                            pos_bias=bias)
                    return ops.split(o.view(B * Cn, inner))
                q, k, vv = ((v(qkv_hi, i), v(qkv_lo, i)) for i in range(3))
                o_hi, o_lo = ops.attention(q, k, vv, A, causal=False, key_mask=mask32, precision=P, scale=1.0, pos_bias=bias)
                return o_hi.view(B * Cn, inner), o_lo.view(B * Cn, inner)

            h = ops.embed_gather(input_ids, e["tok"]).view(B * Cn, E)
            for ly in e["layers"]:
                y_hi, y_lo, _ = ops.rmsnorm(h, ly["ln1"], cfg.enc_ln_eps)
                qkv_hi, qkv_lo, qkv32 = ly["qkv"](y_hi, y_lo, P, out_f32=wide_t5)
                a_hi, a_lo = attn_t5(qkv_hi, qkv_lo, qkv32)
                _, _, h = ly["out"](a_hi, a_lo, P, residual=h, out_f32=True, out_planes=False)
                y_hi, y_lo, _ = ops.rmsnorm(h, ly["ln2"], cfg.enc_ln_eps)
                if cfg.enc_gated_ffn:
                    _, _, g = ly["wi0"](y_hi, y_lo, P, act="gelu_new", out_f32=True, out_planes=False)
                    _, _, u = ly["wi1"](y_hi, y_lo, P, out_f32=True, out_planes=False)
                    f_hi, f_lo = ops.mul_split(g, u)
                else:
                    f_hi, f_lo, _ = ly["wi"](y_hi, y_lo, P, act="relu")
                _, _, h = ly["wo"](f_hi, f_lo, P, residual=h, out_f32=True, out_planes=False)
            c_hi, c_lo, ctx = ops.rmsnorm(h, e["ln_f"], cfg.enc_ln_eps, f32=True)
        else:
            x = ops.embed_gather(input_ids, e["tok"]).view(B * Cn, E)
            h_hi = torch.empty((B * Cn, E), dtype=torch.bfloat16, device=self.device)
            h_lo = torch.empty_like(h_hi)
            ops.step_embed_ln(_lib.STEP_NONE, B * Cn, Cn, E, X=x, time_embeds=e["zero_row"], t=0, pos_emb=e["pos"],
                              tok_type0=e["tok0"], ln_gamma=e["eln"][0], ln_beta=e["eln"][1], ln_eps=1e-12, h_hi=h_hi, h_lo=h_lo)
            ctx = None
            for ly in e["layers"]:
                qkv_hi, qkv_lo, qkv32 = ly["qkv"](h_hi, h_lo, P, out_f32=wide_heads)
                a_hi, a_lo = attn(qkv_hi, qkv_lo, qkv32)
                _, _, t = ly["out"](a_hi, a_lo, P, residual_planes=(h_hi, h_lo), out_f32=True, out_planes=False)
                h_hi, h_lo, _ = ops.layernorm(t, *ly["ln1"], 1e-12)
                f_hi, f_lo, _ = ly["fc1"](h_hi, h_lo, P, act="gelu")
                _, _, t = ly["fc2"](f_hi, f_lo, P, residual_planes=(h_hi, h_lo), out_f32=True, out_planes=False)
                h_hi, h_lo, ctx = ops.layernorm(t, *ly["ln2"], 1e-12, f32=True)
            c_hi, c_lo = h_hi, h_lo
        return c_hi, c_lo, ctx

    def _t5_position_bias(self, c):
        """[heads, C, C] fp32 relative position bias of T5Attention.compute_bias (HF:models/t5/modeling_t5.py:217-279),
        bidirectional buckets; index arithmetic on the host (C <= 96), one gather of the block-0 table on the device."""
        e, cfg = self._enc, self.cfg
        if c not in e["bias_cache"]:
            nb, max_exact = cfg.enc_rel_buckets // 2, cfg.enc_rel_buckets // 4
            rel = torch.arange(c)[None, :] - torch.arange(c)[:, None]  # memory position - query position
            rp = rel.abs()
            large = max_exact + (torch.log(rp.float() / max_exact) / math.log(cfg.enc_rel_max_distance / max_exact) * (nb - max_exact)).to(torch.long)
            large = torch.min(large, torch.full_like(large, nb - 1))
            bucket = (rel > 0).long() * nb + torch.where(rp < max_exact, rp, large)
            e["bias_cache"] = {c: e["rel"][bucket.to(self.device)].permute(2, 0, 1).contiguous()}
        return e["bias_cache"][c]

    # ------------------------------------------------------------------ buffers + hoisted cross K/V
    @property
    def prec(self):
        return ops.PRECISIONS[self.precision]

    @property
    def round_precision(self):
        """Precision of the classifier GEMM after the loop (rounding argmax, logits, cross-entropy): the fp16x2 mode spends
        all three products there (once per batch), as the engine's final rounding does."""
        return "fp16x3" if self.precision == "fp16x2" or self.schedule else self.precision

    def call_precisions(self, time_rows, mode="ddpm"):
        """C-ABI precision of every denoiser call of a loop, from the timestep (time-embedding row) of each and the sampler
        (``mode`` "euler" reads ``schedule_euler``); None = uniform."""
        if not self.schedule:
            return None
        sched = self.schedule_euler if mode == "euler" else self.schedule
        names = [next(p for t_above, p in sched if t > t_above) for t in time_rows]
        n = len(names)
        for k in range(n):
            from_end = n - 1 - k
            for floor_calls, p in self.tail_calls:
                if from_end < floor_calls and _RANK[p] > _RANK[names[k]]:
                    names[k] = p
        return [ops.PRECISIONS[p] for p in names]

    def round_operands(self, x0_hi, x0_lo):
        """(x_hi, x_lo or None, precision) for ops.round_argmax / ops.round_ce on x0 planes."""
        rp = self.round_precision
        return x0_hi, (x0_lo if rp in ("bf16x3", "fp16x3") else None), rp

    def buffers(self, B, S, Cn):
        if self.encoder_only:
            raise RuntimeError("this DiDiSampler was built with encoder_only=True: it has no decoder to sample with")
        key = (B, S, Cn)
        b = self._bufs.get(key)
        if b is None:
            dev, H = self.device, self.cfg.dec_dim
            ws_bytes = lib.didi_decoder_workspace_bytes(C.byref(self._dec), B, S, Cn)
            if ws_bytes == 0:
                raise _lib.DidiHipError(f"unsupported decoder shape B={B} S={S} C={Cn} H={H} heads={self.cfg.dec_heads}")
            kv_bytes = lib.didi_decoder_kv_plane_bytes(C.byref(self._dec), B, Cn)
            b = dict(
                ws=torch.empty(ws_bytes, dtype=torch.uint8, device=dev), ws_bytes=ws_bytes,
                kv_hi=torch.empty(kv_bytes, dtype=torch.uint8, device=dev),
                kv_lo=torch.empty(kv_bytes, dtype=torch.uint8, device=dev),
                x0=torch.empty((B, S, H), dtype=torch.float32, device=dev),
                x0_hi=torch.empty((B * S, H), dtype=self._pd, device=dev),
                x0_lo=torch.empty((B * S, H), dtype=self._pd, device=dev),
                x_state=torch.empty((B, S, H), dtype=torch.float32, device=dev),
                partial=torch.empty(lib.didi_round_argmax_bytes(B * S, self.cfg.vocab), dtype=torch.uint8, device=dev),
                fault=ops.new_fault(dev),  # device fault word (DIDI_FAULT_*), sticky until check_fault() reads and clears it
            )
            self._bufs = {key: b}  # keep one shape resident
            self._ctx_key = None
        return b

    def prepare_context(self, context, attention_mask, S):
        """Hoist the cross-attention K/V of ``context`` for all layers; cached per context tensor."""
        B, Cn, _ = context.shape
        b = self.buffers(B, S, Cn)
        key = (context.data_ptr(), context._version, attention_mask.data_ptr(), attention_mask._version, B, S, Cn)
        if self._ctx_key != key:
            ctx = context.contiguous().float()
            b["mask32"] = attention_mask.to(torch.int32).contiguous()
            b["count32"] = ops.key_counts(b["mask32"])  # right-padded contexts: padded keys are neither loaded nor multiplied
            _lib.check(lib.didi_decoder_prepare_context(C.byref(self._dec), ctx.data_ptr(), B, Cn, self.prec, b["ws"].data_ptr(),
                                                        b["ws_bytes"], b["kv_hi"].data_ptr(), b["kv_lo"].data_ptr(),
                                                        _lib.current_stream()), "prepare_context")
            self._ctx_key = key
            self._ctx_ref = (context, attention_mask)  # keep alive: the key holds raw addresses
        return b

    # ------------------------------------------------------------------ DiDi.forward
    def __call__(self, encoder_input_ids=None, encoder_attention_mask=None, decoder_inputs_embeds=None, time_ids=None,
                 context=None):
        if encoder_input_ids is None and context is None:
            raise ValueError("Either `encoder_input_ids` or `context` must be provided.")
        if context is None:
            context = self._encode_context(encoder_input_ids, encoder_attention_mask)
        x = decoder_inputs_embeds.to(self.device).contiguous().float()
        B, S, H = x.shape
        b = self.prepare_context(context, encoder_attention_mask, S)
        tidx = time_ids.reshape(-1).to(device=self.device, dtype=torch.int32).contiguous()
        if tidx.numel() == 1:
            tidx = tidx.expand(B).contiguous()
        out = torch.empty_like(x)
        fa = _lib.ForwardArgs()
        fa.B, fa.S, fa.C, fa.precision = B, S, context.shape[1], self.prec
        fa.step.mode, fa.step.X, fa.step.cin, fa.step.time_idx = _lib.STEP_NONE, x.data_ptr(), 1.0, tidx.data_ptr()
        fa.kv_hi, fa.kv_lo, fa.ctx_mask = b["kv_hi"].data_ptr(), b["kv_lo"].data_ptr(), b["mask32"].data_ptr()
        fa.ctx_count = b["count32"].data_ptr()
        fa.workspace, fa.workspace_bytes = b["ws"].data_ptr(), b["ws_bytes"]
        fa.x0_f32 = out.data_ptr()
        fa.fault = b["fault"].data_ptr()
        _lib.check(lib.didi_decoder_forward(C.byref(self._dec), C.byref(fa), _lib.current_stream()), "decoder_forward")
        return out, context

    forward = __call__

    def check_fault(self, what="didi"):
        """Synchronise and raise DidiHipError if any kernel since the last check reported a device-side fault
        (LayerNorm exchange timeout, fp16 range).  ``sample()`` calls it wherever it synchronises anyway; callers of
        ``return_ids=True`` / ``model(...)`` (no host sync on those paths) call it when they want the verdict."""
        for b in self._bufs.values():
            try:
                _lib.check_fault(b["fault"], what)
            finally:
                b["fault"].zero_()

    def eval(self):
        return self

    @classmethod
    def from_state_dict(cls, state_dict, cfg, **kw):
        return cls(state_dict, cfg, **kw)
