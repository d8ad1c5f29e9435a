"""Lightning checkpoint -> DiDiSampler.

The reference saves with ``ModelCheckpoint`` (src/training.py:27-33) and reloads with
``DiDi.load_from_checkpoint(path, enc_dim, dec_dim, map_location)`` (scripts/sample.py:58).  A
``.ckpt`` is a dict with ``state_dict`` (key layout: SURVEY.md section 5) and ``hyper_parameters``
(``save_hyperparameters(ignore=[encoder, decoder])``, src/diffusion/model.py:91).  The shapes in the
state dict determine the architecture; what shapes cannot tell (attention head counts, LayerNorm
eps) follows the reference's ``get_components`` choices (model.py:29-54) and can be overridden.

Files are read with ``torch.load(..., weights_only=True)`` only: nothing in a checkpoint is executed.
"""
import re

import torch

from .config import DiDiConfig

# hyper_parameters the sampler reads (DiDi.__init__ keyword names -> DiDiConfig fields)
_HPARAMS = {
    "diffusion_steps": "diffusion_steps", "schedule": "schedule", "step_freq": "step_freq", "pad_idx": "pad_idx",
    "noise_factor": "noise_factor", "tie_weights": "tie_weights", "sampling_mode": "sampling_mode",
    "s_churn": "s_churn", "s_tmin": "s_tmin", "s_tmax": "s_tmax",
}


def _n_layers(sd, pattern):
    idx = {int(m.group(1)) for k in sd for m in [re.match(pattern, k)] if m}
    return max(idx) + 1 if idx else 0


def config_from_state_dict(sd, **overrides) -> DiDiConfig:
    """Infer the DiDiConfig of a reference state dict from its tensor shapes."""
    c = {}
    c["vocab"], c["dec_dim"] = sd["emb.weight"].shape
    c["diffusion_steps"] = sd["time_embeds.weight"].shape[0] - 1
    c["dec_layers"] = _n_layers(sd, r"decoder\.encoder\.layer\.(\d+)\.")
    c["dec_ffn"] = sd["decoder.encoder.layer.0.intermediate.dense.weight"].shape[0]
    c["dec_max_pos"] = sd["decoder.embeddings.position_embeddings.weight"].shape[0]
    c["dec_heads"] = max(1, c["dec_dim"] // 64)  # BertConfig: 12 heads at 768
    if "encoder.layers.0.self_attn.q_proj.weight" in sd:  # BlenderbotEncoder (model.py:31-33)
        c["enc_kind"] = "blenderbot"
        c["enc_dim"] = sd["encoder.embed_tokens.weight"].shape[1]
        c["enc_layers"] = _n_layers(sd, r"encoder\.layers\.(\d+)\.")
        c["enc_ffn"] = sd["encoder.layers.0.fc1.weight"].shape[0]
        c["enc_max_pos"] = sd["encoder.embed_positions.weight"].shape[0]
        c["enc_heads"] = max(1, c["enc_dim"] // 40)  # blenderbot-400M-distill: 32 heads at 1280
        c["enc_ln_eps"] = 1e-5
    elif "encoder.encoder.layer.0.attention.self.query.weight" in sd:  # BertModel (model.py:36-38)
        c["enc_kind"] = "bert"
        c["enc_dim"] = sd["encoder.embeddings.word_embeddings.weight"].shape[1]
        c["enc_layers"] = _n_layers(sd, r"encoder\.encoder\.layer\.(\d+)\.")
        c["enc_ffn"] = sd["encoder.encoder.layer.0.intermediate.dense.weight"].shape[0]
        c["enc_max_pos"] = sd["encoder.embeddings.position_embeddings.weight"].shape[0]
        c["enc_heads"] = max(1, c["enc_dim"] // 64)
        c["enc_ln_eps"] = 1e-12
    elif "encoder.encoder.block.0.layer.0.SelfAttention.q.weight" in sd:  # T5EncoderModel (model.py:39-42)
        q = "encoder.encoder.block.0.layer."
        rel = sd[q + "0.SelfAttention.relative_attention_bias.weight"]  # [buckets, heads]
        inner, c["enc_dim"] = sd[q + "0.SelfAttention.q.weight"].shape
        c["enc_kind"] = "t5"
        c["enc_layers"] = _n_layers(sd, r"encoder\.encoder\.block\.(\d+)\.")
        c["enc_rel_buckets"], c["enc_heads"] = rel.shape
        c["enc_dkv"] = inner // c["enc_heads"]
        c["enc_ffn"] = sd[q + "1.DenseReluDense.wo.weight"].shape[1]
        c["enc_gated_ffn"] = q + "1.DenseReluDense.wi_0.weight" in sd
        c["enc_ln_eps"] = 1e-6
    else:
        raise ValueError("unrecognised context encoder in the state dict")
    has_adapter = "adapter.0.weight" in sd
    if has_adapter != (c["enc_dim"] != c["dec_dim"]):
        raise ValueError("adapter weights and encoder/decoder widths disagree (model.py:122)")
    cls = sd.get("classifier.weight")
    c["tie_weights"] = cls is None or cls.data_ptr() == sd["emb.weight"].data_ptr() or torch.equal(cls, sd["emb.weight"])
    c.update(overrides)
    return DiDiConfig(**c)


def load_state(path):
    """(state_dict, hyper_parameters) of a Lightning ``.ckpt`` or of a bare state-dict file."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(ckpt, dict) and "state_dict" in ckpt:
        return ckpt["state_dict"], dict(ckpt.get("hyper_parameters") or {})
    return ckpt, {}


def load_lightning_checkpoint(path, device="cuda", precision="fp16x2", **overrides):
    """``DiDi.load_from_checkpoint`` for the sampler path -> DiDiSampler."""
    from .model import DiDiSampler

    sd, hp = load_state(path)
    known = {field: hp[name] for name, field in _HPARAMS.items() if name in hp and isinstance(hp[name], (int, float, str, bool))}
    if "diffusion_steps" in known and known["diffusion_steps"] + 1 != sd["time_embeds.weight"].shape[0]:
        raise ValueError("hyper_parameters.diffusion_steps does not match time_embeds.weight")
    known.update(overrides)
    cfg = config_from_state_dict(sd, **known)
    return DiDiSampler(sd, cfg, device=device, precision=precision)
