"""Synthetic weights and CommonSense-shaped batches (there is no network for checkpoints/datasets).

State-dict keys follow the reference's Lightning checkpoint layout:
``emb.weight, time_embeds.weight, classifier.weight, adapter.{0,2}.*, sigmas, std_0, encoder.*,
decoder.embeddings.*, decoder.encoder.layer.{i}.*``.  ``decoder.embeddings.word_embeddings`` and
``decoder.pooler`` are never read on the sampler path (inputs arrive as ``inputs_embeds``; the
pooled output is discarded, model.py:197) and are not generated.

Init: N(0, 0.02^2) for every matrix and table (HF ``initializer_range``), biases N(0, 0.02^2),
LayerNorm gamma 1 + N(0, 0.05^2), beta N(0, 0.02^2) -- non-trivial gamma/beta/bias so that a kernel
that drops them fails parity.  Generated on CPU with a seeded generator: bit-reproducible.
Batch shape follows BASELINE.md section 3: ids ~ U{3..V-1}, lengths clip(round(N(24,12^2)),2,C),
right-padded with pad id 0 (commonsense_dataset.py:27-32 pads to max_length).
"""
import torch

from .config import DiDiConfig
from .schedule import configure_schedule


def _n(g, *shape, std=0.02):
    return torch.randn(*shape, generator=g) * std


def _linear(sd, g, name, n_out, n_in, bias=True):
    sd[name + ".weight"] = _n(g, n_out, n_in)
    if bias:
        sd[name + ".bias"] = _n(g, n_out)


def _ln(sd, g, name, dim):
    sd[name + ".weight"] = 1.0 + _n(g, dim, std=0.05)
    sd[name + ".bias"] = _n(g, dim)


def _bert_layer(sd, g, p, H, I, cross):
    for blk in (["attention"] + (["crossattention"] if cross else [])):
        for nm in ("query", "key", "value"):
            _linear(sd, g, f"{p}{blk}.self.{nm}", H, H)
        _linear(sd, g, f"{p}{blk}.output.dense", H, H)
        _ln(sd, g, f"{p}{blk}.output.LayerNorm", H)
    _linear(sd, g, p + "intermediate.dense", I, H)
    _linear(sd, g, p + "output.dense", H, I)
    _ln(sd, g, p + "output.LayerNorm", H)


def synthetic_state_dict(cfg: DiDiConfig, seed: int = 1234) -> dict:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    H, V = cfg.dec_dim, cfg.vocab
    sd["emb.weight"] = _n(g, V, H)
    sd["emb.weight"][cfg.pad_idx] = 0  # nn.Embedding(padding_idx) zeroes that row (model.py:104)
    sd["time_embeds.weight"] = _n(g, cfg.diffusion_steps + 1, H)
    sd["classifier.weight"] = sd["emb.weight"] if cfg.tie_weights else _n(g, V, H)
    if cfg.has_adapter:
        _linear(sd, g, "adapter.0", H, cfg.enc_dim)
        _linear(sd, g, "adapter.2", H, H)
    sigmas, std_0 = configure_schedule(cfg.diffusion_steps, cfg.schedule)
    sd["sigmas"], sd["std_0"] = sigmas, std_0
    # decoder
    sd["decoder.embeddings.position_embeddings.weight"] = _n(g, cfg.dec_max_pos, H)
    sd["decoder.embeddings.token_type_embeddings.weight"] = _n(g, 2, H)
    _ln(sd, g, "decoder.embeddings.LayerNorm", H)
    for i in range(cfg.dec_layers):
        _bert_layer(sd, g, f"decoder.encoder.layer.{i}.", H, cfg.dec_ffn, cross=True)
    # encoder
    E = cfg.enc_dim
    if cfg.enc_kind == "blenderbot":
        sd["encoder.embed_tokens.weight"] = _n(g, V, E)
        sd["encoder.embed_tokens.weight"][cfg.pad_idx] = 0
        sd["encoder.embed_positions.weight"] = _n(g, cfg.enc_max_pos, E)
        for i in range(cfg.enc_layers):
            p = f"encoder.layers.{i}."
            for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
                _linear(sd, g, p + "self_attn." + nm, E, E)
            _ln(sd, g, p + "self_attn_layer_norm", E)
            _linear(sd, g, p + "fc1", cfg.enc_ffn, E)
            _linear(sd, g, p + "fc2", E, cfg.enc_ffn)
            _ln(sd, g, p + "final_layer_norm", E)
        _ln(sd, g, "encoder.layer_norm", E)
    elif cfg.enc_kind == "bert":
        sd["encoder.embeddings.word_embeddings.weight"] = _n(g, V, E)
        sd["encoder.embeddings.word_embeddings.weight"][cfg.pad_idx] = 0
        sd["encoder.embeddings.position_embeddings.weight"] = _n(g, cfg.dec_max_pos, E)
        sd["encoder.embeddings.token_type_embeddings.weight"] = _n(g, 2, E)
        _ln(sd, g, "encoder.embeddings.LayerNorm", E)
        for i in range(cfg.enc_layers):
            _bert_layer(sd, g, f"encoder.encoder.layer.{i}.", E, cfg.enc_ffn, cross=False)
    else:
        raise ValueError(f"Unsupported encoder kind: {cfg.enc_kind}")
    return sd


def synthetic_batch(batch: int, ctx_len: int, vocab: int, seed: int = 0, mean_len: float = 24.0, std_len: float = 12.0):
    """-> (input_ids [B,C] int64, attention_mask [B,C] int64), CPU tensors."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, vocab, (batch, ctx_len), generator=g)
    lens = torch.clamp(torch.round(mean_len + std_len * torch.randn(batch, generator=g)), 2, ctx_len).long()
    mask = (torch.arange(ctx_len)[None, :] < lens[:, None]).long()
    return ids * mask, mask


class RawContext:
    """What ``sample`` reads from a HF BatchEncoding: ``.input_ids`` and ``.attention_mask``."""

    def __init__(self, input_ids, attention_mask):
        self.input_ids, self.attention_mask = input_ids, attention_mask

    def to(self, device):
        return RawContext(self.input_ids.to(device), self.attention_mask.to(device))
