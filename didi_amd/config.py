"""Shape/config record for the DiDi denoiser as the sampler path needs it.

Mirrors what the reference spreads over ``get_components`` (src/diffusion/model.py:29-54: encoder
family + BertConfig decoder) and ``DiDi.__init__`` (model.py:66-134: diffusion_steps, schedule,
euler knobs).  Field defaults are the reference's defaults (BertConfig defaults for the decoder,
configs/train.yaml:25-31 for the diffusion section).
"""
from dataclasses import asdict, dataclass


@dataclass
class DiDiConfig:
    vocab: int = 8008
    # decoder = transformers BertConfig(is_decoder=True, add_cross_attention=True) defaults
    dec_dim: int = 768
    dec_layers: int = 12
    dec_heads: int = 12
    dec_ffn: int = 3072
    dec_max_pos: int = 512
    dec_ln_eps: float = 1e-12
    # context encoder: "blenderbot" (pre-LN), "bert" (post-LN) or "t5" (T5EncoderModel: RMSNorm, relative position bias)
    enc_kind: str = "blenderbot"
    enc_dim: int = 1280
    enc_layers: int = 2
    enc_heads: int = 32
    enc_ffn: int = 5120
    enc_max_pos: int = 128
    enc_scale_embedding: bool = True
    enc_ln_eps: float = 1e-5
    # t5 only: per-head dim (heads * d_kv need not equal enc_dim), relative-position table, and whether the FFN is the
    # v1.1 gated form (gelu_new(x wi_0) * (x wi_1)) or the original ReLU one
    enc_dkv: int = 64
    enc_rel_buckets: int = 32
    enc_rel_max_distance: int = 128
    enc_gated_ffn: bool = False
    pad_idx: int = 0
    # diffusion section (configs/train.yaml:25-31, model.py:85-87)
    diffusion_steps: int = 2000
    schedule: str = "sqrt"
    sampling_mode: str = "ddpm"
    step_freq: int = 1
    noise_factor: float = 1.0
    s_churn: float = 0.0
    s_tmin: float = 0.0
    s_tmax: float = float("inf")
    tie_weights: bool = False

    @property
    def has_adapter(self) -> bool:  # model.py:122
        return self.enc_dim != self.dec_dim

    def to_dict(self):
        return asdict(self)


def blenderbot_400m_distill(**kw) -> DiDiConfig:
    """BASELINE config 2/3: blenderbot-400M-distill-sized encoder, BertConfig-default decoder."""
    return DiDiConfig(**kw)


def tiny_config1(**kw) -> DiDiConfig:
    """BASELINE config 1: 2-layer encoder/decoder, d=128, 4 heads, FFN 512, V=8008, T=50."""
    base = dict(vocab=8008, dec_dim=128, dec_layers=2, dec_heads=4, dec_ffn=512, enc_dim=128, enc_layers=2,
                enc_heads=4, enc_ffn=512, diffusion_steps=50)
    base.update(kw)
    return DiDiConfig(**base)
