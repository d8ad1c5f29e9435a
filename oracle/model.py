"""Oracle: a DiDi-shaped denoiser object (fp32 CPU) over a reference-layout state dict.

Restates /root/reference/src/diffusion/model.py:104-127 (attributes the sampler reads) and
:165-198 (``_encode_context`` + ``forward``) without Lightning.  TEST INFRASTRUCTURE.
"""
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from .diffusion_math import configure_schedule
from .transformer import bert_decoder, bert_encoder_model, blenderbot_encoder, t5_encoder


@dataclass
class OracleConfig:
    vocab: int
    dec_dim: int = 768
    dec_layers: int = 12
    dec_heads: int = 12
    enc_kind: str = "blenderbot"  # "blenderbot" | "bert" | "t5"
    enc_dim: int = 1280
    enc_layers: int = 2
    enc_heads: int = 32
    enc_scale_embedding: bool = True
    diffusion_steps: int = 2000
    schedule: str = "sqrt"
    s_churn: float = 0.0
    s_tmin: float = 0.0
    s_tmax: float = float("inf")
    dec_ln_eps: float = 1e-12
    enc_ln_eps: float = 1e-5  # blenderbot nn.LayerNorm default; bert encoder uses 1e-12; t5 1e-6
    enc_rel_buckets: int = 32  # t5 relative_attention_num_buckets / relative_attention_max_distance
    enc_rel_max_distance: int = 128


class _Table:
    """nn.Embedding / bias-free nn.Linear stand-ins exposing ``.weight`` and ``__call__``."""

    def __init__(self, weight, kind):
        self.weight, self.kind = weight, kind

    def __call__(self, x):
        return F.embedding(x, self.weight) if self.kind == "emb" else F.linear(x, self.weight)


class OracleDiDi:
    def __init__(self, state_dict, cfg: OracleConfig):
        self.sd = {k: v.detach().to(torch.float32) if v.is_floating_point() else v for k, v in state_dict.items()}
        self.cfg = cfg
        self.diffusion_steps = cfg.diffusion_steps
        self.s_churn, self.s_tmin, self.s_tmax = cfg.s_churn, cfg.s_tmin, cfg.s_tmax
        self.emb = _Table(self.sd["emb.weight"], "emb")  # model.py:104
        self.time_embeds = _Table(self.sd["time_embeds.weight"], "emb")  # model.py:105
        self.classifier = _Table(self.sd["classifier.weight"], "lin")  # model.py:118
        if "sigmas" in self.sd:
            self.sigmas = self.sd["sigmas"]
        else:
            self.sigmas, _ = configure_schedule(cfg.diffusion_steps, cfg.schedule)  # model.py:125-126
        self.has_adapter = "adapter.0.weight" in self.sd  # model.py:122-123

    def encode_context(self, input_ids, attention_mask):
        c = self.cfg
        if c.enc_kind == "blenderbot":
            ctx = blenderbot_encoder(
                self.sd, "encoder.", input_ids, attention_mask, c.enc_layers, c.enc_heads, c.enc_scale_embedding, c.enc_ln_eps
            )
        elif c.enc_kind == "bert":
            ctx = bert_encoder_model(self.sd, "encoder.", input_ids, attention_mask, c.enc_layers, c.enc_heads, 1e-12)
        elif c.enc_kind == "t5":
            ctx = t5_encoder(self.sd, "encoder.", input_ids, attention_mask, c.enc_layers, c.enc_heads, c.enc_rel_buckets,
                             c.enc_rel_max_distance, c.enc_ln_eps)
        else:
            raise ValueError(f"Unsupported encoder kind: {c.enc_kind}")
        if self.has_adapter:  # Linear -> Tanh -> Linear, model.py:123,168-169
            ctx = F.linear(ctx, self.sd["adapter.0.weight"], self.sd["adapter.0.bias"])
            ctx = F.linear(torch.tanh(ctx), self.sd["adapter.2.weight"], self.sd["adapter.2.bias"])
        return ctx

    def __call__(
        self, encoder_input_ids=None, encoder_attention_mask=None, decoder_inputs_embeds=None, time_ids=None, context=None
    ):
        # model.py:172-198
        if encoder_input_ids is None and context is None:
            raise ValueError("Either `encoder_input_ids` or `context` must be provided.")
        if context is None:
            context = self.encode_context(encoder_input_ids, encoder_attention_mask)
        x = decoder_inputs_embeds + self.time_embeds(time_ids)
        out = bert_decoder(
            self.sd, "decoder.", x, context, encoder_attention_mask, self.cfg.dec_layers, self.cfg.dec_heads, self.cfg.dec_ln_eps
        )
        return out, context
