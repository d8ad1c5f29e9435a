"""Oracle: the third-party transformer arithmetic on the path, restated functionally.

The reference's denoiser is ``transformers.BertModel(is_decoder=True, add_cross_attention=True)``
(/root/reference/src/diffusion/model.py:44-52,193-197) and its context encoder is a HF encoder
(model.py:31-42,165-170).  transformers is pinned ``~=4.30.2`` in the reference's
requirements.txt:3 and is NOT under /root/reference; this file restates the published algorithm
following the installed 5.15.0 sources (``HF:`` = site-packages/transformers):

  bert_embeddings     HF:models/bert/modeling_bert.py:53-108   (inputs_embeds + token_type[0] + pos -> LN)
  bert_layer          HF:models/bert/modeling_bert.py:139-416  (post-LN; causal self-attn; cross-attn; erf-GELU FFN)
  bert_decoder        HF:models/bert/modeling_bert.py:594-718  (pooler output is discarded by the caller -> skipped)
  bert_encoder_model  same file, non-decoder BertModel used as the context encoder for "bert" base names
  blenderbot_encoder  HF:models/blenderbot/modeling_blenderbot.py:65-300,424-512 (pre-LN, learned pos, final LN)
  t5_encoder          HF:models/t5/modeling_t5.py (T5LayerNorm, T5DenseActDense / T5DenseGatedActDense, T5Attention with
                      the bucketed relative position bias :217-279, T5Stack): "t5" base names, model.py:39-42

Masks follow the eager 4.30.2 formulation (additive ``finfo.min``): a fully masked key row gives
a uniform softmax, never NaN.  All weights are read from a flat state dict with the reference's
checkpoint key layout.
TEST INFRASTRUCTURE: see oracle/__init__.py.
"""
import math

import torch
import torch.nn.functional as F


def _lin(sd, prefix, x):
    return F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))


def _ln(sd, prefix, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def _heads(x, n_heads):
    b, s, h = x.shape
    return x.view(b, s, n_heads, h // n_heads).transpose(1, 2)


def _attend(q, k, v, add_mask):
    """softmax(q k^T / sqrt(d) + mask) v on [B, A, S, d] tensors -> [B, S, A*d]."""
    d = q.shape[-1]
    scores = torch.matmul(q, k.transpose(2, 3)) * (d**-0.5)
    if add_mask is not None:
        scores = scores + add_mask
    p = torch.softmax(scores, dim=-1)
    o = torch.matmul(p, v).transpose(1, 2)
    return o.reshape(o.shape[0], o.shape[1], -1)


def key_padding_mask(attention_mask, dtype=torch.float32):
    """[B, C] 0/1 -> additive [B, 1, 1, C]  (HF get_extended_attention_mask / invert_attention_mask)."""
    m = attention_mask[:, None, None, :].to(dtype)
    return (1.0 - m) * torch.finfo(dtype).min


def causal_mask(s, dtype=torch.float32):
    """additive [1, 1, S, S]: query i sees keys j <= i (is_decoder=True, HF:bert:696-702)."""
    allowed = torch.tril(torch.ones(s, s, dtype=torch.bool))
    m = torch.zeros(s, s, dtype=dtype)
    m.masked_fill_(~allowed, torch.finfo(dtype).min)
    return m[None, None]


def bert_embeddings(sd, p, inputs_embeds, eps):
    s = inputs_embeds.shape[1]
    x = inputs_embeds + sd[p + "token_type_embeddings.weight"][0]
    x = x + sd[p + "position_embeddings.weight"][:s]
    return _ln(sd, p + "LayerNorm", x, eps)


def bert_layer(sd, p, h, n_heads, eps, self_mask, ctx=None, ctx_mask=None):
    a = p + "attention."
    q = _heads(_lin(sd, a + "self.query", h), n_heads)
    k = _heads(_lin(sd, a + "self.key", h), n_heads)
    v = _heads(_lin(sd, a + "self.value", h), n_heads)
    o = _attend(q, k, v, self_mask)
    h = _ln(sd, a + "output.LayerNorm", _lin(sd, a + "output.dense", o) + h, eps)
    if ctx is not None:
        c = p + "crossattention."
        q = _heads(_lin(sd, c + "self.query", h), n_heads)
        k = _heads(_lin(sd, c + "self.key", ctx), n_heads)
        v = _heads(_lin(sd, c + "self.value", ctx), n_heads)
        o = _attend(q, k, v, ctx_mask)
        h = _ln(sd, c + "output.LayerNorm", _lin(sd, c + "output.dense", o) + h, eps)
    inter = F.gelu(_lin(sd, p + "intermediate.dense", h))  # hidden_act="gelu": exact erf form
    return _ln(sd, p + "output.LayerNorm", _lin(sd, p + "output.dense", inter) + h, eps)


def bert_decoder(sd, prefix, inputs_embeds, context, context_attention_mask, n_layers, n_heads, eps=1e-12):
    """The denoiser body: [B,S,H] noised target (+time embedding already added) -> [B,S,H]."""
    h = bert_embeddings(sd, prefix + "embeddings.", inputs_embeds, eps)
    self_mask = causal_mask(h.shape[1], h.dtype)
    ctx_mask = key_padding_mask(context_attention_mask, h.dtype) if context_attention_mask is not None else None
    for i in range(n_layers):
        h = bert_layer(sd, f"{prefix}encoder.layer.{i}.", h, n_heads, eps, self_mask, context, ctx_mask)
    return h


def bert_encoder_model(sd, prefix, input_ids, attention_mask, n_layers, n_heads, eps=1e-12):
    """Non-causal BertModel as context encoder ("bert" base names, model.py:31-35)."""
    p = prefix + "embeddings."
    x = F.embedding(input_ids, sd[p + "word_embeddings.weight"])
    h = bert_embeddings(sd, p, x, eps)
    mask = key_padding_mask(attention_mask, h.dtype)
    for i in range(n_layers):
        h = bert_layer(sd, f"{prefix}encoder.layer.{i}.", h, n_heads, eps, mask)
    return h


def blenderbot_encoder(sd, prefix, input_ids, attention_mask, n_layers, n_heads, scale_embedding=True, eps=1e-5):
    """Pre-LN encoder with learned positions and a final LayerNorm (HF:blenderbot:424-512)."""
    d = sd[prefix + "embed_tokens.weight"].shape[1]
    x = F.embedding(input_ids, sd[prefix + "embed_tokens.weight"])
    if scale_embedding:
        x = x * math.sqrt(d)
    h = x + sd[prefix + "embed_positions.weight"][: input_ids.shape[1]]
    mask = key_padding_mask(attention_mask, h.dtype)
    for i in range(n_layers):
        p = f"{prefix}layers.{i}."
        r = h
        y = _ln(sd, p + "self_attn_layer_norm", h, eps)
        q = _heads(_lin(sd, p + "self_attn.q_proj", y), n_heads)
        k = _heads(_lin(sd, p + "self_attn.k_proj", y), n_heads)
        v = _heads(_lin(sd, p + "self_attn.v_proj", y), n_heads)
        h = r + _lin(sd, p + "self_attn.out_proj", _attend(q, k, v, mask))
        r = h
        y = _ln(sd, p + "final_layer_norm", h, eps)
        h = r + _lin(sd, p + "fc2", F.gelu(_lin(sd, p + "fc1", y)))
    return _ln(sd, prefix + "layer_norm", h, eps)


def t5_relative_position_bucket(relative_position, num_buckets=32, max_distance=128):
    """Bidirectional bucketing of (memory_pos - query_pos) (HF:t5:217-262)."""
    num_buckets //= 2
    buckets = (relative_position > 0).to(torch.long) * num_buckets
    rp = relative_position.abs()
    max_exact = num_buckets // 2
    is_small = rp < max_exact
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(is_small, rp, large)


def t5_position_bias(table, q_len, k_len, num_buckets=32, max_distance=128):
    """[heads, q_len, k_len] from the block-0 ``relative_attention_bias.weight`` [num_buckets, heads] (HF:t5:264-279)."""
    rel = torch.arange(k_len)[None, :] - torch.arange(q_len)[:, None]
    return table[t5_relative_position_bucket(rel, num_buckets, max_distance)].permute(2, 0, 1).contiguous()


def _rms(sd, key, x, eps):
    """T5LayerNorm: no mean subtraction, no bias; variance in fp32."""
    return sd[key] * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def t5_encoder(sd, prefix, input_ids, attention_mask, n_layers, n_heads, num_buckets=32, max_distance=128, eps=1e-6):
    """T5EncoderModel.last_hidden_state.  ``prefix`` addresses the T5EncoderModel ("encoder." in a DiDi checkpoint), whose
    stack lives under ``encoder.``; the gated FFN (v1.1) is recognised by its ``wi_0`` weight.  No 1/sqrt(d) score scaling."""
    st = prefix + "encoder."
    h = F.embedding(input_ids, sd[st + "embed_tokens.weight"])
    c = input_ids.shape[1]
    bias = t5_position_bias(sd[st + "block.0.layer.0.SelfAttention.relative_attention_bias.weight"], c, c, num_buckets, max_distance)
    add = bias[None] + key_padding_mask(attention_mask, h.dtype)
    for i in range(n_layers):
        a = f"{st}block.{i}.layer.0."
        y = _rms(sd, a + "layer_norm.weight", h, eps)
        q, k, v = (F.linear(y, sd[a + f"SelfAttention.{n}.weight"]) for n in "qkv")
        inner = q.shape[-1]
        q, k, v = (_heads(t, n_heads) for t in (q, k, v))
        p = torch.softmax(torch.matmul(q, k.transpose(2, 3)) + add, dim=-1)
        o = torch.matmul(p, v).transpose(1, 2).reshape(h.shape[0], c, inner)
        h = h + F.linear(o, sd[a + "SelfAttention.o.weight"])
        f = f"{st}block.{i}.layer.1."
        y = _rms(sd, f + "layer_norm.weight", h, eps)
        if f + "DenseReluDense.wi_0.weight" in sd:  # T5DenseGatedActDense, gelu_new
            z = F.gelu(F.linear(y, sd[f + "DenseReluDense.wi_0.weight"]), approximate="tanh") * F.linear(y, sd[f + "DenseReluDense.wi_1.weight"])
        else:  # T5DenseActDense, relu
            z = torch.relu(F.linear(y, sd[f + "DenseReluDense.wi.weight"]))
        h = h + F.linear(z, sd[f + "DenseReluDense.wo.weight"])
    return _rms(sd, st + "final_layer_norm.weight", h, eps)
