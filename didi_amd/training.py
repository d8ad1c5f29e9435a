"""``DiDi.training_step`` on the HIP path, FORWARD ONLY: the loss values of a step from a ``DiDiSampler``.

The whole step -- this forward with its activations kept, the hand-written backward through the decoder, the gradient all-reduce,
clipping and AdamW -- is ``didi_amd.trainer.DiDiTrainer``; this module is the light path for callers that want the five logged
scalars of a batch without training (validation-time loss, the parity tests of the noising and the loss terms) plus
``metric_message``, the one fp32 [5] message both paths reduce over ranks.

Reference: src/diffusion/model.py:200-243 (``training_step``), :62-63 (``flat_mean``), src/diffusion/utils.py:24-40 (``get_x0``,
``get_diffusion_variables``), src/metrics.py:91-93 (``calculate_batch_ce``).  What runs where:
  * target embedding gather, the noising (x_0 = emb + std_0 n0, x_t with a PER-DIALOG t), the denoiser forward with per-dialog
    time embeddings, the masked-MSE / t_T sums and both cross-entropy terms (classifier GEMM with a CE epilogue: the [B, S, V]
    logits never exist) are libdidi_hip kernels;
  * what is left to torch is scalar arithmetic on B-element vectors (the per-dialog means and the five logged scalars).
Reproduced as written: the mse's ``torch.where(t == 1, a, b)`` broadcasts ``t`` [B, 1] against per-dialog means [B] to [B, B], so
mse = p mean(a) + (1 - p) mean(b) with p the share of dialogs at t == 1; ``t_T`` averages over ALL elements (padding counts in
the denominator); a third normal draw (model.py:229) is consumed and discarded, so the generator ends where the reference's does.
``mse_backward`` is the loss gradient with respect to x_0_hat alone (checked against autograd in tests/golden/train_tiny_*).
"""
import torch

from . import ops


def _masked_mean_nll(nll, mask):
    m = mask.reshape(-1).to(nll.dtype)
    return (nll * m).sum() / m.sum()


@torch.no_grad()
def training_step_forward(model, raw_context, target, *, anchor_loss=False, noise_factor=1.0, draws=None, t=None):
    """-> dict(mse, ce, t_T, loss, ce_hat [0-dim device tensors], t, emb, x_0, x_t, x_0_hat).

    ``draws``: None = in-kernel Philox noise keyed from torch's global generator (reproducible under ``torch.manual_seed``), or the
    injected normal draws [n0, n1, ...] in the reference's order (parity tests).  ``t``: None = ``torch.randint(1, T + 1, [B, 1])``
    as the reference draws it, or an injected int64 [B, 1]."""
    dev = model.device
    ids, tmask = target.input_ids.to(dev), target.attention_mask.to(dev)
    B, S = ids.shape
    H, T = model.cfg.dec_dim, model.diffusion_steps
    emb = ops.embed_gather(ids, model.emb.weight)
    seed = 0
    if draws is None:
        seed = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
    if t is None:
        t = torch.randint(1, T + 1, size=(B, 1), device=dev)
    t = t.to(dev)
    sigma = model.sigmas[t.view(-1)]
    s_b = (sigma * float(noise_factor) ** 2).contiguous()
    c_b = (1.0 / torch.sqrt(sigma * sigma + 1.0)).contiguous()
    std0 = float(model.std_0)
    n0 = n1 = None
    if draws is not None:
        n0, n1 = draws[0].to(dev).float().contiguous(), draws[1].to(dev).float().contiguous()
    x_0, x_t = ops.train_noise(emb, s_b, c_b, std0, n0, n1, seed=seed)
    x_0_hat, _ = model(encoder_input_ids=raw_context.input_ids.to(dev), encoder_attention_mask=raw_context.attention_mask.to(dev),
                       decoder_inputs_embeds=x_t, time_ids=t)
    sums = ops.train_loss_terms(x_0_hat, emb, x_0, tmask)
    p = (t.view(-1) == 1).float().mean()
    per = sums / float(S * H)
    mse = p * per[:, 0].mean() + (1.0 - p) * per[:, 1].mean()
    sigma_T = model.sigmas[-1]
    t_T = sums[:, 2].sum() / (sigma_T * sigma_T + 1.0) / float(B * S * H)

    def ce_of(x):
        hi, lo = ops.split(x.reshape(B * S, H).contiguous(), dtype=model._pd)
        xh, xl, rp = model.round_operands(hi, lo)
        _, nll = ops.round_ce(xh, xl, model._cls.hi, model._cls.lo, ids, rp)
        return _masked_mean_nll(nll, tmask)

    ce_hat = ce_of(x_0_hat)
    ce = ce_hat if anchor_loss else ce_of(x_0)
    loss = mse + ce if anchor_loss else mse + ce + t_T
    return dict(mse=mse, ce=ce, t_T=t_T, loss=loss, ce_hat=ce_hat, t=t, emb=emb, x_0=x_0, x_t=x_t, x_0_hat=x_0_hat)


@torch.no_grad()
def mse_backward(x_0_hat, emb, x_0, target_mask, t):
    """d mse / d x_0_hat (model.py:222-227 with its [B, B] broadcast): fp32 [B, S, H]."""
    B, S, H = x_0_hat.shape
    p = float((t.view(-1) == 1).float().mean())
    return ops.train_mse_backward(x_0_hat, emb, x_0, target_mask, 2.0 * p / (B * S * H), 2.0 * (1.0 - p) / (B * S * H))


def metric_message(step_out):
    """The five scalars ``log_dict(..., sync_dist=True)`` reduces one by one (model.py:241-242), as ONE fp32 [5] message."""
    return torch.stack([step_out[k].float() for k in ("mse", "ce", "t_T", "loss", "ce_hat")])
