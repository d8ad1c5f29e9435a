"""``DiDi.validation_step`` (src/diffusion/model.py:245-254) on the HIP path.

The reference samples with ``raw_output=True``, materialises the [B, S, V] logits (525 MB at BASELINE config 2,
17 GB at config 4), and takes ``argmax`` and ``calculate_batch_ce`` (src/metrics.py:91-93) from them.  Here the
classifier GEMM carries the argmax AND an online log-sum-exp + target-logit pick in its epilogue
(``didi_round_ce``): the logits never exist in memory; only per-token (id, nll) come out, and the masked means are
two tiny torch reductions.
"""
import torch

from . import ops
from .sampling import sample


@torch.no_grad()
def validation_step(model, raw_context, target, mode=None, step_freq=None, **sample_kw):
    """-> (ce, accuracy) as the reference appends to ``val_ce`` / ``val_acc`` (model.py:251-254).

    ``target``: object with ``.input_ids`` / ``.attention_mask`` [B, S_target] (the tokenised replies).  ``mode`` and
    ``step_freq`` default to the model's ``sampling_mode`` / ``step_freq`` attributes (model.py:248).
    """
    mode = mode if mode is not None else getattr(model, "sampling_mode", "ddpm")
    step_freq = step_freq if step_freq is not None else getattr(model, "step_freq", 1)
    t_ids = target.input_ids.to(model.device)
    t_mask = target.attention_mask.to(model.device)
    max_trg_len = t_ids.shape[1]
    x0 = sample(raw_context, model, mode, step_freq, max_len=max_trg_len, return_x0=True, **sample_kw)
    B, S, H = x0.shape
    if S != max_trg_len:  # the reference's `predictions == target.input_ids` would fail to broadcast
        raise ValueError(f"target length {max_trg_len} exceeds the sampled length {S} (context length caps it)")
    ids, nll = token_ce(model, x0, t_ids)
    keep = t_mask.reshape(-1).bool()
    ce = nll[keep].mean()  # F.cross_entropy(logits[non_pad], target[non_pad]): mean over non-pad tokens
    acc = ((ids.view(B, S) == t_ids) * t_mask).sum() / t_mask.sum()
    return ce.item(), acc.item()


def token_ce(model, x0, target_ids):
    """ids [M], nll [M] of ``classifier(x0)`` against ``target_ids``, logits never materialised."""
    H = x0.shape[-1]
    hi, lo = ops.split(x0.reshape(-1, H).contiguous(), dtype=model._pd)
    xh, xl, rp = model.round_operands(hi, lo)
    return ops.round_ce(xh, xl, model._cls.hi, model._cls.lo, target_ids, rp)
