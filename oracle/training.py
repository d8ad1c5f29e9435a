"""Oracle: ``DiDi.training_step`` (fp32 CPU), forward and -- through torch autograd over the functional restatement -- backward.

Restates /root/reference/src/diffusion/model.py:200-243 with ``flat_mean`` (:62-63), ``calculate_batch_ce``
(src/metrics.py:91-93) and ``get_x0`` / ``get_diffusion_variables`` (src/diffusion/utils.py:24-40).  Every random draw is
injected: ``draws`` = [get_x0's randn_like, the noise randn_like, (the dead randn_like of :229, unused)] and ``t`` [B, 1], in the
order the reference consumes its generator.  Reproduced as written: ``torch.where(t == 1, a, b)`` with ``t`` [B, 1] against
per-row means [B] broadcasts to [B, B], so the mse is  p * mean(a) + (1 - p) * mean(b)  with p the share of rows at t == 1;
``t_T`` uses ``.mean()`` over ALL elements (padding included in the denominator).
Pinned by tests/golden/train_tiny_*.safetensors (oracle/make_golden.py train).  TEST INFRASTRUCTURE: see oracle/__init__.py.
"""
import torch
import torch.nn.functional as F

from .diffusion_math import get_diffusion_variables, get_x0, scale_input
from .model import OracleDiDi


def calculate_batch_ce(logits, target, non_pad_mask):
    # src/metrics.py:91-93
    non_pad_mask = non_pad_mask.bool().view(-1)
    return F.cross_entropy(logits.view(-1, logits.shape[-1])[non_pad_mask], target.view(-1)[non_pad_mask])


def flat_mean(tensor):
    # model.py:62-63
    return tensor.mean(dim=list(range(1, len(tensor.shape))))


def trainable(name):
    """Parameters the reference trains: everything but the frozen encoder (configs/train.yaml:19, model.py:114-115) and buffers."""
    return not name.startswith("encoder.") and name not in ("sigmas", "std_0") and "position_ids" not in name


def training_step(model: OracleDiDi, raw_context, target, draws, t, anchor_loss=False, noise_factor=1.0):
    """-> dict(emb, x_0, x_t, x_0_hat, mse, ce, t_T, loss, ce_hat).  model.py:200-243."""
    emb = model.emb(target.input_ids)
    x_0 = get_x0(emb, model.sd["std_0"], draws[0])
    noise = draws[1]
    x_t, t = get_diffusion_variables(x_0, model.sigmas, noise, t, noise_factor)
    x_0_hat, _ = model(encoder_input_ids=raw_context.input_ids, encoder_attention_mask=raw_context.attention_mask,
                       decoder_inputs_embeds=x_t, time_ids=t)
    logits = model.classifier(x_0_hat) if anchor_loss else model.classifier(x_0)
    ce = calculate_batch_ce(logits, target.input_ids, target.attention_mask)
    non_pad_mask = target.attention_mask.unsqueeze(-1)
    mse = torch.where(t == 1, flat_mean((x_0_hat - emb) ** 2 * non_pad_mask), flat_mean((x_0_hat - x_0) ** 2 * non_pad_mask)).mean()
    x_T = scale_input(x_0, model.sigmas[-1])
    t_T_loss = (x_T**2 * non_pad_mask).mean()
    loss = mse + ce
    if not anchor_loss:
        loss = loss + t_T_loss
    with torch.no_grad():
        ce_hat = calculate_batch_ce(model.classifier(x_0_hat), target.input_ids, target.attention_mask)
    return dict(emb=emb, x_0=x_0, x_t=x_t, x_0_hat=x_0_hat, mse=mse, ce=ce, t_T=t_T_loss, loss=loss, ce_hat=ce_hat)


def training_step_with_grads(state_dict, cfg, raw_context, target, draws, t, anchor_loss=False, noise_factor=1.0):
    """Forward + ``loss.backward()``: -> (outputs dict, {parameter name: gradient}) for the trainable parameters."""
    sd = {k: (v.detach().clone().float().requires_grad_(trainable(k)) if v.is_floating_point() else v) for k, v in state_dict.items()}
    model = _model_over(sd, cfg)  # the encoder's tensors carry requires_grad = False: frozen, as model.py:184-186 runs it
    out = training_step(model, raw_context, target, draws, t, anchor_loss, noise_factor)
    out["loss"].backward()
    grads = {k: v.grad for k, v in sd.items() if v.is_floating_point() and v.requires_grad and v.grad is not None}
    return out, grads


def _model_over(sd, cfg):
    """An OracleDiDi that reads `sd` as is (no detach), so autograd reaches the tensors."""
    m = OracleDiDi.__new__(OracleDiDi)
    m.sd, m.cfg = sd, cfg
    m.diffusion_steps = cfg.diffusion_steps
    m.s_churn, m.s_tmin, m.s_tmax = cfg.s_churn, cfg.s_tmin, cfg.s_tmax
    from .model import _Table

    m.emb = _Table(sd["emb.weight"], "emb")
    m.time_embeds = _Table(sd["time_embeds.weight"], "emb")
    m.classifier = _Table(sd["classifier.weight"], "lin")
    m.sigmas = sd["sigmas"]
    m.has_adapter = "adapter.0.weight" in sd
    return m
