"""Oracle: the reverse-diffusion sampler loops (fp32 CPU).

Restates /root/reference/src/sampling.py:6-117.  The only deliberate difference is the noise
source: the reference draws from the global torch generator (``randn_like`` at :13, ``normal_``
at :54/:74/:79); here every draw goes through ``noise_source(shape) -> tensor`` so that tests can
inject the exact stream captured from a reference run.  Draw ORDER and COUNT are the
reference's: one initial draw, then one per loop iteration (ddpm) or one per loop iteration
plus one for the final step (euler, drawn even when s_churn == 0).
``trace`` (optional list) receives the model INPUT x_t of every denoiser call, in call order.
``clamp=True`` is the per-step clamp EXTENSION of BASELINE config 4 (not in the reference, which never
clamps): every denoiser output is replaced by emb[argmax(classifier(x0))] before the sampler update.
TEST INFRASTRUCTURE: see oracle/__init__.py.
"""
import torch

from .diffusion_math import get_euler_variables, scale_input


def default_noise_source(shape):
    return torch.randn(shape)


class Context:
    """Minimal stand-in for the HF BatchEncoding the reference passes as ``raw_context``."""

    def __init__(self, input_ids, attention_mask):
        self.input_ids, self.attention_mask = input_ids, attention_mask


@torch.no_grad()
def sample(
    raw_context,
    model,
    mode,
    step_freq,
    noise_factor=1,
    tokenizer=None,
    max_len=-1,
    raw_output=False,
    skip_special=True,
    noise_source=default_noise_source,
    trace=None,
    clamp=False,
):
    emb = model.emb(raw_context.input_ids)[:, :max_len]  # sampling.py:11 (default -1 drops the last position)
    shape = tuple(emb.shape)
    x_t = noise_source(shape) * model.sigmas[-1] * noise_factor**2  # sampling.py:13
    ones = torch.ones((shape[0], 1), dtype=torch.long)
    if mode == "ddpm":
        logits = sample_ddpm(model, x_t, raw_context, ones, step_freq, noise_source, trace, clamp)
    elif mode == "euler":
        logits = sample_euler(model, x_t, raw_context, ones, step_freq, noise_source, trace, clamp)
    else:
        raise NotImplementedError(f"No {mode} sampling strategy")
    if raw_output:
        return logits
    if tokenizer is None:
        raise ValueError("`tokenizer` must be provided.")
    predictions = logits.argmax(-1)
    eos_id = tokenizer.eos_token_id or tokenizer.sep_token_id
    return tokenizer.batch_decode(truncate_predictions(predictions, eos_id), skip_special_tokens=skip_special)


def _clamp(model, x_0):
    return model.emb(model.classifier(x_0).argmax(-1))


def sample_ddpm(model, x_t, raw_context, ones, step_freq, noise_source, trace=None, clamp=False):
    # sampling.py:38-65
    T = model.diffusion_steps
    x_t = scale_input(x_t, model.sigmas[-1])
    ctx = None
    for t in range(T, 1, -step_freq):
        if trace is not None:
            trace.append(x_t.clone())
        x_0, ctx = model(
            encoder_input_ids=raw_context.input_ids,
            encoder_attention_mask=raw_context.attention_mask,
            decoder_inputs_embeds=x_t,
            time_ids=t * ones,
            context=ctx,
        )
        if clamp:
            x_0 = _clamp(model, x_0)
        sigma_t = model.sigmas[max(t - step_freq, 1)]
        noise = noise_source(tuple(x_t.shape))
        x_t = scale_input(x_0 + sigma_t * noise, sigma_t)
    if trace is not None:
        trace.append(x_t.clone())
    x_0, _ = model(
        encoder_input_ids=None,  # sampling.py:57-62 passes no ids: T<=1 raises ValueError, as the reference does
        encoder_attention_mask=raw_context.attention_mask,
        decoder_inputs_embeds=x_t,
        time_ids=ones,
        context=ctx,
    )
    if clamp:
        x_0 = _clamp(model, x_0)
    return model.classifier(x_0)


def euler_step(model, x_t, sigma_t, raw_context, ctx, next_t, num_sigmas, ones, noise, trace=None, clamp=False):
    # sampling.py:86-101
    x_t, sigma_hat = get_euler_variables(x_t, noise, sigma_t, model.s_churn, model.s_tmin, model.s_tmax, num_sigmas)
    t_hat = (model.sigmas - sigma_hat).abs().argmin()
    x_in = scale_input(x_t, sigma_hat)
    if trace is not None:
        trace.append(x_in.clone())
    x_0, ctx = model(
        encoder_input_ids=raw_context.input_ids,
        encoder_attention_mask=raw_context.attention_mask,
        decoder_inputs_embeds=x_in,
        time_ids=t_hat * ones,
        context=ctx,
    )
    if clamp:
        x_0 = _clamp(model, x_0)
    d = (x_t - x_0) / sigma_hat
    dt = model.sigmas[next_t] - sigma_hat
    return x_t + d * dt, ctx


def sample_euler(model, x_t, raw_context, ones, step_freq, noise_source, trace=None, clamp=False):
    # sampling.py:68-83
    T = model.diffusion_steps
    timesteps = range(T, 1, -step_freq)
    num_sigmas = len(timesteps)
    ctx = None
    for t in timesteps:
        noise = noise_source(tuple(x_t.shape))
        x_t, ctx = euler_step(
            model, x_t, model.sigmas[t], raw_context, ctx, max(t - step_freq, 1), num_sigmas, ones, noise, trace, clamp
        )
    noise = noise_source(tuple(x_t.shape))
    x_0, _ = euler_step(model, x_t, model.sigmas[1], raw_context, ctx, 0, num_sigmas, ones, noise, trace, clamp)
    return model.classifier(x_0)


def truncate_predictions(predictions, eos_id):
    # sampling.py:104-113: keep each row up to and including its first eos
    res = []
    for row in predictions:
        idx = (row == eos_id).nonzero()
        res.append(row[: idx[0] + 1] if len(idx) else row)
    return res
