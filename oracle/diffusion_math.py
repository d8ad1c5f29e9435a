"""Oracle: noise schedules and the sampler's scalar helpers (fp32 torch-CPU semantics).

Restates /root/reference/src/diffusion/schedules.py:4-21 and src/diffusion/utils.py:6-57.
The arithmetic is kept in fp32 torch ops in the same order as the reference so that
``sigmas`` is bit-identical to ``configure_schedule`` (checked against tests/golden).
TEST INFRASTRUCTURE: see oracle/__init__.py.
"""
import math

import torch


def linear_beta_schedule(steps, low=1e-4, high=0.02):
    # reference: src/diffusion/schedules.py:4-5
    return torch.linspace(low, high, steps)


def cosine_beta_schedule(steps, s=0.008):
    # reference: src/diffusion/schedules.py:8-13
    x = torch.arange(steps + 1)
    ac = torch.cos(((x / steps) + s) / (1 + s) * torch.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.clip(betas, 1e-4, 1 - 1e-4)


def sqrt_beta_schedule(steps):
    # reference: src/diffusion/schedules.py:16-21
    x = torch.linspace(0, 1, steps + 1)
    ac = 1 - torch.sqrt(x + 0.0001)
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.clip(betas, 1e-4, 1 - 1e-4)


def configure_schedule(steps, schedule):
    """-> (sigmas[T+1] fp32 with sigmas[0]=0, std_0).  reference: src/diffusion/utils.py:6-21"""
    if schedule == "linear":
        betas = linear_beta_schedule(steps)
    elif schedule == "cosine":
        betas = cosine_beta_schedule(steps)
    elif schedule == "sqrt":
        betas = sqrt_beta_schedule(steps)
    else:
        raise NotImplementedError(f"TODO: implement {schedule} schedule")
    alphas = 1 - betas
    ac = torch.cumprod(alphas, dim=0)
    sigmas = ((1 - ac) / ac) ** 0.5
    sigmas = torch.cat([sigmas.new_zeros([1]), sigmas])
    return sigmas, torch.sqrt(betas[0])


def scale_input(x, sigma_t):
    # reference: src/diffusion/utils.py:30-31
    return x / (sigma_t**2 + 1) ** 0.5


def get_euler_variables(x_t, noise, sigma_t, s_churn, s_tmin, s_tmax, num_sigmas):
    # reference: src/diffusion/utils.py:43-57
    gamma = min(s_churn / num_sigmas, math.sqrt(2.0) - 1) if s_tmin <= sigma_t <= s_tmax else 0.0
    sigma_hat = sigma_t * (gamma + 1)
    if gamma > 0:
        x_t = x_t + (sigma_hat**2 - sigma_t**2) ** 0.5 * noise
    return x_t, sigma_hat


def get_x0(emb, std, noise):
    # reference: src/diffusion/utils.py:24-27 (its randn_like is the caller-supplied `noise` here)
    return emb + std * noise


def get_diffusion_variables(x_0, sigmas, noise, t, noise_factor=1):
    """reference: src/diffusion/utils.py:34-40.  The reference draws ``t = randint(1, T + 1, [B, 1])`` itself; here the
    caller passes it (tests inject the captured draw), so the function is deterministic."""
    sigma_t = sigmas[t].view(-1, 1, 1)
    x_t = scale_input(x_0 + sigma_t * noise * noise_factor**2, sigma_t)
    return x_t, t
