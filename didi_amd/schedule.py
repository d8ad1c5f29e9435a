"""Host-side schedule tables for the sampler.

Everything here is data-independent, so it is computed ONCE on the host and handed to the
device loop as small arrays indexed by the step counter (no per-step D2H sync; the reference
syncs on 0-dim device tensors inside ``get_euler_variables``, src/diffusion/utils.py:52-55).

* ``configure_schedule`` reproduces reference ``configure_schedule`` (src/diffusion/utils.py:6-21
  over src/diffusion/schedules.py:4-21) bit-for-bit: same fp32 torch-CPU ops in the same order
  (fp32 ``cumprod`` order matters), so ``sigmas`` can be compared with ``torch.equal``.
* ``ddpm_plan`` / ``euler_plan`` turn ``sigmas`` + (T, step_freq, churn knobs) into per-call
  coefficient rows for the fused step kernels (didi_amd/csrc/elementwise.hip).  Coefficients are
  formed in fp64 and rounded once to fp32: sigma_T is ~7e3, so ``sigma*eps`` is never formed in
  low precision.
"""
import math
from dataclasses import dataclass

import torch


def _betas_from_alpha_bar(alpha_bar: torch.Tensor) -> torch.Tensor:
    alpha_bar = alpha_bar / alpha_bar[0]
    return torch.clip(1 - alpha_bar[1:] / alpha_bar[:-1], 1e-4, 1 - 1e-4)


def _betas(steps: int, schedule: str) -> torch.Tensor:
    if schedule == "sqrt":  # schedules.py:16-21
        return _betas_from_alpha_bar(1 - torch.sqrt(torch.linspace(0, 1, steps + 1) + 0.0001))
    if schedule == "cosine":  # schedules.py:8-13, s = 0.008
        x = torch.arange(steps + 1)
        return _betas_from_alpha_bar(torch.cos(((x / steps) + 0.008) / (1 + 0.008) * torch.pi * 0.5) ** 2)
    if schedule == "linear":  # schedules.py:4-5
        return torch.linspace(1e-4, 0.02, steps)
    raise NotImplementedError(f"TODO: implement {schedule} schedule")


def configure_schedule(steps: int, schedule: str):
    """-> (sigmas [steps+1] fp32 with sigmas[0] = 0, std_0 0-dim fp32)."""
    betas = _betas(steps, schedule)
    alpha_bar = torch.cumprod(1 - betas, dim=0)
    sigmas = ((1 - alpha_bar) / alpha_bar) ** 0.5
    return torch.cat([sigmas.new_zeros([1]), sigmas]), torch.sqrt(betas[0])


@dataclass
class DDPMPlan:
    """One row per denoiser call k = 0..n_calls-1 (src/sampling.py:38-65).

    time_row[k]   row of ``time_embeds`` added before call k          (t, then 1 for the last call)
    a[k], b[k]    after call k<n_calls-1:  x_t <- a*x0_hat + b*eps    (a = 1/sqrt(s^2+1), b = s*a,
                  s = sigmas[max(t-f,1)]); unused for the last call
    init_scale    x_T*scale_input folded: x_in(0) = init_scale * eps0 (sampling.py:13,42)
    """

    time_row: list
    a: list
    b: list
    init_scale: float

    @property
    def n_calls(self):
        return len(self.time_row)


def ddpm_plan(sigmas: torch.Tensor, diffusion_steps: int, step_freq: int, noise_factor: float = 1.0) -> DDPMPlan:
    sig = sigmas.double().tolist()
    rows, a, b = [], [], []
    for t in range(diffusion_steps, 1, -step_freq):
        s = sig[max(t - step_freq, 1)]
        inv = 1.0 / math.sqrt(s * s + 1.0)
        rows.append(t)
        a.append(inv)
        b.append(s * inv)
    rows.append(1)
    a.append(0.0)
    b.append(0.0)
    sT = sig[-1]
    return DDPMPlan(rows, a, b, sT * float(noise_factor) ** 2 / math.sqrt(sT * sT + 1.0))


@dataclass
class EulerPlan:
    """One row per euler_step call (src/sampling.py:68-101, src/diffusion/utils.py:43-57).

    For call k with current VE-coordinate state x:
      x'      = x + churn[k]*eps_k            (churn = sqrt(sigma_hat^2 - sigma^2), 0 when gamma == 0)
      x_in    = c_in[k] * x'                  (scale_input(x', sigma_hat))
      x0_hat  = denoiser(x_in, time row t_hat[k])
      x       = x' + (x' - x0_hat) * ratio[k] (ratio = (sigma_next - sigma_hat)/sigma_hat)
    init_scale: x = init_scale * eps0  (sigma_T * noise_factor^2, sampling.py:13; no scale_input)
    """

    time_row: list
    churn: list
    c_in: list
    ratio: list
    init_scale: float

    @property
    def n_calls(self):
        return len(self.time_row)


def euler_plan(sigmas: torch.Tensor, diffusion_steps: int, step_freq: int, noise_factor: float = 1.0,
               s_churn: float = 0.0, s_tmin: float = 0.0, s_tmax: float = float("inf")) -> EulerPlan:
    timesteps = list(range(diffusion_steps, 1, -step_freq))
    num_sigmas = len(timesteps)
    sig32 = sigmas.float().cpu()
    rows, churn, c_in, ratio = [], [], [], []

    def one(t_cur, t_next):
        s32 = sig32[t_cur]
        gamma = min(s_churn / num_sigmas, 2**0.5 - 1) if s_tmin <= float(s32) <= s_tmax else 0.0
        # t_hat is chosen by the reference from the fp32 product sigma*(gamma+1) (utils.py:53,
        # sampling.py:88): keep that decision in fp32 so the same time-embedding row is picked.
        s_hat32 = s32 * (gamma + 1)
        rows.append(int((sig32 - s_hat32).abs().argmin()))
        s, s_hat, s_next = float(s32), float(s_hat32), float(sig32[t_next])
        churn.append(math.sqrt(max(s_hat * s_hat - s * s, 0.0)) if gamma > 0 else 0.0)
        c_in.append(1.0 / math.sqrt(s_hat * s_hat + 1.0))
        ratio.append((s_next - s_hat) / s_hat)

    for t in timesteps:
        one(t, max(t - step_freq, 1))
    one(1, 0)
    return EulerPlan(rows, churn, c_in, ratio, float(sig32[-1]) * float(noise_factor) ** 2)
