"""didi_amd: MI355X-native drop-in for DiDi's reverse-diffusion sampler path.

Public surface mirrors the reference:
  didi_amd.sampling.sample(raw_context, model, mode, step_freq, ...)   <- src/sampling.py:7
  didi_amd.model.DiDiSampler                                           <- src/diffusion/model.py:66 (sampler-visible part)
All device arithmetic runs in hand-written HIP kernels behind the C ABI in include/didi_hip.h
(didi_amd/csrc -> didi_amd/libdidi_hip.so); there is no CPU or eager-torch fallback.
"""
from .config import DiDiConfig, blenderbot_400m_distill, tiny_config1  # noqa: F401

__version__ = "0.1.0"
