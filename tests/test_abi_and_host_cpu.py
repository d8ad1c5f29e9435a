"""CPU-side checks: the C-ABI library loads and exports what include/didi_hip.h declares; host-side
schedule tables match the reference-generated fixtures and the oracle's per-step arithmetic."""
import ctypes
import math
import os
import re

import pytest
import torch

from golden_util import load_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "didi_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(didi_[a-z0-9_]+)\s*\(", src))


def test_library_exports_every_declared_symbol():
    from didi_amd import _lib

    declared = _declared()
    assert declared, "no prototypes parsed from the header"
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert getattr(raw, name) is not None
    assert _lib.lib.didi_version() == _lib.ABI_VERSION == 7
    assert b"workspace" in _lib.lib.didi_error_string(-4)
    assert b"fault" in _lib.lib.didi_error_string(_lib.ERR_FAULT)
    assert b"co-resident" in _lib.lib.didi_fault_string(_lib.FAULT_LN_EXCHANGE)
    assert b"fp16" in _lib.lib.didi_fault_string(_lib.FAULT_FP16_RANGE)
    assert _lib.lib.didi_fault_string(0) == b"no fault"


def test_options_are_explicit_and_validated():
    """The library reads no environment variable: tuning switches go through didi_set_option (host-only, no GPU needed)."""
    from didi_amd import _lib

    assert _lib.get_option("GEMM_LN") == 1 and _lib.get_option("BIG_MIN_M") == 2048 and _lib.get_option("GEMM_MT") == 4
    _lib.set_option("GEMM_LN", 0)
    assert _lib.get_option("GEMM_LN") == 0
    _lib.set_option("GEMM_LN", 1)
    assert _lib.lib.didi_set_option(99, 1) == -1 and _lib.lib.didi_set_option(_lib.OPTIONS["GEMM_MT"], 3) == -1
    src = "".join(open(os.path.join(ROOT, "didi_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "didi_amd", "csrc"))
                  if f.endswith((".hip", ".hpp")))
    assert "getenv" not in src


def test_size_queries_without_gpu():
    from didi_amd import _lib

    d = _lib.Decoder()
    d.H, d.heads, d.I, d.L = 768, 12, 3072, 12
    ws = _lib.lib.didi_decoder_workspace_bytes(ctypes.byref(d), 256, 64, 64)
    assert ws >= 256 * 64 * (24 * 768 + 4 * 3072)
    assert _lib.lib.didi_decoder_kv_plane_bytes(ctypes.byref(d), 256, 64) == 12 * 256 * 64 * 2 * 768 * 2
    assert _lib.lib.didi_decoder_workspace_bytes(ctypes.byref(d), 256, 97, 64) == 0  # S > 96 unsupported
    assert _lib.lib.didi_round_argmax_bytes(16384, 8008) == 16384 * 63 * 8


@pytest.mark.parametrize("sched", ["sqrt", "cosine", "linear"])
@pytest.mark.parametrize("T", [50, 2000])
def test_product_schedule_bit_exact(sched, T):
    from didi_amd.schedule import configure_schedule

    t, _ = load_case("schedules")
    sigmas, std0 = configure_schedule(T, sched)
    assert torch.equal(sigmas, t[f"{sched}.{T}.sigmas"])
    assert torch.equal(std0.reshape(1), t[f"{sched}.{T}.std_0"])
    with pytest.raises(NotImplementedError):
        configure_schedule(10, "quadratic")


def test_ddpm_plan_matches_reference_arithmetic():
    from didi_amd.schedule import configure_schedule, ddpm_plan

    sigmas, _ = configure_schedule(2000, "sqrt")
    for f in (1, 10, 7):
        plan = ddpm_plan(sigmas, 2000, f)
        ts = list(range(2000, 1, -f))
        assert plan.time_row == ts + [1] and plan.n_calls == len(ts) + 1
        for k, t in enumerate(ts[:5] + ts[-5:]):
            kk = k if k < 5 else len(ts) - 10 + k
            s = sigmas[max(t - f, 1)].double().item()
            assert math.isclose(plan.a[kk], 1 / math.sqrt(s * s + 1), rel_tol=1e-12)
            assert math.isclose(plan.b[kk], s / math.sqrt(s * s + 1), rel_tol=1e-12)
    sT = sigmas[-1].double().item()
    assert math.isclose(ddpm_plan(sigmas, 2000, 1, 2.0).init_scale, sT * 4 / math.sqrt(sT * sT + 1), rel_tol=1e-12)


def test_euler_plan_matches_oracle_steps():
    from didi_amd.schedule import configure_schedule, euler_plan
    from oracle.diffusion_math import get_euler_variables

    sigmas, _ = configure_schedule(20, "sqrt")
    kw = dict(s_churn=1.5, s_tmin=0.5, s_tmax=50.0)
    plan = euler_plan(sigmas, 20, 4, **kw)
    ts = list(range(20, 1, -4))
    steps = [(t, max(t - 4, 1)) for t in ts] + [(1, 0)]
    assert plan.n_calls == len(steps)
    x = torch.zeros(1)
    for k, (t, nt) in enumerate(steps):
        _, s_hat = get_euler_variables(x, x, sigmas[t], kw["s_churn"], kw["s_tmin"], kw["s_tmax"], len(ts))
        assert plan.time_row[k] == int((sigmas - s_hat).abs().argmin())
        s, sh = sigmas[t].item(), float(s_hat)
        assert math.isclose(plan.c_in[k], 1 / math.sqrt(sh * sh + 1), rel_tol=1e-6)
        assert math.isclose(plan.ratio[k], (sigmas[nt].item() - sh) / sh, rel_tol=1e-6, abs_tol=1e-9)
        assert math.isclose(plan.churn[k], math.sqrt(max(sh * sh - s * s, 0.0)), rel_tol=1e-4, abs_tol=1e-7)
    assert plan.ratio[-1] == -1.0
    assert any(c > 0 for c in plan.churn) and any(c == 0 for c in plan.churn)


def test_target_len_follows_reference_slicing():
    from didi_amd.sampling import _target_len

    assert _target_len(64, -1) == 63 and _target_len(64, 64) == 64 and _target_len(24, 100) == 24 and _target_len(12, 8) == 8
