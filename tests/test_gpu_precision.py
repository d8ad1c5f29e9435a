"""Precision hardening of the fp16 (default) mode, through the C ABI.

fp16 planes carry 11 mantissa bits but only 5 exponent bits: |x| > 65504 does not fit.  Defined behaviour (include/didi_hip.h,
DIDI_FAULT_FP16_RANGE): GEMM epilogues and fp32 -> plane splits SATURATE finite values to +-65504 (the hardware's FP16_OVFL
conversion mode: no inf is ever created; an inf or NaN that was already in the data stays) and the GEMM raises a bit in the caller's device fault word, so that `sample()` / `model.check_fault()` fail loudly; bf16x3 (bf16 planes,
fp32's exponent range) is the mode to rerun in.  The reference computes in fp32 and has no counterpart for any of this.

The second half is a trained-weights-like stress model: random-init N(0, 0.02) weights (every other fixture) have no outlier
features, trained BERT-style decoders do.  Expected values come from the CPU oracle (parity unpinned by the reference:
it ships no trained checkpoint and no golden outputs).
"""
import pytest
import torch

from golden_util import trained_like as _trained_like  # noqa: F401  (kept under its old name for the probes)
from golden_util import ListNoise, config1, config1_state_dict, config_to_oracle_config

pytestmark = pytest.mark.gpu
F16_MAX = 65504.0


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,N,K,small", [(2304, 768, 768, False), (300, 384, 256, False), (504, 768, 768, True)])
def test_fp16_gemm_saturates_and_reports(M, N, K, small):
    """Operands of magnitude ~60 with K >= 256 give |A W^T| up to ~1e5: past the fp16 range on the 256x192 tile, the 128x128 tile
    and the split-K small-batch form.  Planes must come back finite and saturated, the fp32 output exact, the fault bit set;
    the same operands at 1/100 of the scale must leave the fault word clear."""
    from didi_amd import _lib, ops

    g = torch.Generator().manual_seed(3)
    a = (torch.randn(M, K, generator=g) * 60).cuda()
    w = (torch.randn(N, K, generator=g) * 60).cuda()
    bias = torch.randn(N, generator=g).cuda()
    a_hi, a_lo = ops.split(a, dtype=torch.float16)
    w_hi, w_lo = ops.split(w, dtype=torch.float16)
    want = a_hi.double() @ w_hi.double().T + bias.double()
    assert (want.abs() > F16_MAX).float().mean().item() > 0.01, "the test must actually leave the fp16 range"
    fault = ops.new_fault("cuda")
    kw = dict(small=()) if small else {}
    c_hi, c_lo, c = ops.gemm(a_hi, a_lo, w_hi, w_lo, bias=bias, precision="fp16", out_f32=True, fault=fault, **kw)
    assert rel_l2(c, want) <= 2e-5, "the fp32 output is not touched by the saturation"
    assert bool(torch.isfinite(c_hi.float()).all()) and bool(torch.isfinite(c_lo.float()).all())
    sat = want.clamp(-F16_MAX, F16_MAX)
    got = c_hi.double() + c_lo.double()
    over = want.abs() > F16_MAX
    assert torch.equal(c_hi.double()[over], sat[over]), "the hi plane (the GEMM operand) holds out-of-range values as exactly +-65504"
    # the remainder plane saturates on its own: hi + lo (what a residual add re-joins) reaches 2 x 65504 and is never further from
    # the true value than the saturated hi plane alone
    assert bool(((got - want).abs()[over] <= (sat - want).abs()[over]).all()) and float(got.abs().max()) <= 2 * F16_MAX
    assert rel_l2(got[~over], sat[~over]) <= 1e-3
    with pytest.raises(_lib.DidiHipError, match="fp16 range"):
        _lib.check_fault(fault, "gemm")
    assert fault.item() & _lib.FAULT_FP16_RANGE
    # in range: no fault
    fault.zero_()
    a2_hi, a2_lo = ops.split(a * 0.01, dtype=torch.float16)
    ops.gemm(a2_hi, a2_lo, w_hi, w_lo, bias=bias, precision="fp16", fault=fault, **kw)
    _lib.check_fault(fault, "gemm in range")
    # bf16 planes have fp32's range: nothing to report, values intact
    b_hi, b_lo = ops.split(a, dtype=torch.bfloat16)
    v_hi, v_lo = ops.split(w, dtype=torch.bfloat16)
    y_hi, y_lo, _ = ops.gemm(b_hi, b_lo, v_hi, v_lo, bias=bias, precision="bf16x3", fault=fault, **kw)
    _lib.check_fault(fault, "bf16x3")
    assert rel_l2(y_hi.double() + y_lo.double(), a.double() @ w.double().T + bias.double()) <= 3e-5


def test_fp16_split_saturates_and_keeps_nan():
    from didi_amd import ops

    x = torch.tensor([1e5, -3e38, 65504.0, 65520.0, -65519.9, float("inf"), float("-inf"), float("nan"), 1.5, -0.0, 7e4, 1e-8]).cuda()
    hi, lo = ops.split(x, dtype=torch.float16)
    h = hi.float().cpu()
    # finite overflow saturates (the hardware's FP16_OVFL mode); a true infinity or NaN in the INPUT stays what it is
    assert h[:5].tolist() == [F16_MAX, -F16_MAX, F16_MAX, F16_MAX, -F16_MAX]
    assert h[5].item() == float("inf") and h[6].item() == float("-inf")
    assert torch.isnan(h[7]) and h[8].item() == 1.5 and h[10].item() == F16_MAX
    assert bool(torch.isfinite(lo.float().cpu()[[0, 1, 2, 3, 4, 8, 9, 10, 11]]).all())
    # bf16 planes are not touched: the value's own exponent range
    hb, _ = ops.split(x, dtype=torch.bfloat16)
    assert hb.float().cpu()[0].item() == pytest.approx(1e5, rel=4e-3) and torch.isinf(hb.float().cpu()[5])


def test_fp16_ffn_gelu_past_range_in_decoder_reports_fault():
    """End to end: a decoder whose FFN-up bias pushes GELU outputs past 65504 must not return plausible ids silently:
    fp16 raises the device fault (and sample() refuses to decode); bf16x3 runs the same weights cleanly."""
    from didi_amd import _lib
    from didi_amd.model import DiDiSampler
    from didi_amd.sampling import sample
    from didi_amd.synthetic import RawContext, synthetic_batch

    cfg, sd = config1(), dict(config1_state_dict())
    sd["decoder.encoder.layer.0.intermediate.dense.bias"] = sd["decoder.encoder.layer.0.intermediate.dense.bias"].clone()
    sd["decoder.encoder.layer.0.intermediate.dense.bias"][7] = 1.0e5
    ids, mask = synthetic_batch(4, 24, cfg.vocab, seed=3)
    ctx = RawContext(ids.cuda(), mask.cuda())

    class Tok:
        eos_token_id, sep_token_id = 2, None

        def batch_decode(self, rows, skip_special_tokens=True):
            return [" ".join(str(int(x)) for x in r) for r in rows]

    torch.manual_seed(1)
    model = DiDiSampler(sd, cfg, device="cuda", precision="fp16")
    with pytest.raises(_lib.DidiHipError, match="fp16 range"):
        sample(ctx, model, "ddpm", 10, tokenizer=Tok())
    model.check_fault("cleared by the failed call")  # the word was reset: the next call starts clean
    m3 = DiDiSampler(sd, cfg, device="cuda", precision="bf16x3")
    out = sample(ctx, m3, "ddpm", 10, tokenizer=Tok())
    assert len(out) == 4


def test_trained_like_stress_model_vs_oracle():
    """Config-1-sized model with trained-like statistics, 32 dialogs x 23 positions, T = 50 ddpm, injected noise.
    bf16x3 keeps the fp32-class bar (x_t 1e-4, ids >= 99.9 %).  fp16 must stay finite, raise no fault and hold
    x_t <= 2e-3; its id agreement is reported and bounded at >= 99 % -- outlier features cost one-product precision
    more than random-init weights do, which is why the fp32-class mode exists."""
    from didi_amd.model import DiDiSampler
    from didi_amd.sampling import sample
    from didi_amd.synthetic import RawContext, synthetic_batch
    from oracle.model import OracleDiDi
    from oracle.sampling import Context, sample as oracle_sample

    cfg = config1()
    sd = _trained_like(config1_state_dict())
    B, C = 32, 24
    ids, mask = synthetic_batch(B, C, cfg.vocab, seed=9)
    torch.manual_seed(13)
    draws = [torch.randn(B, C - 1, cfg.dec_dim) for _ in range(cfg.diffusion_steps)]
    trace = []
    want = oracle_sample(Context(ids, mask), OracleDiDi(sd, config_to_oracle_config(cfg)), "ddpm", 1, raw_output=True,
                         noise_source=ListNoise(draws), trace=trace)
    want_ids = want.argmax(-1)
    ctx = RawContext(ids.cuda(), mask.cuda())
    res = {}
    for precision in ("bf16x3", "fp16x2", "fp16"):
        model = DiDiSampler(sd, cfg, device="cuda", precision=precision)
        tr = []
        got, _ = sample(ctx, model, "ddpm", 1, noise=torch.stack(draws), return_ids=True, trace=tr)
        model.check_fault(precision)
        assert all(bool(torch.isfinite(t).all()) for t in tr)
        res[precision] = ((got.cpu() == want_ids).float().mean().item(), max(rel_l2(a, b) for a, b in zip(tr, trace)))
        print(f"\n[trained-like stress {precision}] id match {res[precision][0]:.4f} on {want_ids.numel()}  max x_t rel-L2 {res[precision][1]:.3e}")
    assert res["bf16x3"][1] <= 1e-4 and res["bf16x3"][0] >= 0.999
    assert res["fp16x2"][1] <= 1e-3 and res["fp16x2"][0] >= 0.999
    assert res["fp16"][1] <= 2e-3 and res["fp16"][0] >= 0.99
