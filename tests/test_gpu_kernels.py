"""GPU parity tests of every C-ABI kernel entry point against fp64/fp32 torch-CPU references.

Tolerances (stated per test):
  * bf16 mode   : compared with the SAME bf16-rounded operands multiplied in fp64 -> only the
                  fp32 accumulation order differs: rel-L2 <= 2e-5.
  * fp16 mode   : the same with fp16-rounded operands (one product on fp16 planes).
  * bf16x3 mode : compared with the un-rounded fp32 operands in fp64: rel-L2 <= 3e-5 (the split
                  representation carries ~16 mantissa bits; the lo*lo term is dropped).
  * integer outputs (argmax ids, eos lengths, raw Philox stream): exact.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from didi_amd import ops as o

    return o


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


PD = {"bf16": torch.bfloat16, "fp16": torch.float16, "bf16x3": torch.bfloat16, "fp16x2": torch.float16, "fp16x3": torch.float16}  # plane dtype per precision
ALL_PREC = ["bf16", "fp16", "bf16x3", "fp16x2"]


def ref_operands(precision, a, w, a_hi, w_hi, w_lo):
    """The operands a precision multiplies, as fp32 CPU tensors: one product = the hi planes; fp16x2 = A_hi x (W_hi + W_lo);
    bf16x3 = the un-rounded values (its split carries ~16 bits, the lo*lo term is dropped: tolerance 3e-5)."""
    if precision in ("bf16x3", "fp16x3"):
        return a, w
    if precision == "fp16x2":
        return a_hi.cpu().float(), (w_hi.float() + w_lo.float()).cpu()
    return a_hi.cpu().float(), w_hi.cpu().float()


def is_nearest_plane(hi, y, pd):
    """The hi plane is A nearest 16-bit value of y: fp16 epilogues of the 256x192 kernel round ties away from zero (packed
    truncating conversion with a half-ulp offset, common.hpp f16x2_dither), bf16 ones and every other kernel to even.  Below
    fp16's normal range (|y| < 2^-14) the offset is less than half a subnormal step: there the plane may be one step (2^-24) off."""
    d_hi, d_rne = (hi.float() - y).abs().cpu(), (y.to(pd).float() - y).abs().cpu()
    return bool((d_hi <= d_rne + (2.0 ** -24 if pd == torch.float16 else 0.0)).all())


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ------------------------------------------------------------------ split
def test_split_planes(ops):
    x = rnd(1000, 128, seed=1, scale=3.0)
    x[0, :4] = torch.tensor([0.0, -0.0, 1e-30, 65504.0])
    hi, lo = ops.split(x.cuda())
    assert torch.equal(hi.cpu(), x.to(torch.bfloat16)), "hi plane must be RNE bf16 of the value"
    back = ops.join(hi, lo).cpu()
    err = (back - x).abs() / x.abs().clamp_min(1e-20)
    assert err.max().item() <= 2.0**-15, err.max().item()


def test_split_planes_fp16(ops):
    x = rnd(1000, 128, seed=1, scale=3.0)
    x[0, :4] = torch.tensor([0.0, -0.0, 1e-30, 65504.0])
    hi, lo = ops.split(x.cuda(), dtype=torch.float16)
    assert hi.dtype == torch.float16 and torch.equal(hi.cpu(), x.to(torch.float16)), "hi plane must be RNE fp16 of the value"
    back = ops.join(hi, lo).cpu()
    # |x| < ~0.1 puts the remainder into fp16 subnormals (quantum 2^-24): absolute error <= 2^-25 there,
    # relative 2^-22 above
    assert ((back - x).abs() <= torch.maximum(x.abs() * 2.0**-21, torch.tensor(2.0**-24))).all()


# ------------------------------------------------------------------ philox
def test_philox_matches_oracle(ops):
    from oracle.philox import normal

    n, seed, off = 1 << 16, 0xC0FFEE1234, 977
    z = ops.philox_normal((n,), seed, off).cpu().numpy()
    want = normal(n, seed, off)
    # same raw 32-bit stream; device logf/sincosf differ from numpy by a few ulp
    assert np.abs(z - want).max() <= 5e-5, np.abs(z - want).max()
    z2 = ops.philox_normal((n // 2,), seed, off + n // 8).cpu().numpy()
    assert np.array_equal(z2, z[n // 2:]), "counter-based: a later offset must reproduce the tail"
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02


# ------------------------------------------------------------------ layernorm / gather
@pytest.mark.parametrize("M,H", [(7, 128), (184, 128), (33, 768), (5, 1280), (9, 96), (3, 2048)])
def test_layernorm(ops, M, H):
    x = rnd(M, H, seed=M + H, scale=2.0) + 0.5
    g, b = 1 + rnd(H, seed=1, scale=0.1), rnd(H, seed=2, scale=0.1)
    want = torch.nn.functional.layer_norm(x.double(), (H,), g.double(), b.double(), 1e-12)
    hi, lo, y = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-12, f32=True)
    assert rel_l2(y, want) <= 1e-6
    assert rel_l2(ops.join(hi, lo), want) <= 2e-5
    assert torch.equal(hi.cpu(), y.cpu().to(torch.bfloat16))
    hi, lo, y = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-12, f32=True, dtype=torch.float16)
    assert torch.equal(hi.cpu(), y.cpu().to(torch.float16)) and rel_l2(ops.join(hi, lo), want) <= 2e-6


@pytest.mark.parametrize("M,H", [(5, 96), (130, 128), (33, 768), (9, 2048)])
def test_rmsnorm(ops, M, H):
    """T5LayerNorm: x * rsqrt(mean(x^2) + eps) * gamma (no mean subtraction, no bias)."""
    x = rnd(M, H, seed=M + H, scale=2.0) + 0.5
    g = 1 + rnd(H, seed=1, scale=0.1)
    want = x.double() * torch.rsqrt(x.double().pow(2).mean(-1, keepdim=True) + 1e-6) * g.double()
    hi, lo, y = ops.rmsnorm(x.cuda(), g.cuda(), 1e-6, f32=True)
    assert rel_l2(y, want) <= 1e-6
    assert rel_l2(ops.join(hi, lo), want) <= 2e-5 and torch.equal(hi.cpu(), y.cpu().to(torch.bfloat16))
    hi, lo, y = ops.rmsnorm(x.cuda(), g.cuda(), 1e-6, f32=True, dtype=torch.float16)
    assert torch.equal(hi.cpu(), y.cpu().to(torch.float16)) and rel_l2(ops.join(hi, lo), want) <= 2e-6


def test_mul_split(ops):
    a, b = rnd(37, 192, seed=8), rnd(37, 192, seed=9)
    hi, lo = ops.mul_split(a.cuda(), b.cuda())
    assert rel_l2(ops.join(hi, lo), a.double() * b.double()) <= 2e-5
    assert torch.equal(hi.cpu(), (a * b).to(torch.bfloat16))
    hi, lo = ops.mul_split(a.cuda(), b.cuda(), need_lo=False, dtype=torch.float16)
    assert lo is None and torch.equal(hi.cpu(), (a * b).to(torch.float16))


def test_embed_gather(ops):
    table, pos = rnd(50, 96, seed=3), rnd(12, 96, seed=4)
    ids = torch.randint(0, 50, (3, 12), generator=torch.Generator().manual_seed(5))
    out = ops.embed_gather(ids.cuda(), table.cuda(), 9.5, pos.cuda()).cpu()
    assert torch.allclose(out, table[ids] * 9.5 + pos[None], rtol=1e-6, atol=1e-6)
    out = ops.embed_gather(ids.cuda(), table.cuda()).cpu()
    assert torch.equal(out, table[ids])


# ------------------------------------------------------------------ fused step + embedding + LN
@pytest.mark.parametrize("H", [128, 768])
def test_step_embed_ln_modes(ops, H):
    from didi_amd import _lib

    B, S, T = 3, 11, 20
    M = B * S
    X, X0, E = rnd(M, H, seed=1, scale=50.0), rnd(M, H, seed=2), rnd(M, H, seed=3)
    temb, pos, tok = rnd(T + 1, H, seed=4), rnd(S + 3, H, seed=5, scale=0.1), rnd(H, seed=6, scale=0.1)
    g, b = 1 + rnd(H, seed=7, scale=0.1), rnd(H, seed=8, scale=0.1)
    tidx = torch.tensor([3, 20, 1], dtype=torch.int32)
    dev = lambda t: t.cuda()  # noqa: E731
    cases = [
        (_lib.STEP_INIT, dict(ce=7034.0 / 7034.1), lambda: (7034.0 / 7034.1) * E),
        (_lib.STEP_DDPM, dict(cx0=0.6, ce=0.8), lambda: 0.6 * X0 + 0.8 * E),
        (_lib.STEP_EULER, dict(ratio=-0.25, ce=0.3, cin=0.02), lambda: (X + (X - X0) * -0.25) + 0.3 * E),
        (_lib.STEP_NONE, dict(), lambda: X),
    ]
    for mode, kw, ref in cases:
        for per_row_t in (False, True):
            x_ref = ref().double()
            cin = kw.get("cin", 1.0)
            t_rows = tidx.long().repeat_interleave(S) if per_row_t else torch.full((M,), 5)
            pre = cin * x_ref + temb.double()[t_rows] + tok.double() + pos.double()[:S].repeat(B, 1)
            want = torch.nn.functional.layer_norm(pre, (H,), g.double(), b.double(), 1e-12)
            x_out = torch.empty(M, H, device="cuda")
            xin = torch.empty(M, H, device="cuda")
            h_hi = torch.empty(M, H, dtype=torch.float16 if per_row_t else torch.bfloat16, device="cuda")
            h_lo = torch.empty_like(h_hi)
            ops.step_embed_ln(mode, M, S, H, X=dev(X), X0=dev(X0), E=dev(E), x_out=x_out, xin_out=xin,
                              time_embeds=dev(temb), t=5, time_idx=dev(tidx) if per_row_t else None, pos_emb=dev(pos),
                              tok_type0=dev(tok), ln_gamma=dev(g), ln_beta=dev(b), ln_eps=1e-12, h_hi=h_hi, h_lo=h_lo,
                              cx0=kw.get("cx0", 0.0), ce=kw.get("ce", 0.0), ratio=kw.get("ratio", 0.0), cin=cin)
            assert rel_l2(x_out, x_ref) <= 1e-6, (mode, per_row_t)
            assert rel_l2(xin, cin * x_ref) <= 1e-6
            assert rel_l2(ops.join(h_hi, h_lo), want) <= 3e-5, (mode, per_row_t)


def test_step_in_kernel_philox_matches_standalone(ops):
    from didi_amd import _lib

    M, S, H = 8, 4, 128
    seed, off = 99, 12345
    x_out = torch.empty(M, H, device="cuda")
    ops.step_embed_ln(_lib.STEP_INIT, M, S, H, ce=2.0, seed=seed, offset=off, x_out=x_out)
    z = ops.philox_normal((M * H,), seed, off).view(M, H)
    assert torch.equal(x_out, 2.0 * z)


# ------------------------------------------------------------------ GEMM
def _gemm_ref(a, w, bias, act, resid):
    v = a.double() @ w.double().t()
    if bias is not None:
        v = v + bias.double()
    if act == "gelu":
        v = torch.nn.functional.gelu(v)
    elif act == "tanh":
        v = torch.tanh(v)
    elif act == "relu":
        v = torch.relu(v)
    elif act == "gelu_new":
        v = torch.nn.functional.gelu(v, approximate="tanh")
    if resid is not None:
        v = v + resid.double()
    return v


@pytest.mark.parametrize("M,N,K", [(184, 384, 128), (1, 128, 128), (130, 600, 128), (257, 8008, 128), (64, 288, 96),
                                   (300, 768, 768), (200, 768, 3072), (129, 2304, 768), (50, 768, 1280)])
@pytest.mark.parametrize("precision", ALL_PREC)
def test_gemm_plain(ops, M, N, K, precision):
    a, w, bias = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3)
    a_hi, a_lo = ops.split(a.cuda(), dtype=PD[precision])
    w_hi, w_lo = ops.split(w.cuda(), dtype=PD[precision])
    c_hi, c_lo, c = ops.gemm(a_hi, a_lo, w_hi, w_lo, bias=bias.cuda(), precision=precision, out_f32=True)
    want = _gemm_ref(*ref_operands(precision, a, w, a_hi, w_hi, w_lo), bias, None, None)
    tol = 3e-5 if precision == "bf16x3" else 2e-5
    assert rel_l2(c, want) <= tol, rel_l2(c, want)
    assert (c.cpu().double() - want).abs().max().item() <= 50 * tol * want.abs().max().item()
    assert rel_l2(ops.join(c_hi, c_lo), c) <= 2e-5
    assert torch.equal(c_hi.cpu(), c.cpu().to(PD[precision]))


@pytest.mark.parametrize("act", [None, "gelu", "tanh", "relu", "gelu_new"])
@pytest.mark.parametrize("resid", [None, "f32", "planes"])
def test_gemm_epilogues(ops, act, resid):
    M, N, K = 150, 256, 192
    a, w, bias, r = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=0.1), rnd(N, seed=6), rnd(M, N, seed=7)
    a_hi, a_lo = ops.split(a.cuda())
    w_hi, w_lo = ops.split(w.cuda())
    kw = {}
    r_used = None
    if resid == "f32":
        kw["residual"], r_used = r.cuda(), r
    elif resid == "planes":
        kw["residual_planes"] = ops.split(r.cuda())
        r_used = ops.join(*kw["residual_planes"]).cpu()
    _, _, c = ops.gemm(a_hi, a_lo, w_hi, w_lo, bias=bias.cuda(), act=act, precision="bf16x3", out_f32=True, out_planes=False, **kw)
    want = _gemm_ref(a, w, bias, act, r_used)
    assert rel_l2(c, want) <= 3e-5, rel_l2(c, want)
    # no bias
    _, _, c = ops.gemm(a_hi, a_lo, w_hi, w_lo, precision="bf16x3", out_f32=True, out_planes=False)
    assert rel_l2(c, _gemm_ref(a, w, None, None, None)) <= 3e-5


@pytest.mark.parametrize("M,N,K", [(63, 768, 768), (63, 2304, 768), (1, 3072, 768), (504, 768, 3072), (130, 128, 256), (2016, 768, 768)])
@pytest.mark.parametrize("precision", ALL_PREC)
@pytest.mark.parametrize("ln", [False, True])
def test_gemm_small_batch_form(ops, M, N, K, precision, ln):
    """didi_gemm_small (split-K + row epilogue) == [LayerNorm](gelu(A W^T + bias) + residual) on the same rounded operands, and
    agrees with the plain launch of didi_gemm to summation-order rounding."""
    from didi_amd._lib import lib

    a, w, bias, r = rnd(M, K, seed=41), rnd(N, K, seed=42, scale=0.05), rnd(N, seed=43), rnd(M, N, seed=44)
    g, b = 1 + rnd(N, seed=45, scale=0.1), rnd(N, seed=46, scale=0.1)
    pd, x3 = PD[precision], precision == "bf16x3"
    assert lib.didi_gemm_small_workspace_bytes(M, N, K, ops.PRECISIONS[precision]) > 0
    a_hi, a_lo = ops.split(a.cuda(), dtype=pd)
    w_hi, w_lo = ops.split(w.cuda(), dtype=pd)
    rp = ops.split(r.cuda(), dtype=pd)
    act = None if ln else "gelu"
    small = (g.cuda(), b.cuda(), 1e-12) if ln else ()
    c_hi, c_lo, c = ops.gemm(a_hi, a_lo, w_hi, w_lo, bias=bias.cuda(), act=act, precision=precision, residual_planes=rp,
                             out_f32=True, small=small)
    ar, wr = (ops.join(a_hi, a_lo).cpu(), ops.join(w_hi, w_lo).cpu()) if x3 else ref_operands(precision, a, w, a_hi, w_hi, w_lo)
    want = _gemm_ref(ar, wr, bias, act, ops.join(*rp).cpu())
    if ln:
        want = torch.nn.functional.layer_norm(want, (N,), g.double(), b.double(), 1e-12)
    assert rel_l2(c, want) <= 5e-6, rel_l2(c, want)
    assert torch.equal(c_hi.cpu(), c.cpu().to(pd)) and rel_l2(ops.join(c_hi, c_lo), c) <= 2e-5
    if not ln:
        _, _, plain = ops.gemm(a_hi, a_lo, w_hi, w_lo, bias=bias.cuda(), act=act, precision=precision, residual_planes=rp,
                               out_f32=True, out_planes=False)
        assert rel_l2(c, plain) <= 2e-6


@pytest.mark.parametrize("M,N,K", [(768, 768, 16384), (2304, 768, 8192), (700, 3072, 16384), (3072, 768, 16512)])
@pytest.mark.parametrize("precision", ["bf16x3", "fp16x2"])
def test_gemm_split_k_on_the_big_tile(ops, M, N, K, precision):
    """Weight-gradient shapes (a few thousand rows, K = the batch's rows): didi_gemm_small takes its partial sums from the staggered
    256x192 kernel (grid.y = K-splits).  Against fp64 on the same rounded operands, a ragged last row band (700), a last split
    shorter than the others (16512 = 129 x 128), and against the plain launch to summation-order rounding."""
    from didi_amd._lib import lib

    a, w, bias = rnd(M, K, seed=47), rnd(N, K, seed=48, scale=0.05), rnd(N, seed=49)
    pd = PD[precision]
    ws = lib.didi_gemm_small_workspace_bytes(M, N, K, ops.PRECISIONS[precision])
    tiles = ((M + 255) // 256) * (N // 192)
    assert ws > 0 and ws % (M * N * 4) == 0 and 192 < ws // (M * N * 4) * tiles <= 256  # one workgroup per CU: everything in ONE round
    a_hi, a_lo = ops.split(a.cuda(), dtype=pd)
    w_hi, w_lo = ops.split(w.cuda(), dtype=pd)
    _, _, c = ops.gemm(a_hi, a_lo, w_hi, w_lo, bias=bias.cuda(), precision=precision, out_f32=True, out_planes=False, small=())
    ar, wr = (ops.join(a_hi, a_lo).cpu(), ops.join(w_hi, w_lo).cpu()) if precision == "bf16x3" else ref_operands(precision, a, w, a_hi, w_hi, w_lo)
    want = _gemm_ref(ar, wr, bias, None, None)
    assert rel_l2(c, want) <= 5e-6, rel_l2(c, want)
    _, _, plain = ops.gemm(a_hi, a_lo, w_hi, w_lo, bias=bias.cuda(), precision=precision, out_f32=True, out_planes=False)
    assert rel_l2(c, plain) <= 2e-6


@pytest.mark.parametrize("M,N,K", [(16384, 768, 768), (8192, 2304, 768), (16384, 768, 3072), (16512, 1536, 768), (8192, 712, 3000), (8192, 512, 192)])
@pytest.mark.parametrize("precision", ["bf16x3", "fp16x2", "fp16x3"])
def test_gemm_wgrad_from_untransposed_planes(ops, M, N, K, precision):
    """didi_gemm_wgrad: dW [N, K] = dC^T A read from the row-major planes of dC [M, N] and A [M, K] (transposing LDS reads; no
    transposed copy).  Against fp64 on the same rounded operands -- the decoder's five weight shapes at the training batch, a last
    split shorter than the others (16512 rows), ragged last tiles on both sides (712 x 3000), the smallest output the form takes
    (two tiles, 128 splits of two K-tiles) -- and against the route it replaces
    (transposed planes through didi_gemm_small) to summation-order rounding."""
    dc, a = rnd(M, N, seed=53, scale=0.02), rnd(M, K, seed=54)
    pd = PD[precision]
    assert ops.wgrad_supported(M, N, K, precision)
    d_hi, d_lo = ops.split(dc.cuda(), dtype=pd)
    a_hi, a_lo = ops.split(a.cuda(), dtype=pd)
    dW = ops.gemm_wgrad(d_hi, d_lo, a_hi, a_lo, precision=precision)
    dr = d_hi.double().cpu() if precision == "fp16x2" else ops.join(d_hi, d_lo).double().cpu()  # fp16x2: the first operand enters as its hi plane
    want = dr.t() @ ops.join(a_hi, a_lo).double().cpu()
    assert dW.shape == (N, K) and rel_l2(dW, want) <= 5e-6, rel_l2(dW, want)
    dt_hi, dt_lo = ops.transpose_split(dc.cuda(), dtype=pd)
    at_hi, at_lo = ops.transpose_split(a.cuda(), dtype=pd)
    _, _, old = ops.gemm(dt_hi, dt_lo, at_hi, at_lo, precision=precision, out_f32=True, out_planes=False)
    assert rel_l2(dW, old) <= 2e-6, rel_l2(dW, old)


@pytest.mark.parametrize("M,N,K", [(16384, 768, 768), (8192, 2304, 768), (4096, 3072, 768), (8192, 768, 3072), (2100, 1536, 3000), (2048, 32, 192),
                                   (2304, 64, 384)])
@pytest.mark.parametrize("precision", ["bf16x3", "fp16x2", "fp16x3"])
def test_gemm_dgrad_from_the_forward_weight_planes(ops, M, N, K, precision):
    """didi_gemm_dgrad: dA [M, K] = dC W + residual with W [N, K] read as the forward holds it (transposing LDS reads on the weight
    side only).  Against fp64 on the same rounded operands -- the decoder's four shapes, ragged last tiles on both sides (2100 x
    3000), contractions of one and two K-tiles -- and against the route it replaces (didi_gemm on a transposed copy of W) to summation-order rounding."""
    dc, w, r = rnd(M, N, seed=57, scale=0.02), rnd(N, K, seed=58, scale=0.05), rnd(M, K, seed=59)
    pd = PD[precision]
    assert ops.dgrad_supported(M, N, K, precision)
    d_hi, d_lo = ops.split(dc.cuda(), dtype=pd)
    w_hi, w_lo = ops.split(w.cuda(), dtype=pd)
    dA = ops.gemm_dgrad(d_hi, d_lo, w_hi, w_lo, precision=precision, residual=r.cuda())
    dr = d_hi.double().cpu() if precision == "fp16x2" else ops.join(d_hi, d_lo).double().cpu()
    want = dr @ ops.join(w_hi, w_lo).double().cpu() + r.double()
    assert dA.shape == (M, K) and rel_l2(dA, want) <= 5e-6, rel_l2(dA, want)
    wt_hi, wt_lo = ops.transpose_split(w.cuda(), dtype=pd)
    _, _, old = ops.gemm(d_hi, d_lo, wt_hi, wt_lo, precision=precision, out_f32=True, out_planes=False, residual=r.cuda())
    assert rel_l2(dA, old) <= 2e-6, rel_l2(dA, old)
    plain = ops.gemm_dgrad(d_hi, d_lo, w_hi, w_lo, precision=precision)
    assert rel_l2(plain, want - r.double()) <= 5e-6


def test_gemm_wgrad_limits(ops):
    assert not ops.wgrad_supported(4096, 768, 768, "bf16x3")     # too few rows for the split-K form
    assert not ops.wgrad_supported(16384, 768, 768, "fp16")      # one-product precisions keep the transposing route
    assert not ops.wgrad_supported(16384, 8008, 768, "bf16x3")   # 128 output tiles fill the chip without a split
    assert not ops.wgrad_supported(16384, 768, 100, "bf16x3")
    assert not ops.wgrad_supported(16400, 768, 768, "bf16x3")    # rows not a multiple of the 32-row K-tile
    assert not ops.dgrad_supported(1024, 768, 768, "bf16x3")     # too few rows for the 256-row tile to pay
    assert not ops.dgrad_supported(16384, 8008, 768, "bf16x3")   # contraction not a multiple of the K-step (the classifier)
    assert not ops.dgrad_supported(16384, 768, 768, "bf16")


def test_gemm_small_batch_form_limits(ops):
    from didi_amd._lib import lib

    P = ops.PRECISIONS["fp16"]
    assert lib.didi_gemm_small_workspace_bytes(16384, 768, 768, P) == 0   # large batches fill the chip without a split
    assert lib.didi_gemm_small_workspace_bytes(2016, 3072, 768, P) == 0   # 384 workgroups already
    assert lib.didi_gemm_small_workspace_bytes(63, 768, 96, P) == 0       # K not a multiple of 128
    assert lib.didi_gemm_small_workspace_bytes(63, 8008, 768, P) == 0     # wider than the row kernel holds


@pytest.mark.parametrize("M,N,K", [(2304, 768, 768), (2100, 2304, 768), (2048, 8008, 768), (2050, 576, 3072), (4096, 3072, 128)])
@pytest.mark.parametrize("precision", ALL_PREC)
def test_gemm_big_tile(ops, M, N, K, precision):
    """M >= 2048 selects the 256x192 8-wave tile (M and N tails, all epilogue branches)."""
    a, w, bias, r = rnd(M, K, seed=21), rnd(N, K, seed=22, scale=0.05), rnd(N, seed=23), rnd(M, N, seed=24)
    a_hi, a_lo = ops.split(a.cuda(), dtype=PD[precision])
    w_hi, w_lo = ops.split(w.cuda(), dtype=PD[precision])
    rp = ops.split(r.cuda(), dtype=PD[precision])
    x3 = precision == "bf16x3"
    ar, wr = ref_operands(precision, a, w, a_hi, w_hi, w_lo)
    tol = 3e-5 if x3 else 2e-5
    c_hi, c_lo, c = ops.gemm(a_hi, a_lo, w_hi, w_lo, bias=bias.cuda(), precision=precision, out_f32=True)
    want = _gemm_ref(ar, wr, bias, None, None)
    assert rel_l2(c, want) <= tol, rel_l2(c, want)
    assert (c.cpu().double() - want).abs().max().item() <= 50 * tol * want.abs().max().item()
    assert is_nearest_plane(c_hi, c, PD[precision])
    assert rel_l2(ops.join(c_hi, c_lo), c) <= 2e-5
    _, _, c = ops.gemm(a_hi, a_lo, w_hi, w_lo, bias=bias.cuda(), act="gelu", precision=precision, residual_planes=rp,
                       out_f32=True, out_planes=False)
    assert rel_l2(c, _gemm_ref(ar, wr, bias, "gelu", ops.join(*rp).cpu())) <= tol


@pytest.mark.parametrize("M,K", [(2048, 768), (2100, 128), (4160, 3072), (16684, 128)])
@pytest.mark.parametrize("precision", ALL_PREC + ["fp16x3"])
def test_gemm_ln_fused(ops, M, K, precision):
    """didi_gemm_ln == LayerNorm(A W^T + bias + residual) (fp64 reference on the same rounded operands), ragged last
    row band, more bands than one launch holds (16684 rows = 66 bands > 256 CUs / 4), in place over the residual."""
    N = 768
    a, w, bias, r = rnd(M, K, seed=31), rnd(N, K, seed=32, scale=0.05), rnd(N, seed=33), rnd(M, N, seed=34)
    r[:, 5] += 30.0  # an outlier feature: large mean relative to the spread, as trained BERT hidden states have
    g, b = 1 + rnd(N, seed=35, scale=0.1), rnd(N, seed=36, scale=0.1)
    pd, x3 = PD[precision], precision == "bf16x3"
    a_hi, a_lo = ops.split(a.cuda(), dtype=pd)
    w_hi, w_lo = ops.split(w.cuda(), dtype=pd)
    rp = ops.split(r.cuda(), dtype=pd)
    r_used = ops.join(*rp).cpu()
    y_hi, y_lo, y = ops.gemm_ln(a_hi, a_lo, w_hi, w_lo, bias.cuda(), rp, g.cuda(), b.cuda(), 1e-12, precision=precision, f32=True)
    ar, wr = ref_operands(precision, a, w, a_hi, w_hi, w_lo)
    want = torch.nn.functional.layer_norm(_gemm_ref(ar, wr, bias, None, r_used), (N,), g.double(), b.double(), 1e-12)
    assert rel_l2(y, want) <= 5e-6, rel_l2(y, want)  # measured 2.5e-6 (bf16x3: dropped lo*lo term), 3e-7 (one product)
    assert (y.cpu().double() - want).abs().max().item() <= 1e-3
    assert is_nearest_plane(y_hi, y, pd) and rel_l2(ops.join(y_hi, y_lo), y) <= 2e-5
    # twice in a row in place (epoch 0: the call clears the statistics buffer itself), as the decoder engine uses it
    h_hi, h_lo = rp[0].clone(), rp[1].clone()
    ops.gemm_ln(a_hi, a_lo, w_hi, w_lo, bias.cuda(), (h_hi, h_lo), g.cuda(), b.cuda(), 1e-12, precision=precision, out=(h_hi, h_lo))
    assert torch.equal(h_hi, y_hi) and torch.equal(h_lo, y_lo)


@pytest.mark.parametrize("precision", ["fp16", "fp16x2"])
def test_gemm_ln_remainder_plane_as_bytes(ops, precision):
    """didi_gemm_ln_lo8: the residual stream's remainder plane as bytes (top byte of fp16(256 * (v - hi))), as the engine carries
    it between the fused launches of a one- / two-product call.  Written bytes decode to the remainder within the format's
    1 / 8 (3 significant bits) or 2^-25 absolute; the hi plane and the fp32 output are those of didi_gemm_ln; a byte-plane
    residual gives the output of the 16-bit-plane residual holding the same values; in place; other precisions refused."""
    M, N, K = 2100, 768, 768
    a, w, bias, r = rnd(M, K, seed=31), rnd(N, K, seed=32, scale=0.05), rnd(N, seed=33), rnd(M, N, seed=34)
    r[:, 5] += 30.0
    r[:, 7] *= 1e-3  # small values: remainders far below fp16's normal range
    g, b = (1 + rnd(N, seed=35, scale=0.1)).cuda(), rnd(N, seed=36, scale=0.1).cuda()
    a_hi, a_lo = ops.split(a.cuda(), dtype=torch.float16)
    w_hi, w_lo = ops.split(w.cuda(), dtype=torch.float16)
    rp = ops.split(r.cuda(), dtype=torch.float16)
    y_hi, y_lo, y = ops.gemm_ln(a_hi, a_lo, w_hi, w_lo, bias.cuda(), rp, g, b, 1e-12, precision=precision, f32=True)
    z_hi, z_b, z = ops.gemm_ln(a_hi, a_lo, w_hi, w_lo, bias.cuda(), rp, g, b, 1e-12, precision=precision, f32=True, lo_bytes=2)
    assert z_b.dtype == torch.uint8 and torch.equal(z_hi, y_hi) and torch.equal(z, y)
    dec = (z_b.to(torch.int16) << 8).view(torch.float16).double() / 256.0  # what the next launch re-joins
    rem = y.double() - y_hi.double()
    err = (dec - rem).abs()
    assert bool((err <= rem.abs() / 8 + 2.0 ** -25).all()), (err - rem.abs() / 8).max().item()
    assert rel_l2(y_hi.double() + dec, y) <= 2e-5  # 16-bit remainder plane: 2e-5 too (its own test); hi alone: 2.6e-4
    assert rel_l2(y_hi.double(), y) >= 1e-4
    # the byte plane as the residual of the next launch == a 16-bit plane holding the same remainders (exactly representable here:
    # |remainder| >= 2^-24 * 256 or zero is not guaranteed, so compare with a tolerance far below the plane's own step)
    lo16 = dec.to(torch.float16)
    want_hi, _, want = ops.gemm_ln(a_hi, a_lo, w_hi, w_lo, bias.cuda(), (y_hi, lo16), g, b, 1e-12, precision=precision, f32=True)
    got_hi, got_b, got = ops.gemm_ln(a_hi, a_lo, w_hi, w_lo, bias.cuda(), (y_hi, z_b), g, b, 1e-12, precision=precision, f32=True, lo_bytes=3)
    assert rel_l2(got, want) <= 1e-6 and (got_hi != want_hi).float().mean().item() <= 1e-3
    # in place over the residual, as the engine runs it
    h_hi, h_b = y_hi.clone(), z_b.clone()
    ops.gemm_ln(a_hi, a_lo, w_hi, w_lo, bias.cuda(), (h_hi, h_b), g, b, 1e-12, precision=precision, out=(h_hi, h_b), lo_bytes=3)
    assert torch.equal(h_hi, got_hi) and torch.equal(h_b, got_b)
    # residual as its hi plane alone, no remainder plane written
    n_hi, n_lo, n = ops.gemm_ln(a_hi, a_lo, w_hi, w_lo, bias.cuda(), (y_hi, None), g, b, 1e-12, precision=precision, f32=True, out=(torch.empty_like(y_hi), None))
    zero = torch.zeros_like(y_hi)
    w0_hi, _, w0 = ops.gemm_ln(a_hi, a_lo, w_hi, w_lo, bias.cuda(), (y_hi, zero), g, b, 1e-12, precision=precision, f32=True)
    assert n_lo is None and torch.equal(n, w0) and torch.equal(n_hi, w0_hi)
    with pytest.raises(RuntimeError):
        b_hi, b_lo = ops.split(a.cuda(), dtype=torch.bfloat16)
        ops.gemm_ln(b_hi, b_lo, *ops.split(w.cuda(), dtype=torch.bfloat16), bias.cuda(), ops.split(r.cuda(), dtype=torch.bfloat16), g, b, 1e-12,
                    precision="bf16x3", lo_bytes=2)


@pytest.mark.parametrize("M,N,K", [(300, 768, 768), (2304, 8008, 768), (63, 768, 768)])
def test_gemm_and_rounding_fp16x3(ops, M, N, K):
    """DIDI_PREC_FP16X3 (all three products on fp16 planes, ~22 bits; the final rounding / logits of the fp16x2 mode): 128x128 tile,
    256x192 tile and the split-K small-batch form against fp64 on the UN-rounded operands at 5e-6; rounding ids exact where the
    top-2 gap is not a near-tie (the LayerNorm-fused launch in this precision: test_gemm_ln_fused)."""
    from didi_amd._lib import lib

    a, w, bias = rnd(M, K, seed=61), rnd(N, K, seed=62, scale=0.05), rnd(N, seed=63)
    a_hi, a_lo = ops.split(a.cuda(), dtype=torch.float16)
    w_hi, w_lo = ops.split(w.cuda(), dtype=torch.float16)
    want = _gemm_ref(a, w, bias, None, None)
    c_hi, c_lo, c = ops.gemm(a_hi, a_lo, w_hi, w_lo, bias=bias.cuda(), precision="fp16x3", out_f32=True)
    assert rel_l2(c, want) <= 5e-6, rel_l2(c, want)
    assert c_hi.dtype == torch.float16 and rel_l2(ops.join(c_hi, c_lo), c) <= 2e-5
    if lib.didi_gemm_small_workspace_bytes(M, N, K, ops.PRECISIONS["fp16x3"]):
        _, _, cs = ops.gemm(a_hi, a_lo, w_hi, w_lo, bias=bias.cuda(), precision="fp16x3", out_f32=True, out_planes=False, small=())
        assert rel_l2(cs, want) <= 5e-6
    ids = ops.round_argmax(a_hi, a_lo, w_hi, w_lo, "fp16x3").cpu()
    logits = a.double() @ w.double().t()
    top2 = logits.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5 * top2[:, 0].abs().clamp_min(1e-3)
    assert torch.equal(ids[clear], logits.argmax(-1)[clear])
    assert lib.didi_gemm_ln_supported(4096, 768, 768, ops.PRECISIONS["fp16x3"]) == 1


def test_gemm_ln_rejects_unsupported(ops):
    from didi_amd._lib import DidiHipError, lib

    assert lib.didi_gemm_ln_supported(16384, 768, 3072, ops.PRECISIONS["fp16"]) == 1
    assert lib.didi_gemm_ln_supported(16384, 512, 768, ops.PRECISIONS["fp16"]) == 0  # not the BertConfig hidden size
    assert lib.didi_gemm_ln_supported(184, 768, 768, ops.PRECISIONS["fp16"]) == 0   # small batches take GEMM + LayerNorm
    a_hi, _ = ops.split(rnd(64, 64).cuda(), dtype=torch.float16)
    w_hi, w_lo = ops.split(rnd(768, 64).cuda(), dtype=torch.float16)
    rp = ops.split(rnd(64, 768).cuda(), dtype=torch.float16)
    with pytest.raises(DidiHipError):
        ops.gemm_ln(a_hi, None, w_hi, w_lo, torch.zeros(768).cuda(), rp, torch.ones(768).cuda(), torch.zeros(768).cuda(), 1e-12)


def test_gemm_big_tile_identity_asymmetric(ops):
    M, K = 2304, 192
    a = torch.zeros(M, K)
    a[torch.arange(M), torch.arange(M) % K] = 1.0  # row m selects column m % K of W^T
    w = (torch.arange(768 * K, dtype=torch.float32).reshape(768, K) % 251) - 125  # exact in bf16
    for precision in ALL_PREC:
        a_hi, a_lo = ops.split(a.cuda(), dtype=PD[precision])
        w_hi, w_lo = ops.split(w.cuda(), dtype=PD[precision])
        _, _, c = ops.gemm(a_hi, a_lo, w_hi, w_lo, precision=precision, out_f32=True, out_planes=False)
        assert torch.equal(c.cpu(), w.t()[torch.arange(M) % K].contiguous())


def test_gemm_identity_asymmetric(ops):
    """A = I with an asymmetric W: catches a transposed or permuted C write (exact integers)."""
    K = 128
    a = torch.eye(K)
    w = (torch.arange(192 * K, dtype=torch.float32).reshape(192, K) % 251) - 125  # exact in bf16
    a_hi, a_lo = ops.split(a.cuda())
    w_hi, w_lo = ops.split(w.cuda())
    _, _, c = ops.gemm(a_hi, a_lo, w_hi, w_lo, precision="bf16", out_f32=True, out_planes=False)
    assert torch.equal(c.cpu(), w.t().contiguous())


def test_gemm_rejects_bad_shapes(ops):
    from didi_amd._lib import DidiHipError

    a_hi, a_lo = ops.split(rnd(8, 20).cuda())
    w_hi, w_lo = ops.split(rnd(8, 20).cuda())
    with pytest.raises(DidiHipError):
        ops.gemm(a_hi, a_lo, w_hi, w_lo)  # K % 8 != 0


# ------------------------------------------------------------------ rounding (classifier + argmax)
@pytest.mark.parametrize("M,V,K", [(184, 8008, 128), (77, 600, 128), (300, 8008, 768), (5, 130, 64), (168, 50000, 768),
                                   (2304, 8008, 768), (2100, 1000, 128), (2048, 50000, 768)])  # >= 2048 rows: the 256x192 kernel
@pytest.mark.parametrize("precision", ALL_PREC)
def test_round_argmax(ops, M, V, K, precision):
    x, w = rnd(M, K, seed=11), rnd(V, K, seed=12, scale=0.02)
    x_hi, x_lo = ops.split(x.cuda(), dtype=PD[precision])
    w_hi, w_lo = ops.split(w.cuda(), dtype=PD[precision])
    ids = ops.round_argmax(x_hi, x_lo, w_hi, w_lo, precision).cpu()
    xr, wr = ref_operands(precision, x, w, x_hi, w_hi, w_lo)
    logits = xr.double() @ wr.double().t()
    want = logits.argmax(-1)
    top2 = logits.topk(2, dim=-1).values
    gap = top2[:, 0] - top2[:, 1]
    clear = gap > 1e-4 * top2[:, 0].abs().clamp_min(1e-3)  # ids must agree wherever the max is not a near-tie
    assert torch.equal(ids[clear], want[clear]), (ids[clear] != want[clear]).sum().item()
    picked = logits.gather(1, ids[:, None])[:, 0]
    assert ((top2[:, 0] - picked).abs() <= 1e-4 * top2[:, 0].abs().clamp_min(1e-3)).all()


@pytest.mark.parametrize("M", [40, 2100])
def test_round_argmax_first_index_wins_ties(ops, M):
    V, K = 700, 128
    x, w = rnd(M, K, seed=13), rnd(V, K, seed=14, scale=0.05)
    w[650] = w[3]
    w[400] = w[3]
    x[:] = w[3] * 50  # rows aligned with w[3]: rows 3, 400, 650 of W give bitwise-identical maxima
    for precision in ALL_PREC:
        x_hi, x_lo = ops.split(x.cuda(), dtype=PD[precision])
        w_hi, w_lo = ops.split(w.cuda(), dtype=PD[precision])
        ids = ops.round_argmax(x_hi, x_lo, w_hi, w_lo, precision).cpu()
        assert (ids == 3).all(), ids


@pytest.mark.parametrize("M,V,K", [(184, 8008, 128), (300, 8008, 768), (5, 130, 64), (1, 100, 64), (168, 50000, 768)])
@pytest.mark.parametrize("precision", ALL_PREC)
def test_round_ce(ops, M, V, K, precision):
    """didi_round_ce: argmax + per-token cross-entropy terms (logsumexp - target logit) of X W^T against fp64 logits of the
    same rounded operands (what F.cross_entropy(reduction='none') gives); a target outside [0, V) yields +inf."""
    x, w = rnd(M, K, seed=21, scale=3.0), rnd(V, K, seed=22, scale=0.05)
    tg = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(23))
    x_hi, x_lo = ops.split(x.cuda(), dtype=PD[precision])
    w_hi, w_lo = ops.split(w.cuda(), dtype=PD[precision])
    ids, nll = ops.round_ce(x_hi, x_lo, w_hi, w_lo, tg.cuda(), precision)
    xr, wr = ref_operands(precision, x, w, x_hi, w_hi, w_lo)
    logits = xr.double() @ wr.double().t()
    want = torch.nn.functional.cross_entropy(logits, tg, reduction="none")
    tol = 2e-4 if precision == "bf16x3" else 2e-5  # bf16x3 drops the lo*lo product: logits to ~1e-5 of |x||w|
    assert (nll.cpu().double() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item()), (nll.cpu().double() - want).abs().max()
    assert torch.equal(ids.cpu(), ops.round_argmax(x_hi, x_lo, w_hi, w_lo, precision).cpu())
    bad = tg.clone()
    bad[0] = V
    _, nll2 = ops.round_ce(x_hi, x_lo, w_hi, w_lo, bad.cuda(), precision)
    assert torch.isinf(nll2[0]).item() and nll2[0].item() > 0 and torch.equal(nll2[1:], nll[1:])


def test_first_eos(ops):
    ids = torch.randint(5, 100, (9, 23), generator=torch.Generator().manual_seed(1))
    ids[0, 0] = 2
    ids[1, 22] = 2
    ids[2, 5] = 2
    ids[2, 9] = 2
    lengths = ops.first_eos(ids.cuda(), 2).cpu().tolist()
    want = [1, 23, 6] + [23] * 6
    assert lengths == want


# ------------------------------------------------------------------ attention
def _attn_ref(q, k, v, heads, causal, key_mask):
    B, Sq, HD = q.shape
    Sk, d = k.shape[1], HD // heads
    qh = q.double().view(B, Sq, heads, d).transpose(1, 2)
    kh = k.double().view(B, Sk, heads, d).transpose(1, 2)
    vh = v.double().view(B, Sk, heads, d).transpose(1, 2)
    s = qh @ kh.transpose(2, 3) * d**-0.5
    neg = torch.finfo(torch.float32).min
    if causal:
        s = s.masked_fill(~torch.tril(torch.ones(Sq, Sk, dtype=torch.bool)), neg)
    if key_mask is not None:
        s = s.masked_fill(key_mask[:, None, None, :] == 0, neg)
    p = torch.softmax(s, -1)
    return (p @ vh).transpose(1, 2).reshape(B, Sq, HD)


@pytest.mark.parametrize("B,heads,Sq,Sk,d,causal,masked", [
    (3, 4, 23, 23, 32, True, False), (2, 4, 11, 12, 32, False, True), (2, 12, 64, 64, 64, True, False),
    (2, 12, 64, 64, 64, False, True), (2, 3, 84, 84, 64, True, False), (2, 3, 83, 84, 64, False, True),
    (2, 4, 12, 12, 24, False, True), (1, 32, 24, 24, 40, False, True), (2, 2, 1, 1, 64, True, False),
    (1, 2, 96, 96, 64, True, False),
])
@pytest.mark.parametrize("precision", ALL_PREC)
def test_attention(ops, B, heads, Sq, Sk, d, causal, masked, precision):
    HD = heads * d
    qkv = rnd(B, max(Sq, Sk), 3 * HD, seed=Sq * 7 + d, scale=1.5)
    q, k, v = qkv[:, :Sq, :HD], qkv[:, :Sk, HD:2 * HD], qkv[:, :Sk, 2 * HD:]
    key_mask = None
    if masked:
        lens = torch.tensor([Sk, max(1, Sk // 3), 2, 1][:B])
        key_mask = (torch.arange(Sk)[None] < lens[:, None]).int()
    x3 = precision == "bf16x3"
    hi, lo = ops.split(qkv.cuda(), need_lo=x3, dtype=PD[precision])

    def view(t, s, c0):
        return None if t is None else t[:, :s, c0:c0 + HD]

    if Sq == Sk:  # strided views of one fused buffer (ld = 3*HD), as the decoder uses them
        qp, kp, vp = [(view(hi, Sk, i * HD), view(lo, Sk, i * HD)) for i in range(3)]
    else:
        qp = tuple(None if t is None else t[:, :Sq, :HD].contiguous() for t in (hi, lo))
        kv_hi = hi[:, :Sk, HD:].contiguous()
        kv_lo = None if lo is None else lo[:, :Sk, HD:].contiguous()
        kp = (kv_hi[:, :, :HD], None if kv_lo is None else kv_lo[:, :, :HD])
        vp = (kv_hi[:, :, HD:], None if kv_lo is None else kv_lo[:, :, HD:])
    o_hi, o_lo = ops.attention(qp, kp, vp, heads, causal=causal, key_mask=None if key_mask is None else key_mask.cuda(),
                               precision=precision)
    if x3:
        want = _attn_ref(q, k, v, heads, causal, key_mask)
        got, tol = ops.join(o_hi, o_lo), 5e-5
    else:
        f = lambda t: t.to(PD[precision]).float()  # noqa: E731
        want = _attn_ref(f(q), f(k), f(v), heads, causal, key_mask)
        got, tol = o_hi.float(), (1.2e-2 if precision == "bf16" else 1.5e-3)  # P and O are additionally rounded to 16 bits
    assert rel_l2(got, want) <= tol, rel_l2(got, want)
    assert torch.isfinite(got).all()
    if masked:  # right-padded masks: with the per-dialog key counts the padded keys are skipped -- same bits out
        counts = ops.key_counts(key_mask.cuda())
        assert counts.cpu().tolist() == lens.tolist()
        c_hi, c_lo = ops.attention(qp, kp, vp, heads, causal=causal, key_mask=key_mask.cuda(), precision=precision, key_count=counts)
        assert torch.equal(c_hi, o_hi) and (o_lo is None or torch.equal(c_lo, o_lo))


def test_key_counts_only_trim_prefix_masks(ops):
    mask = torch.tensor([[1, 1, 1, 0, 0, 0], [1, 0, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1], [0, 1, 1, 1, 1, 1]], dtype=torch.int32)
    assert ops.key_counts(mask.cuda()).cpu().tolist() == [3, 6, 0, 6, 6]  # holes / left padding: no trimming (count = Sk)
    # a count of 0 (nothing visible) keeps the reference's uniform-attention corner: same output as without counts
    B, S, HD, heads = 5, 6, 64, 2
    qkv = rnd(B, S, 3 * HD, seed=3)
    hi, lo = ops.split(qkv.cuda())
    v3 = lambda t, i: t[:, :, i * HD:(i + 1) * HD]  # noqa: E731
    q, k, v = ((v3(hi, i), v3(lo, i)) for i in range(3))
    a_hi, a_lo = ops.attention(q, k, v, heads, key_mask=mask.cuda(), precision="bf16x3")
    b_hi, b_lo = ops.attention(q, k, v, heads, key_mask=mask.cuda(), precision="bf16x3", key_count=ops.key_counts(mask.cuda()))
    assert torch.equal(a_hi, b_hi) and torch.equal(a_lo, b_lo)


@pytest.mark.parametrize("B,heads,S,d", [(2, 4, 40, 16), (3, 4, 23, 32), (1, 8, 96, 64)])
@pytest.mark.parametrize("precision", ALL_PREC)
def test_attention_position_bias_and_scale(ops, B, heads, S, d, precision):
    """T5Attention: softmax(q k^T * 1 + position_bias + mask) v -- explicit scale and an additive [heads, Sq, Sk] bias."""
    HD = heads * d
    qkv = rnd(B, S, 3 * HD, seed=S + d, scale=0.6)
    bias = rnd(heads, S, S, seed=S, scale=2.0)
    lens = torch.tensor([S, max(1, S // 3), 2][:B])
    key_mask = (torch.arange(S)[None] < lens[:, None]).int()
    x3 = precision == "bf16x3"
    hi, lo = ops.split(qkv.cuda(), need_lo=x3, dtype=PD[precision])
    parts = [(hi[:, :, i * HD:(i + 1) * HD], None if lo is None else lo[:, :, i * HD:(i + 1) * HD]) for i in range(3)]
    o_hi, o_lo = ops.attention(*parts, heads, causal=False, key_mask=key_mask.cuda(), precision=precision, scale=1.0, pos_bias=bias.cuda())
    f = (lambda t: t) if x3 else (lambda t: t.to(PD[precision]).float())
    q, k, v = (f(qkv[:, :, i * HD:(i + 1) * HD]).double().view(B, S, heads, d).transpose(1, 2) for i in range(3))
    sc = q @ k.transpose(2, 3) + bias.double()[None]
    sc = sc.masked_fill(key_mask[:, None, None, :] == 0, torch.finfo(torch.float32).min)
    want = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B, S, HD)
    got, tol = (ops.join(o_hi, o_lo), 5e-5) if x3 else (o_hi.float(), 1.2e-2 if precision == "bf16" else 1.5e-3)
    assert rel_l2(got, want) <= tol, rel_l2(got, want)


def test_attention_all_keys_masked_is_uniform(ops):
    """HF additive-mask semantics: a dialog whose context is fully padded averages V uniformly."""
    B, heads, S, d = 2, 2, 8, 32
    qkv = rnd(B, S, 3 * heads * d, seed=3)
    hi, lo = ops.split(qkv.cuda())
    HD = heads * d
    parts = [(hi[:, :, i * HD:(i + 1) * HD], lo[:, :, i * HD:(i + 1) * HD]) for i in range(3)]
    mask = torch.zeros(B, S, dtype=torch.int32)
    mask[1, :3] = 1
    o_hi, o_lo = ops.attention(*parts, heads, causal=False, key_mask=mask.cuda(), precision="bf16x3")
    got = ops.join(o_hi, o_lo).cpu()
    v = qkv[:, :, 2 * HD:]
    assert torch.allclose(got[0], v[0].mean(0, keepdim=True).expand(S, HD), atol=1e-4)
    assert torch.isfinite(got).all()


def test_attention_fp16x3_three_products(ops):
    """didi_attention in DIDI_PREC_FP16X3: fp16 planes with remainders, three products in QK^T and PV, against fp64 on the un-rounded
    q, k, v at the bf16x3 bar or better."""
    B, heads, S, d = 3, 4, 40, 64
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(B, S, heads * d, generator=g) for _ in range(3))
    mask = torch.ones(B, S, dtype=torch.int32)
    mask[1, 25:] = 0
    qp, kp, vp = (ops.split(t.cuda(), dtype=torch.float16) for t in (q, k, v))
    o_hi, o_lo = ops.attention(qp, kp, vp, heads, causal=False, key_mask=mask.cuda(), precision="fp16x3")
    want = _attn_ref(q, k, v, heads, False, mask)
    assert o_lo is not None and rel_l2(ops.join(o_hi, o_lo), want) <= 2e-5
    o1, _ = ops.attention(qp, kp, vp, heads, causal=False, key_mask=mask.cuda(), precision="fp16")
    assert rel_l2(o1.float(), want) > rel_l2(ops.join(o_hi, o_lo), want)


@pytest.mark.parametrize("H", [2560, 4096, 2052])
def test_layernorm_and_rmsnorm_wide_rows(H):
    """Context encoders wider than 2048 (Blenderbot-3B 2560, T5 v1.1 xxl 4096)."""
    from didi_amd import ops

    g = torch.Generator().manual_seed(H)
    x = torch.randn(37, H, generator=g) * 2 + 0.3
    gamma, beta = 1 + 0.1 * torch.randn(H, generator=g), 0.1 * torch.randn(H, generator=g)
    hi, lo, y = ops.layernorm(x.cuda(), gamma.cuda(), beta.cuda(), 1e-5, f32=True)
    want = torch.nn.functional.layer_norm(x.double(), (H,), gamma.double(), beta.double(), 1e-5)
    assert rel_l2(y, want) <= 1e-6 and rel_l2(ops.join(hi, lo), want) <= 2e-5
    hi, lo, y = ops.rmsnorm(x.cuda(), gamma.cuda(), 1e-6, f32=True)
    want = x.double() * torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + 1e-6) * gamma.double()
    assert rel_l2(y, want) <= 1e-6 and rel_l2(ops.join(hi, lo), want) <= 2e-5


@pytest.mark.parametrize("B,heads,Sq,Sk,D", [(2, 2, 20, 20, 80), (2, 4, 64, 64, 128), (1, 3, 96, 96, 80), (2, 2, 33, 47, 96)])
def test_attention_train_forward_wide_heads(B, heads, Sq, Sk, D):
    """Head dims past 64 (the fp32 attention the wide-head context encoders run through), with a key padding mask."""
    import math

    from didi_amd import ops

    g = torch.Generator().manual_seed(D + Sq)
    H = heads * D
    q, k, v = (torch.randn(B, s_, H, generator=g) for s_ in (Sq, Sk, Sk))
    mask = torch.ones(B, Sk, dtype=torch.int32)
    mask[0, Sk - 5:] = 0
    sh = lambda t, S: t.double().view(B, S, heads, D).transpose(1, 2)  # noqa: E731
    s = sh(q, Sq) @ sh(k, Sk).transpose(-1, -2) / math.sqrt(D) + (1.0 - mask[:, None, None, :].double()) * torch.finfo(torch.float32).min
    want = (torch.softmax(s, -1) @ sh(v, Sk)).transpose(1, 2).reshape(B, Sq, H)
    got = ops.attention_train(q.cuda(), k.cuda(), v.cuda(), heads, key_mask=mask.cuda())
    assert rel_l2(got, want) <= 2e-6


@pytest.mark.parametrize("precision", ["fp16", "fp16x2"])
@pytest.mark.parametrize("B", [32, 33, 41])
def test_qkv_self_attention_fused_equals_the_two_launches(ops, B, precision):
    """didi_qkv_self_attention (projections + causal self-attention in one launch, S = 64, 64-wide heads) against didi_gemm into
    [M, 3H] planes followed by didi_attention(causal): bit-identical outputs (same conversions, same MFMA and summation order),
    for whole row bands (B = 32), a ragged last band (33, 41 dialogs), with and without the activation dither; and against an
    fp64 reference of the attention on the rounded projections."""
    import ctypes as C

    from didi_amd import _lib

    heads, S, H = 12, 64, 768
    M = B * S
    x = rnd(M, H, seed=71)
    w, bias = rnd(3 * H, H, seed=72, scale=0.05), rnd(3 * H, seed=73, scale=0.5)
    x_hi, x_lo = ops.split(x.cuda(), dtype=torch.float16)
    w_hi, w_lo = ops.split(w.cuda(), dtype=torch.float16)
    for dither in (0, 3072):
        g = _lib.GemmArgs()
        g.M, g.N, g.K, g.precision = M, 3 * H, H, ops.PRECISIONS[precision]
        g.A_hi, g.A_lo, g.lda = x_hi.data_ptr(), x_lo.data_ptr(), H
        g.W_hi, g.W_lo, g.ldw = w_hi.data_ptr(), w_lo.data_ptr(), H
        b_dev = bias.cuda()
        g.bias, g.act, g.act_dither = b_dev.data_ptr(), _lib.ACT_NONE, dither if precision == "fp16" else 0
        # the two launches
        qkv = torch.empty((M, 3 * H), dtype=torch.float16, device="cuda")
        g.C_hi, g.ldc = qkv.data_ptr(), 3 * H
        _lib.check(_lib.lib.didi_gemm(C.byref(g), _lib.current_stream()), "gemm")
        want = torch.empty((M, H), dtype=torch.float16, device="cuda")
        a = _lib.AttnArgs()
        a.B, a.heads, a.Sq, a.Sk, a.D = B, heads, S, S, 64
        a.precision, a.causal, a.act_dither = ops.PRECISIONS["fp16"], 1, g.act_dither
        a.Q_hi, a.ldq, a.K_hi, a.ldk, a.V_hi, a.ldv = qkv.data_ptr(), 3 * H, qkv.data_ptr() + 2 * H, 3 * H, qkv.data_ptr() + 4 * H, 3 * H
        a.O_hi, a.ldo = want.data_ptr(), H
        _lib.check(_lib.lib.didi_attention(C.byref(a), _lib.current_stream()), "attention")
        # one launch
        got = torch.full((M, H), float("nan"), dtype=torch.float16, device="cuda")
        g.C_hi, g.ldc = got.data_ptr(), H
        assert _lib.lib.didi_qkv_self_attention_supported(M, H, heads, S, H, g.precision)
        _lib.check(_lib.lib.didi_qkv_self_attention(C.byref(g), heads, S, 0.0, _lib.current_stream()), "qkv_self_attention")
        torch.cuda.synchronize()
        assert torch.equal(got, want), f"B={B} {precision} dither={dither}: {(got.float() - want.float()).abs().max().item()}"
    # fp64 attention on the rounded projections of the undithered run
    q, k, v = (qkv[:, i * H:(i + 1) * H].double().cpu().view(B, S, heads, 64).transpose(1, 2) for i in range(3))
    sc = q @ k.transpose(-1, -2) / 8.0
    sc = sc.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf"))
    ref = (sc.softmax(-1) @ v).transpose(1, 2).reshape(M, H)
    assert rel_l2(want, ref) <= 1e-3
    # shapes the fused form does not take
    assert not _lib.lib.didi_qkv_self_attention_supported(M, H, heads, 63, H, g.precision)
    assert not _lib.lib.didi_qkv_self_attention_supported(M, H, heads, S, H, ops.PRECISIONS["bf16x3"])
    assert not _lib.lib.didi_qkv_self_attention_supported(1024, H, heads, S, H, g.precision)


@pytest.mark.parametrize("precision", ["fp16", "fp16x2"])
@pytest.mark.parametrize("B", [32, 33, 41])
def test_q_cross_attention_fused_equals_the_two_launches(ops, B, precision):
    """didi_q_cross_attention (cross-attention query projection + attention over the hoisted keys / values in one launch,
    S = C = 64, 64-wide heads) against didi_gemm followed by didi_attention(key_mask, key_count): bit-identical outputs for whole
    row bands (B = 32) and ragged last bands (33, 41 dialogs), with right-padded contexts of every length class (1 key, a
    partial 16-key tile, exactly 32, 63, all 64, and the nothing-visible corner), with and without key counts, with and
    without the activation dither; and against an fp64 reference of the attention on the rounded queries."""
    import ctypes as C

    from didi_amd import _lib

    heads, S, Cn, H = 12, 64, 64, 768
    M = B * S
    x = rnd(M, H, seed=81)
    w, bias = rnd(H, H, seed=82, scale=0.05), rnd(H, seed=83, scale=0.5)
    kv = rnd(B * Cn, 2 * H, seed=84)
    x_hi, x_lo = ops.split(x.cuda(), dtype=torch.float16)
    w_hi, w_lo = ops.split(w.cuda(), dtype=torch.float16)
    kv_hi = kv.cuda().to(torch.float16)
    lens = torch.tensor([1, 5, 16, 17, 31, 32, 33, 48, 63, 64, 0, 24], dtype=torch.int64).repeat((B + 11) // 12)[:B]
    mask = (torch.arange(Cn)[None, :] < lens[:, None]).to(torch.int32)
    mask[3, 2] = 0  # a hole inside the visible prefix: the mask, not the count, hides it
    mask_dev = mask.cuda()
    counts = torch.empty(B, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib.didi_key_counts(mask_dev.data_ptr(), B, Cn, counts.data_ptr(), _lib.current_stream()), "key_counts")
    b_dev = bias.cuda()
    for dither, use_counts in ((0, True), (3072, True), (0, False)):
        g = _lib.GemmArgs()
        g.M, g.N, g.K, g.precision = M, H, H, ops.PRECISIONS[precision]
        g.A_hi, g.A_lo, g.lda = x_hi.data_ptr(), x_lo.data_ptr(), H
        g.W_hi, g.W_lo, g.ldw = w_hi.data_ptr(), w_lo.data_ptr(), H
        g.bias, g.act, g.act_dither = b_dev.data_ptr(), _lib.ACT_NONE, dither if precision == "fp16" else 0
        cnt_ptr = counts.data_ptr() if use_counts else None
        # the two launches
        q = torch.empty((M, H), dtype=torch.float16, device="cuda")
        g.C_hi, g.ldc = q.data_ptr(), H
        _lib.check(_lib.lib.didi_gemm(C.byref(g), _lib.current_stream()), "gemm")
        want = torch.empty((M, H), dtype=torch.float16, device="cuda")
        a = _lib.AttnArgs()
        a.B, a.heads, a.Sq, a.Sk, a.D = B, heads, S, Cn, 64
        a.precision, a.causal, a.act_dither = ops.PRECISIONS["fp16"], 0, g.act_dither
        a.Q_hi, a.ldq = q.data_ptr(), H
        a.K_hi, a.ldk, a.V_hi, a.ldv = kv_hi.data_ptr(), 2 * H, kv_hi.data_ptr() + 2 * H, 2 * H
        a.key_mask, a.key_count = mask_dev.data_ptr(), cnt_ptr
        a.O_hi, a.ldo = want.data_ptr(), H
        _lib.check(_lib.lib.didi_attention(C.byref(a), _lib.current_stream()), "attention")
        # one launch
        got = torch.full((M, H), float("nan"), dtype=torch.float16, device="cuda")
        g.C_hi, g.ldc = got.data_ptr(), H
        assert _lib.lib.didi_q_cross_attention_supported(M, H, heads, S, Cn, H, g.precision)
        _lib.check(_lib.lib.didi_q_cross_attention(C.byref(g), kv_hi.data_ptr(), kv_hi.data_ptr() + 2 * H, 2 * H, mask_dev.data_ptr(), cnt_ptr,
                                                    heads, S, Cn, 0.0, _lib.current_stream()), "q_cross_attention")
        torch.cuda.synchronize()
        assert torch.equal(got, want), (f"B={B} {precision} dither={dither} counts={use_counts}: "
                                        f"{(got.float() - want.float()).abs().max().item()} at rows "
                                        f"{(got != want).any(1).nonzero().flatten()[:8].tolist()}")
    # fp64 attention on the rounded queries of the last (undithered) run; the nothing-visible dialogs attend uniformly
    qd = q.double().cpu().view(B, S, heads, 64).transpose(1, 2)
    kd, vd = (kv_hi[:, i * H:(i + 1) * H].double().cpu().view(B, Cn, heads, 64).transpose(1, 2) for i in range(2))
    sc = qd @ kd.transpose(-1, -2) / 8.0 + (1.0 - mask[:, None, None, :].double()) * torch.finfo(torch.float32).min
    ref = (sc.float().double().softmax(-1) @ vd).transpose(1, 2).reshape(M, H)
    assert rel_l2(want, ref) <= 1e-3
    # shapes the fused form does not take
    assert not _lib.lib.didi_q_cross_attention_supported(M, H, heads, S, 63, H, g.precision)
    assert not _lib.lib.didi_q_cross_attention_supported(M, 512, 8, S, Cn, 512, g.precision)
    assert not _lib.lib.didi_q_cross_attention_supported(M, H, heads, S, Cn, H, ops.PRECISIONS["fp16x3"])
    assert not _lib.lib.didi_q_cross_attention_supported(1024, H, heads, S, Cn, H, g.precision)
