"""Tensor-level wrappers over the C ABI (one per entry point).

torch here is plumbing only: it owns device memory and the current HIP stream; every function
below passes raw device pointers to libdidi_hip and returns torch tensors that alias the outputs.
Nothing falls back to eager torch arithmetic.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import lib

PRECISIONS = {"bf16": _lib.PREC_BF16, "fp16": _lib.PREC_FP16, "bf16x3": _lib.PREC_BF16X3, "fp16x2": _lib.PREC_FP16X2,
              "fp16x3": _lib.PREC_FP16X3}  # fp16x3: rounding / logits GEMMs only (three products on fp16 planes)
ACTS = {None: _lib.ACT_NONE, "none": _lib.ACT_NONE, "gelu": _lib.ACT_GELU, "tanh": _lib.ACT_TANH, "relu": _lib.ACT_RELU,
        "gelu_new": _lib.ACT_GELU_NEW}


def _prec(p):
    return PRECISIONS[p] if isinstance(p, str) else int(p)


def plane_dtype(precision):
    """torch dtype of the split planes a precision works on (fp16 planes for "fp16", bf16 otherwise)."""
    return torch.float16 if _prec(precision) in (_lib.PREC_FP16, _lib.PREC_FP16X2, _lib.PREC_FP16X3) else torch.bfloat16


def _fmt(dtype):
    return _lib.FMT_FP16 if dtype == torch.float16 else _lib.FMT_BF16


def _f32c(t):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda, (t.dtype, t.is_contiguous(), t.device)
    return t


def split(x, need_lo=True, dtype=torch.bfloat16, out=None):
    """fp32 tensor -> (hi, lo) planes of the same shape, bf16 (default) or fp16.  ``out`` = (hi, lo): written in place."""
    x = _f32c(x)
    if out is not None:
        hi, lo = out
        assert hi.is_contiguous() and hi.dtype == dtype and hi.numel() == x.numel() and (lo is None or (lo.is_contiguous() and lo.dtype == dtype and lo.numel() == x.numel()))
    else:
        hi = torch.empty_like(x, dtype=dtype)
        lo = torch.empty_like(x, dtype=dtype) if need_lo else None
    fn = lib.didi_split_fp16 if dtype == torch.float16 else lib.didi_split_bf16
    _lib.check(fn(x.data_ptr(), hi.data_ptr(), _lib.ptr(lo), x.numel(), _lib.current_stream()), "split")
    return hi, lo


def join(hi, lo=None):
    """Test helper: planes -> fp32 (torch arithmetic; never used on the product path)."""
    return hi.float() if lo is None else hi.float() + lo.float()


def philox_normal(shape, seed, offset=0, device="cuda"):
    out = torch.empty(shape, dtype=torch.float32, device=device)
    _lib.check(lib.didi_philox_normal(out.data_ptr(), out.numel(), seed, offset, _lib.current_stream()), "philox")
    return out


def embed_gather(ids, table, scale=1.0, pos=None):
    """ids int64 [B, S] -> fp32 [B, S, H] = table[ids] * scale (+ pos[s])."""
    B, S = ids.shape
    H = table.shape[1]
    ids = ids.contiguous()
    out = torch.empty((B, S, H), dtype=torch.float32, device=ids.device)
    _lib.check(lib.didi_embed_gather(ids.data_ptr(), table.data_ptr(), scale, _lib.ptr(pos), B * S, S, H, out.data_ptr(),
                                     _lib.current_stream()), "embed_gather")
    return out


def layernorm(x, gamma, beta, eps, planes=True, need_lo=True, f32=False, dtype=torch.bfloat16):
    x = _f32c(x)
    M, H = x.numel() // x.shape[-1], x.shape[-1]
    hi = torch.empty_like(x, dtype=dtype) if planes else None
    lo = torch.empty_like(x, dtype=dtype) if planes and need_lo else None
    y = torch.empty_like(x) if f32 else None
    _lib.check(lib.didi_layernorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, M, H, _lib.ptr(hi), _lib.ptr(lo),
                                  _lib.ptr(y), _fmt(dtype), _lib.current_stream()), "layernorm")
    return hi, lo, y


def new_fault(device):
    """A zeroed device fault word (uint32 as one int32 element); pass it as ``fault=`` and test it with ``_lib.check_fault``."""
    return torch.zeros(1, dtype=torch.int32, device=device)


def gemm(a_hi, a_lo, w_hi, w_lo, bias=None, act=None, precision="bf16x3", residual=None, residual_planes=None,
         out_f32=False, out_planes=True, out_lo=True, small=None, fault=None):
    """C = act(A W^T + bias) + residual.  A planes [M,K], W planes [N,K].  -> (C_hi, C_lo, C_f32)."""
    M, K = a_hi.shape
    N = w_hi.shape[0]
    dev = a_hi.device
    g = _lib.GemmArgs()
    g.M, g.N, g.K, g.precision = M, N, K, _prec(precision)
    g.A_hi, g.A_lo, g.lda = a_hi.data_ptr(), _lib.ptr(a_lo), a_hi.stride(0)
    g.W_hi, g.W_lo, g.ldw = w_hi.data_ptr(), _lib.ptr(w_lo), w_hi.stride(0)
    g.bias, g.act = _lib.ptr(bias), ACTS[act]
    g.fault = _lib.ptr(fault)
    if residual is not None:
        g.R_f32, g.ldr = _f32c(residual).data_ptr(), residual.stride(0)
    if residual_planes is not None:
        g.R_hi, g.R_lo, g.ldr = residual_planes[0].data_ptr(), _lib.ptr(residual_planes[1]), residual_planes[0].stride(0)
    c_f32 = torch.empty((M, N), dtype=torch.float32, device=dev) if out_f32 else None
    pd = plane_dtype(precision)
    assert a_hi.dtype == pd and w_hi.dtype == pd, f"{precision} operands must be {pd} planes"
    c_hi = torch.empty((M, N), dtype=pd, device=dev) if out_planes else None
    c_lo = torch.empty((M, N), dtype=pd, device=dev) if out_planes and out_lo else None
    g.C_f32, g.C_hi, g.C_lo, g.ldc = _lib.ptr(c_f32), _lib.ptr(c_hi), _lib.ptr(c_lo), N
    if small is not None:  # split-K small-batch form; `small` = (gamma, beta, eps) for a LayerNorm over the columns, or ()
        ws_bytes = lib.didi_gemm_small_workspace_bytes(M, N, K, g.precision)
        if ws_bytes == 0:
            raise _lib.DidiHipError(f"the small-batch GEMM form does not apply to M={M} N={N} K={K}")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        gamma, beta, eps = small if small else (None, None, 0.0)
        _lib.check(lib.didi_gemm_small(C.byref(g), _lib.ptr(gamma), _lib.ptr(beta), eps, ws.data_ptr(), _lib.current_stream()), "gemm_small")
    else:
        _lib.check(lib.didi_gemm(C.byref(g), _lib.current_stream()), "gemm")
    return c_hi, c_lo, c_f32


def gemm_ln(a_hi, a_lo, w_hi, w_lo, bias, residual_planes, gamma, beta, eps, precision="fp16", f32=False, out=None, fault=None, lo_bytes=0):
    """LayerNorm(A W^T + bias + residual) in one launch -> (y_hi, y_lo, y_f32).  N == 768, M >= 2048.
    ``out`` = (y_hi, y_lo) to write into (may be the residual planes themselves, as the decoder does).
    ``lo_bytes`` (fp16 / fp16x2): bit 0: residual_planes[1] is a uint8 plane, bit 1: y_lo is written as one (didi_gemm_ln_lo8);
    residual_planes[1] None: the residual is its hi plane."""
    M, K = a_hi.shape
    N = w_hi.shape[0]
    dev = a_hi.device
    g = _lib.GemmArgs()
    g.M, g.N, g.K, g.precision = M, N, K, _prec(precision)
    g.A_hi, g.A_lo, g.lda = a_hi.data_ptr(), _lib.ptr(a_lo), a_hi.stride(0)
    g.W_hi, g.W_lo, g.ldw = w_hi.data_ptr(), _lib.ptr(w_lo), w_hi.stride(0)
    g.bias, g.act = bias.data_ptr(), _lib.ACT_NONE
    g.fault = _lib.ptr(fault)
    g.R_hi, g.R_lo, g.ldr = residual_planes[0].data_ptr(), _lib.ptr(residual_planes[1]), residual_planes[0].stride(0)
    pd = plane_dtype(precision)
    if lo_bytes & 1:
        assert residual_planes[1].dtype == torch.uint8 and residual_planes[1].stride(0) == g.ldr
    y_hi, y_lo = out if out is not None else (torch.empty((M, N), dtype=pd, device=dev),
                                              torch.empty((M, N), dtype=torch.uint8 if lo_bytes & 2 else pd, device=dev))
    if lo_bytes & 2:
        assert y_lo.dtype == torch.uint8
    y = torch.empty((M, N), dtype=torch.float32, device=dev) if f32 else None
    stats = torch.empty(lib.didi_gemm_ln_stats_bytes(M), dtype=torch.uint8, device=dev)
    _lib.check(lib.didi_gemm_ln_lo8(C.byref(g), gamma.data_ptr(), beta.data_ptr(), eps, y_hi.data_ptr(), _lib.ptr(y_lo), _lib.ptr(y),
                                    stats.data_ptr(), 0, lo_bytes, _lib.current_stream()), "gemm_ln")
    return y_hi, y_lo, y


def key_counts(mask32):
    """int32 [B, Sk] mask -> int32 [B]: visible keys per row when they form a prefix (right padding), else Sk."""
    B, Sk = mask32.shape
    counts = torch.empty(B, dtype=torch.int32, device=mask32.device)
    _lib.check(lib.didi_key_counts(mask32.data_ptr(), B, Sk, counts.data_ptr(), _lib.current_stream()), "key_counts")
    return counts


def attention(q, k, v, heads, causal=False, key_mask=None, precision="bf16x3", scale=None, pos_bias=None, key_count=None):
    """q/k/v: (hi, lo) plane pairs shaped [B, S, heads*D] (views with a row stride are fine).
    ``scale``: softmax scale (default 1/sqrt(D)); ``pos_bias``: fp32 [heads, Sq, Sk] added to the scaled scores (T5)."""
    B, Sq, HD = q[0].shape
    Sk = k[0].shape[1]
    a = _lib.AttnArgs()
    a.B, a.heads, a.Sq, a.Sk, a.D = B, heads, Sq, Sk, HD // heads
    a.precision, a.causal, a.key_mask = _prec(precision), int(causal), _lib.ptr(key_mask)
    a.key_count = _lib.ptr(key_count)
    a.Q_hi, a.Q_lo, a.ldq = q[0].data_ptr(), _lib.ptr(q[1]), q[0].stride(1)
    a.K_hi, a.K_lo, a.ldk = k[0].data_ptr(), _lib.ptr(k[1]), k[0].stride(1)
    a.V_hi, a.V_lo, a.ldv = v[0].data_ptr(), _lib.ptr(v[1]), v[0].stride(1)
    for t, s in ((q[0], Sq), (k[0], Sk), (v[0], Sk)):
        assert t.stride(0) == s * t.stride(1) and t.stride(2) == 1, "planes must be [B*S, ld] row-major"
    assert q[0].dtype == plane_dtype(precision), "operand planes must match the precision's format"
    o_hi = torch.empty((B, Sq, HD), dtype=q[0].dtype, device=q[0].device)
    o_lo = torch.empty_like(o_hi) if a.precision in (_lib.PREC_BF16X3, _lib.PREC_FP16X3) else None
    a.O_hi, a.O_lo, a.ldo = o_hi.data_ptr(), _lib.ptr(o_lo), HD
    a.scale = 0.0 if scale is None else float(scale)
    if pos_bias is not None:
        assert pos_bias.dtype == torch.float32 and pos_bias.is_contiguous() and tuple(pos_bias.shape) == (heads, Sq, Sk)
        a.pos_bias = pos_bias.data_ptr()
    _lib.check(lib.didi_attention(C.byref(a), _lib.current_stream()), "attention")
    return o_hi, o_lo


def rmsnorm(x, gamma, eps, planes=True, need_lo=True, f32=False, dtype=torch.bfloat16):
    """T5LayerNorm: x * rsqrt(mean(x^2) + eps) * gamma -> (hi, lo, f32) like ``layernorm``."""
    x = _f32c(x)
    M, H = x.numel() // x.shape[-1], x.shape[-1]
    hi = torch.empty_like(x, dtype=dtype) if planes else None
    lo = torch.empty_like(x, dtype=dtype) if planes and need_lo else None
    y = torch.empty_like(x) if f32 else None
    _lib.check(lib.didi_rmsnorm(x.data_ptr(), gamma.data_ptr(), eps, M, H, _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(y), _fmt(dtype),
                                _lib.current_stream()), "rmsnorm")
    return hi, lo, y


def mul_split(a, b, need_lo=True, dtype=torch.bfloat16):
    """(a * b) -> split planes."""
    a, b = _f32c(a), _f32c(b)
    assert a.shape == b.shape
    hi = torch.empty_like(a, dtype=dtype)
    lo = torch.empty_like(hi) if need_lo else None
    _lib.check(lib.didi_mul_split(a.data_ptr(), b.data_ptr(), a.numel(), hi.data_ptr(), _lib.ptr(lo), _fmt(dtype),
                                  _lib.current_stream()), "mul_split")
    return hi, lo


def round_argmax(x_hi, x_lo, w_hi, w_lo, precision="bf16x3"):
    M, K = x_hi.shape
    V = w_hi.shape[0]
    partial = torch.empty(lib.didi_round_argmax_bytes(M, V), dtype=torch.uint8, device=x_hi.device)
    ids = torch.empty(M, dtype=torch.int64, device=x_hi.device)
    _lib.check(lib.didi_round_argmax(x_hi.data_ptr(), _lib.ptr(x_lo), x_hi.stride(0), w_hi.data_ptr(), _lib.ptr(w_lo),
                                     w_hi.stride(0), M, V, K, _prec(precision), partial.data_ptr(), ids.data_ptr(), None,
                                     _lib.current_stream()), "round_argmax")
    return ids


def round_ce(x_hi, x_lo, w_hi, w_lo, targets, precision="bf16x3"):
    """-> (ids [M] int64, nll [M] fp32): argmax and per-token cross-entropy terms of X W^T, logits never written."""
    M, K = x_hi.shape
    V = w_hi.shape[0]
    targets = targets.reshape(-1).to(device=x_hi.device, dtype=torch.int64).contiguous()
    assert targets.numel() == M
    partial = torch.empty(lib.didi_round_ce_bytes(M, V), dtype=torch.uint8, device=x_hi.device)
    ids = torch.empty(M, dtype=torch.int64, device=x_hi.device)
    nll = torch.empty(M, dtype=torch.float32, device=x_hi.device)
    _lib.check(lib.didi_round_ce(x_hi.data_ptr(), _lib.ptr(x_lo), x_hi.stride(0), w_hi.data_ptr(), _lib.ptr(w_lo), w_hi.stride(0),
                                 M, V, K, _prec(precision), targets.data_ptr(), partial.data_ptr(), ids.data_ptr(), nll.data_ptr(),
                                 _lib.current_stream()), "round_ce")
    return ids, nll


def first_eos(ids, eos_id):
    B, S = ids.shape
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    lengths = torch.empty(B, dtype=torch.int32, device=ids.device)
    _lib.check(lib.didi_first_eos(ids.data_ptr(), B, S, int(eos_id), lengths.data_ptr(), _lib.current_stream()), "first_eos")
    return lengths


def step_embed_ln(mode, M, S, H, *, X=None, X0=None, E=None, cx0=0.0, ce=0.0, ratio=0.0, cin=1.0, seed=0, offset=0,
                  x_out=None, xin_out=None, time_embeds=None, t=0, time_idx=None, pos_emb=None, tok_type0=None,
                  ln_gamma=None, ln_beta=None, ln_eps=1e-12, h_hi=None, h_lo=None):
    a = _lib.StepArgs()
    a.mode, a.M, a.S, a.H = mode, M, S, H
    a.X, a.X0, a.E = _lib.ptr(X), _lib.ptr(X0), _lib.ptr(E)
    a.cx0, a.ce, a.ratio, a.cin = cx0, ce, ratio, cin
    a.seed, a.offset = seed, offset
    a.x_out, a.xin_out = _lib.ptr(x_out), _lib.ptr(xin_out)
    a.time_embeds, a.t, a.time_idx = _lib.ptr(time_embeds), t, _lib.ptr(time_idx)
    a.pos_emb, a.tok_type0, a.ln_gamma, a.ln_beta, a.ln_eps = (_lib.ptr(pos_emb), _lib.ptr(tok_type0), _lib.ptr(ln_gamma),
                                                               _lib.ptr(ln_beta), ln_eps)
    a.h_hi, a.h_lo = _lib.ptr(h_hi), _lib.ptr(h_lo)
    a.plane_format = _fmt(h_hi.dtype) if h_hi is not None else _lib.FMT_BF16
    _lib.check(lib.didi_step_embed_ln(C.byref(a), _lib.current_stream()), "step_embed_ln")


# ------------------------------------------------------------------ training-step kernels (didi_amd/csrc/train.hip)
def train_noise(emb, s_b, c_b, std0, n0=None, n1=None, seed=0, offset=0):
    """emb fp32 [B,S,H]; s_b, c_b fp32 [B] -> (x_0, x_t) fp32 [B,S,H].  n0 / n1: injected draws or None (in-kernel Philox)."""
    emb = _f32c(emb)
    B, S, H = emb.shape
    x0, xt = torch.empty_like(emb), torch.empty_like(emb)
    _lib.check(lib.didi_train_noise(emb.data_ptr(), _lib.ptr(n0), _lib.ptr(n1), float(std0), _f32c(s_b).data_ptr(), _f32c(c_b).data_ptr(),
                                    B, S, H, seed, offset, x0.data_ptr(), xt.data_ptr(), _lib.current_stream()), "train_noise")
    return x0, xt


def train_loss_terms(x0hat, emb, x0, mask):
    """-> fp32 [B, 3]: per dialog sums of m (x0hat-emb)^2, m (x0hat-x0)^2, m x0^2 (m = int64 mask [B,S])."""
    B, S, H = x0hat.shape
    out = torch.empty((B, 3), dtype=torch.float32, device=x0hat.device)
    mask = mask.to(torch.int64).contiguous()
    _lib.check(lib.didi_train_loss_terms(_f32c(x0hat).data_ptr(), _f32c(emb).data_ptr(), _f32c(x0).data_ptr(), mask.data_ptr(), B, S, H,
                                         out.data_ptr(), _lib.current_stream()), "train_loss_terms")
    return out


def train_mse_backward(x0hat, emb, x0, mask, w_emb, w_x0):
    B, S, H = x0hat.shape
    dx = torch.empty_like(x0hat)
    mask = mask.to(torch.int64).contiguous()
    _lib.check(lib.didi_train_mse_backward(_f32c(x0hat).data_ptr(), _f32c(emb).data_ptr(), _f32c(x0).data_ptr(), mask.data_ptr(),
                                           float(w_emb), float(w_x0), B * S, H, dx.data_ptr(), _lib.current_stream()), "train_mse_backward")
    return dx


def layernorm_backward(dy, x, gamma, eps):
    """-> (dx [M,H], dgamma [H], dbeta [H]) of torch.nn.LayerNorm over the last dim, from dy, the layer INPUT x and gamma."""
    dy, x = _f32c(dy), _f32c(x)
    H = x.shape[-1]
    M = x.numel() // H
    dx = torch.empty_like(x)
    dg, db = torch.empty(H, dtype=torch.float32, device=x.device), torch.empty(H, dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.didi_layernorm_backward_bytes(M, H), dtype=torch.uint8, device=x.device)
    _lib.check(lib.didi_layernorm_backward(dy.data_ptr(), x.data_ptr(), _f32c(gamma).data_ptr(), float(eps), M, H, dx.data_ptr(),
                                           dg.data_ptr(), db.data_ptr(), ws.data_ptr(), _lib.current_stream()), "layernorm_backward")
    return dx, dg, db


def colsum(x):
    x = _f32c(x)
    M, N = x.shape
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.didi_colsum_bytes(M, N), dtype=torch.uint8, device=x.device)
    _lib.check(lib.didi_colsum(x.data_ptr(), M, N, out.data_ptr(), ws.data_ptr(), _lib.current_stream()), "colsum")
    return out


def transpose_split(x, dtype=torch.bfloat16):
    """fp32 [R, C] -> planes (hi, lo) of x^T, shaped [C, ldo] with ldo = R rounded up to 8 (zero padded): a GEMM operand."""
    x = _f32c(x)
    R, Cc = x.shape
    ldo = (R + 7) // 8 * 8
    hi = torch.empty((Cc, ldo), dtype=dtype, device=x.device)
    lo = torch.empty_like(hi)
    _lib.check(lib.didi_transpose_split(x.data_ptr(), R, Cc, ldo, hi.data_ptr(), lo.data_ptr(), _fmt(dtype), _lib.current_stream()),
               "transpose_split")
    return hi, lo


def transpose_split_colsum(x, dtype=torch.bfloat16):
    """``transpose_split`` + the column sums of x from the same pass -> (hi, lo, colsum fp32 [C])."""
    x = _f32c(x)
    R, Cc = x.shape
    ldo = (R + 7) // 8 * 8
    hi = torch.empty((Cc, ldo), dtype=dtype, device=x.device)
    lo = torch.empty_like(hi)
    out = torch.empty(Cc, dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.didi_transpose_split_colsum_bytes(R, Cc), dtype=torch.uint8, device=x.device)
    _lib.check(lib.didi_transpose_split_colsum(x.data_ptr(), R, Cc, ldo, hi.data_ptr(), lo.data_ptr(), _fmt(dtype), out.data_ptr(),
                                               ws.data_ptr(), _lib.current_stream()), "transpose_split_colsum")
    return hi, lo, out


def split_both(x, dtype=torch.bfloat16, colsum=True, transposed=True, drop=None, gelu_u=None):
    """fp32 [R, C] -> (planes of x [R, C], planes of x^T [C, R8] (None, None without ``transposed``), column sums or None) in one
    pass (C % 4 == 0).  ``drop`` = (p, seed, offset): everything is taken of dropout(x) under ``dropout_add``'s mask; ``gelu_u``
    fp32 [R, C]: of x * gelu'(gelu_u) (``act_backward("gelu", x, gelu_u)``)."""
    x = _f32c(x)
    R, Cc = x.shape
    ldo = (R + 7) // 8 * 8
    p_hi = torch.empty((R, Cc), dtype=dtype, device=x.device)
    p_lo = torch.empty_like(p_hi)
    t_hi = torch.empty((Cc, ldo), dtype=dtype, device=x.device) if transposed else None
    t_lo = torch.empty_like(t_hi) if transposed else None
    out = torch.empty(Cc, dtype=torch.float32, device=x.device) if colsum else None
    ws = torch.empty(lib.didi_transpose_split_colsum_bytes(R, Cc), dtype=torch.uint8, device=x.device) if colsum else None
    if (drop is not None and drop[0] > 0) or gelu_u is not None:
        drop = drop if drop is not None else (0.0, 0, 0)
        if gelu_u is not None:
            gelu_u = _f32c(gelu_u)
            assert gelu_u.shape == x.shape
        _lib.check(lib.didi_split_both_ex(x.data_ptr(), R, Cc, ldo, _lib.ptr(t_hi), _lib.ptr(t_lo), p_hi.data_ptr(), p_lo.data_ptr(), _fmt(dtype),
                                          _lib.ptr(out), _lib.ptr(ws), _lib.ptr(gelu_u), float(drop[0]), int(drop[1]), int(drop[2]),
                                          _lib.current_stream()), "split_both_ex")
    else:
        _lib.check(lib.didi_split_both(x.data_ptr(), R, Cc, ldo, _lib.ptr(t_hi), _lib.ptr(t_lo), p_hi.data_ptr(), p_lo.data_ptr(), _fmt(dtype),
                                       _lib.ptr(out), _lib.ptr(ws), _lib.current_stream()), "split_both")
    return (p_hi, p_lo), (t_hi, t_lo), out


def wgrad_supported(M, N, K, precision):
    """Does didi_gemm_wgrad (dW = dC^T A from un-transposed planes) take dC [M, N], A [M, K] in this precision?"""
    return lib.didi_gemm_wgrad_workspace_bytes(M, N, K, _prec(precision)) > 0


def gemm_wgrad(dc_hi, dc_lo, a_hi, a_lo, precision="bf16x3"):
    """dW [N, K] fp32 = dC^T A from the row-major planes of dC [M, N] and A [M, K] (``wgrad_supported`` shapes only)."""
    M, N = dc_hi.shape
    K = a_hi.shape[1]
    assert a_hi.shape[0] == M and dc_hi.stride(1) == 1 and a_hi.stride(1) == 1
    nbytes = lib.didi_gemm_wgrad_workspace_bytes(M, N, K, _prec(precision))
    if not nbytes:
        raise ValueError(f"didi_gemm_wgrad does not take M={M} N={N} K={K} in {precision}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dc_hi.device)
    dW = torch.empty((N, K), dtype=torch.float32, device=dc_hi.device)
    _lib.check(lib.didi_gemm_wgrad(dc_hi.data_ptr(), _lib.ptr(dc_lo), dc_hi.stride(0), a_hi.data_ptr(), _lib.ptr(a_lo), a_hi.stride(0), M, N, K,
                                   _prec(precision), dW.data_ptr(), ws.data_ptr(), _lib.current_stream()), "gemm_wgrad")
    return dW


def dgrad_supported(M, N, K, precision):
    """Does didi_gemm_dgrad (dA = dC W from W's own planes) take dC [M, N], W [N, K] in this precision?"""
    return lib.didi_gemm_dgrad_supported(M, N, K, _prec(precision)) != 0


def gemm_dgrad(dc_hi, dc_lo, w_hi, w_lo, precision="bf16x3", residual=None):
    """dA [M, K] fp32 = dC W (+ residual fp32 [M, K]) from the planes of dC [M, N] and of W [N, K] as the forward holds them."""
    M, N = dc_hi.shape
    K = w_hi.shape[1]
    assert w_hi.shape[0] == N and dc_hi.stride(1) == 1 and w_hi.stride(1) == 1
    if residual is not None:
        residual = _f32c(residual)
        assert residual.shape == (M, K)
    dA = torch.empty((M, K), dtype=torch.float32, device=dc_hi.device)
    _lib.check(lib.didi_gemm_dgrad(dc_hi.data_ptr(), _lib.ptr(dc_lo), dc_hi.stride(0), w_hi.data_ptr(), _lib.ptr(w_lo), w_hi.stride(0), M, N, K,
                                   _prec(precision), _lib.ptr(residual), K, dA.data_ptr(), K, _lib.current_stream()), "gemm_dgrad")
    return dA


def linear_backward(dC, A, W, precision="bf16x3", bias=True, need_dA=True, residual=None, fault=None, A_planes=None, W_planes=None, dC_drop=None, dC_gelu_u=None):
    """Backward of C = A W^T + b (nn.Linear): dC [M,N], A [M,K], W [N,K] fp32 -> (dA [M,K] (+ residual), dW [N,K], db [N] or None).
    dgrad is a didi_gemm launch on the transposed weight planes; wgrad is didi_gemm_wgrad on the planes of dC and of A as they are
    (``A_planes``: the (hi, lo) the forward multiplied with -- passed, nothing of A is re-read in fp32; A may then be None) where that
    form applies, else a didi_gemm launch on transposed operand planes; with ``W_planes`` (the forward's weight planes) dgrad is
    didi_gemm_dgrad on them where that form applies (no transposed copy of W); db comes out of the pass that splits dC.
    ``residual`` fp32 [M,K]: added to dA in the dgrad GEMM's epilogue (the gradient of a skip connection joining there).
    ``dC_drop`` = (p, seed, offset): the Linear's output went through hidden dropout; dC is the gradient BEHIND it, and the mask is
    applied in the pass that splits dC (or by ``dropout_add`` where that pass is not the one-read form).  ``dC_gelu_u`` fp32 [M,N]:
    the Linear's output went through GELU with this pre-activation; likewise (or ``act_backward``)."""
    pd = plane_dtype(precision)
    dC, W = _f32c(dC), _f32c(W)
    M, N = dC.shape
    K = W.shape[1]
    tn = wgrad_supported(M, N, K, precision) and (A_planes is not None or A is not None)
    if dC_gelu_u is not None and not tn:
        dC, dC_gelu_u = act_backward("gelu", dC, dC_gelu_u), None
    if dC_drop is not None and dC_drop[0] > 0 and not tn:
        dC, dC_drop = dropout_add(dC, None, *dC_drop), None
    one_pass = N % 8 == 0 and (need_dA or tn)  # dC's planes, its transposed planes and its column sums from one read of dC
    dct = None
    if tn:
        (dc_hi, dc_lo), _, db1 = split_both(dC, dtype=pd, colsum=bias, transposed=False, drop=dC_drop, gelu_u=dC_gelu_u)
        a_hi, a_lo = A_planes if A_planes is not None else split(_f32c(A), dtype=pd)
    elif one_pass:
        (dc_hi, dc_lo), dct, db1 = split_both(dC, dtype=pd, colsum=bias)
    else:
        dc_hi, dc_lo = split(dC, dtype=pd)
    dA = None
    if need_dA and W_planes is not None and dgrad_supported(M, N, K, precision):
        dA = gemm_dgrad(dc_hi, dc_lo, W_planes[0], W_planes[1], precision=precision, residual=residual)
    elif need_dA:
        wt_hi, wt_lo = transpose_split(W, dtype=pd)          # [K, N8]
        if wt_hi.shape[1] != N:  # K of this GEMM is N: the A operand must carry the same zero padding
            pad = wt_hi.shape[1]
            z = torch.zeros((M, pad), dtype=torch.float32, device=dC.device)
            z[:, :N] = dC
            dc_hi, dc_lo = split(z, dtype=pd)
        Kd = dc_hi.shape[1]  # the dgrad's contraction length (N, padded to 8)
        K0 = Kd // 64 * 64
        if Kd % 32 != 0 and K0 >= 1024:
            # a long contraction that is not a multiple of the LDS-DMA kernels' K-step (the classifier: V = 8008) would fall to the
            # register-staged kernel whole (1.1 ms at M = 16384); the aligned part goes through the fast kernel, the tail of < 64 columns
            # adds on through the residual input
            head = gemm(dc_hi[:, :K0], dc_lo[:, :K0], wt_hi[:, :K0], wt_lo[:, :K0], precision=precision, out_f32=True, out_planes=False,
                        residual=residual, fault=fault)[2]
            dA = gemm(dc_hi[:, K0:], dc_lo[:, K0:], wt_hi[:, K0:], wt_lo[:, K0:], precision=precision, out_f32=True, out_planes=False,
                      residual=head, fault=fault)[2]
        else:
            dA = gemm(dc_hi, dc_lo, wt_hi, wt_lo, precision=precision, out_f32=True, out_planes=False, residual=residual, fault=fault)[2]
    if tn:
        return dA, gemm_wgrad(dc_hi, dc_lo, a_hi, a_lo, precision=precision), (db1 if bias else None)
    A = _f32c(A)
    if dct is not None:
        (dct_hi, dct_lo), db = dct, (db1 if bias else None)
    elif bias:
        dct_hi, dct_lo, db = transpose_split_colsum(dC, dtype=pd)  # [N, M8] + the bias gradient
    else:
        (dct_hi, dct_lo), db = transpose_split(dC, dtype=pd), None
    at_hi, at_lo = transpose_split(A, dtype=pd)          # [K, M8]
    # wgrad: few output tiles, K = the batch's rows -> the split-K form wherever it applies
    use_small = lib.didi_gemm_small_workspace_bytes(N, K, dct_hi.shape[1], _prec(precision)) > 0
    _, _, dW = gemm(dct_hi, dct_lo, at_hi, at_lo, precision=precision, out_f32=True, out_planes=False, small=() if use_small else None, fault=fault)
    return dA, dW, db


def act_backward(act, dy, saved):
    """dx = dy * act'(.): act "gelu" takes the PRE-activation, "tanh" the OUTPUT."""
    dy, saved = _f32c(dy), _f32c(saved)
    assert dy.shape == saved.shape
    dx = torch.empty_like(dy)
    _lib.check(lib.didi_act_backward(ACTS[act], dy.data_ptr(), saved.data_ptr(), dy.numel(), dx.data_ptr(), _lib.current_stream()),
               "act_backward")
    return dx


def axpby(a, x, b, y):
    """y <- a x + b y in place (fp32, contiguous, same shape)."""
    assert x.shape == y.shape and x.is_contiguous() and y.is_contiguous() and x.dtype == y.dtype == torch.float32
    _lib.check(lib.didi_axpby(float(a), x.data_ptr(), float(b), y.data_ptr(), x.numel(), _lib.current_stream()), "axpby")
    return y


def ce_backward(logits, targets, mask, scale):
    """In place: logits [M, V] fp32 <- scale (softmax - onehot) on rows with mask != 0, zero elsewhere.
    -> (the same tensor, nll fp32 [M]: the per-token cross-entropy terms, 0 on masked rows)."""
    M, V = logits.shape
    assert logits.is_contiguous() and logits.dtype == torch.float32
    targets = targets.reshape(-1).to(torch.int64).contiguous()
    mask = mask.reshape(-1).to(torch.int64).contiguous()
    nll = torch.empty(M, dtype=torch.float32, device=logits.device)
    _lib.check(lib.didi_ce_backward(logits.data_ptr(), targets.data_ptr(), mask.data_ptr(), M, V, float(scale), nll.data_ptr(),
                                    _lib.current_stream()), "ce_backward")
    return logits, nll


def act_forward(act, x, dtype=torch.bfloat16, f32=True):
    """act(x) -> (hi, lo, y fp32): "gelu" (exact erf form) or "tanh"."""
    x = _f32c(x)
    hi, lo = torch.empty_like(x, dtype=dtype), torch.empty_like(x, dtype=dtype)
    y = torch.empty_like(x) if f32 else None
    _lib.check(lib.didi_act_forward(ACTS[act], x.data_ptr(), x.numel(), _lib.ptr(y), hi.data_ptr(), lo.data_ptr(), _fmt(dtype),
                                    _lib.current_stream()), "act_forward")
    return hi, lo, y


def embedding_backward(ids, dy, grad, scale=1.0, skip_id=-1):
    """grad[ids[m], :] += scale * dy[m, :] (fp32 atomics); rows with id == skip_id, or outside the table, get nothing."""
    ids = ids.reshape(-1).to(torch.int64).contiguous()
    dy = _f32c(dy)
    H = dy.shape[-1]
    assert grad.is_contiguous() and grad.dtype == torch.float32 and grad.shape[-1] == H and dy.numel() == ids.numel() * H
    _lib.check(lib.didi_embedding_backward(ids.data_ptr(), dy.data_ptr(), ids.numel(), H, grad.numel() // H, float(scale), int(skip_id),
                                           grad.data_ptr(), _lib.current_stream()), "embedding_backward")
    return grad


def _attn_train_args(q, k, v, heads, causal, key_mask, scale, drop_p, seed, offset):
    B, Sq, HD = q.shape
    Sk = k.shape[1]
    for t_, s_ in ((q, Sq), (k, Sk), (v, Sk)):
        assert t_.dtype == torch.float32 and t_.stride(2) == 1 and t_.stride(0) == s_ * t_.stride(1), "operands must be [B*S, ld] row-major"
    a = _lib.AttnTrainArgs()
    a.Q, a.K, a.V, a.key_mask = q.data_ptr(), k.data_ptr(), v.data_ptr(), _lib.ptr(key_mask)
    a.B, a.heads, a.Sq, a.Sk, a.D = B, heads, Sq, Sk, HD // heads
    a.ldq, a.ldk, a.ldv, a.ldo, a.causal = q.stride(1), k.stride(1), v.stride(1), HD, int(causal)
    a.scale, a.drop_p, a.seed, a.offset = (0.0 if scale is None else float(scale)), float(drop_p), int(seed), int(offset)
    return a


def attention_train(q, k, v, heads, causal=False, key_mask=None, scale=None, drop_p=0.0, seed=0, offset=0, pos_bias=None):
    """Training-step attention in fp32: O = dropout(softmax(q k^T scale + pos_bias + masks)) v -> fp32 [B, Sq, heads*D] contiguous.
    q/k/v fp32 [B, S, heads*D] (row-strided views of a fused projection are fine); ``pos_bias`` fp32 [heads, Sq, Sk] (T5)."""
    a = _attn_train_args(q, k, v, heads, causal, key_mask, scale, drop_p, seed, offset)
    if pos_bias is not None:
        assert pos_bias.dtype == torch.float32 and pos_bias.is_contiguous() and tuple(pos_bias.shape) == (heads, a.Sq, a.Sk)
        a.pos_bias = pos_bias.data_ptr()
    out = torch.empty((a.B, a.Sq, a.heads * a.D), dtype=torch.float32, device=q.device)
    a.O = out.data_ptr()
    _lib.check(lib.didi_attention_train_forward(C.byref(a), _lib.current_stream()), "attention_train_forward")
    return out


def attention_backward(q, k, v, dO, heads, causal=False, key_mask=None, scale=None, out=None, drop_p=0.0, seed=0, offset=0):
    """Backward of ``attention_train`` (same drop_p / seed / offset as the forward): dO [B,Sq,heads*D] contiguous -> (dq, dk, dv).
    ``out``: views to write into, strided exactly like q, k, v (e.g. the thirds of one [B,S,3H] buffer, which is then the
    gradient of the fused projection's output); default: fresh contiguous tensors (needs contiguous q, k, v)."""
    dO = _f32c(dO)
    a = _attn_train_args(q, k, v, heads, causal, key_mask, scale, drop_p, seed, offset)
    if out is None:
        out = tuple(torch.empty(t_.shape, dtype=torch.float32, device=q.device) for t_ in (q, k, v))
    for t_, g_ in zip((q, k, v), out):
        assert g_.dtype == torch.float32 and g_.stride() == t_.stride() and g_.shape == t_.shape, "gradients are laid out like their operands"
    a.dO, a.dQ, a.dK, a.dV = dO.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr()
    _lib.check(lib.didi_attention_train_backward(C.byref(a), _lib.current_stream()), "attention_train_backward")
    return out


def dropout_add(x, residual, p, seed, offset):
    """x o mask / (1 - p) + residual (or None) -> fp32, Philox mask keyed (seed, offset); the backward is the same call on dy."""
    x = _f32c(x)
    r = None if residual is None else _f32c(residual)
    out = torch.empty_like(x)
    _lib.check(lib.didi_dropout_add(x.data_ptr(), _lib.ptr(r), x.numel(), float(p), int(seed), int(offset), out.data_ptr(),
                                    _lib.current_stream()), "dropout_add")
    return out


def dropout_add_layernorm(x, residual, p, seed, offset, gamma, beta, eps, dtype=torch.bfloat16, f32=True):
    """s = dropout_add(x, residual, ...) and layernorm(s) in one pass -> (hi, lo, y fp32 or None, s)."""
    x, residual = _f32c(x), _f32c(residual)
    assert x.shape == residual.shape
    M, H = x.numel() // x.shape[-1], x.shape[-1]
    s = torch.empty_like(x)
    hi, lo = torch.empty_like(x, dtype=dtype), torch.empty_like(x, dtype=dtype)
    y = torch.empty_like(x) if f32 else None
    _lib.check(lib.didi_dropout_add_layernorm(x.data_ptr(), residual.data_ptr(), float(p), int(seed), int(offset), s.data_ptr(), gamma.data_ptr(),
                                              beta.data_ptr(), eps, M, H, hi.data_ptr(), lo.data_ptr(), _lib.ptr(y), _fmt(dtype),
                                              _lib.current_stream()), "dropout_add_layernorm")
    return hi, lo, y, s


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, gscale=None):
    """torch.optim.AdamW on one fp32 tensor, in place; gradient pre-scaled by the device scalar ``gscale`` (clipping)."""
    for t_ in (p, g, m, v):
        assert t_.dtype == torch.float32 and t_.is_contiguous() and t_.numel() == p.numel()
    _lib.check(lib.didi_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), float(beta1), float(beta2),
                                   float(eps), float(weight_decay), int(step), _lib.ptr(gscale), _lib.current_stream()), "adamw_step")


def sumsq_accumulate(x, out):
    """out[0] += sum x^2."""
    assert x.dtype == torch.float32 and x.is_contiguous() and out.dtype == torch.float32
    _lib.check(lib.didi_sumsq_accumulate(x.data_ptr(), x.numel(), out.data_ptr(), _lib.current_stream()), "sumsq")


def train_embed_sum(x_t, time_embeds, t, pos_emb, tok0):
    """x_t + time_embeds[t_b] + pos[s] + tok0 -> fp32 [B,S,H] (the decoder-embeddings LayerNorm input)."""
    x_t = _f32c(x_t)
    B, S, H = x_t.shape
    t = t.reshape(-1).to(torch.int64).contiguous()
    assert t.numel() == B and pos_emb.shape[0] >= S
    out = torch.empty_like(x_t)
    assert time_embeds.is_contiguous() and time_embeds.shape[-1] == H
    _lib.check(lib.didi_train_embed_sum(x_t.data_ptr(), time_embeds.data_ptr(), t.data_ptr(), pos_emb.data_ptr(), tok0.data_ptr(), B, S, H,
                                        time_embeds.shape[0], out.data_ptr(), _lib.current_stream()), "train_embed_sum")
    return out


def seq_sum(x):
    x = _f32c(x)
    B, S, H = x.shape
    out = torch.empty((B, H), dtype=torch.float32, device=x.device)
    _lib.check(lib.didi_seq_sum(x.data_ptr(), B, S, H, out.data_ptr(), _lib.current_stream()), "seq_sum")
    return out


def train_target_backward(x0hat, emb, x0, mask, w_emb, w_x0, w_T, c_b, dxt, extra=None):
    B, S, H = x0hat.shape
    out = torch.empty_like(x0hat)
    mask = mask.to(torch.int64).contiguous()
    _lib.check(lib.didi_train_target_backward(_f32c(x0hat).data_ptr(), _f32c(emb).data_ptr(), _f32c(x0).data_ptr(), mask.data_ptr(),
                                              float(w_emb), float(w_x0), float(w_T), _f32c(c_b).data_ptr(), _f32c(dxt).data_ptr(),
                                              _lib.ptr(None if extra is None else _f32c(extra)), B, S, H, out.data_ptr(),
                                              _lib.current_stream()), "train_target_backward")
    return out
