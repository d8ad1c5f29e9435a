"""didi_round_argmax_static (the entry point that allocates nothing) against didi_round_argmax, through the C ABI.

Bit-exact bar (integer ids).  Two calls in a row on the same GPU, the second with a smaller input: the static scratch keeps the
first call's partials, and stale slots must not leak into the second result.  Collected last (test_zz_*): it must pass alone in
a fresh process and after every other test.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _planes(M, V, K, seed, dtype):
    from didi_amd import ops

    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(V, K, generator=g) * 0.02).cuda()
    return ops.split(x, dtype=dtype), ops.split(w, dtype=dtype)


def _static(x, w, precision):
    from didi_amd import _lib, ops

    (x_hi, x_lo), (w_hi, w_lo) = x, w
    M, K = x_hi.shape
    V = w_hi.shape[0]
    ids = torch.full((M,), -1, dtype=torch.int64, device="cuda")
    lo = precision == "bf16x3"
    rc = _lib.lib.didi_round_argmax_static(x_hi.data_ptr(), x_lo.data_ptr() if lo else None, x_hi.stride(0), w_hi.data_ptr(),
                                           w_lo.data_ptr() if lo else None, w_hi.stride(0), M, V, K, ops.PRECISIONS[precision],
                                           ids.data_ptr(), None, _lib.current_stream())
    return rc, ids


@pytest.mark.parametrize("precision", ["fp16", "bf16x3"])
def test_round_argmax_static_matches_workspace_form_twice(precision):
    from didi_amd import _lib, ops

    dtype = ops.plane_dtype(precision)
    # first call: the 256x192 big-tile form (M >= 2048); second, smaller: the 128x128 chunked form with fewer rows AND fewer
    # vocabulary chunks, so most of the scratch still holds the first call's partials
    for M, V, K, seed in ((2304, 8008, 768, 1), (300, 1000, 768, 2)):
        x, w = _planes(M, V, K, seed, dtype)
        want = ops.round_argmax(x[0], x[1] if precision == "bf16x3" else None, w[0], w[1], precision)
        rc, got = _static(x, w, precision)
        assert rc == _lib.OK
        assert torch.equal(got, want), f"M={M} V={V}: {(got != want).sum().item()} ids differ"
        ref = (x[0].double() @ w[0].double().T).argmax(-1)  # and both are the argmax of the hi-plane product where it is clear-cut
        assert (got == ref).float().mean().item() >= 0.99


def test_round_argmax_static_rejects_what_does_not_fit():
    from didi_amd import _lib, ops

    limit = _lib.lib.didi_round_argmax_static_max_bytes()
    assert limit == 16 << 20
    M, V, K = 16384, 50000, 64  # 16384 * 391 * 8 B = 51 MB of partials
    assert _lib.lib.didi_round_argmax_bytes(M, V) > limit
    x_hi = torch.zeros((M, K), dtype=torch.float16, device="cuda")
    w_hi = torch.zeros((V, K), dtype=torch.float16, device="cuda")
    rc, _ = _static((x_hi, x_hi), (w_hi, w_hi), "fp16")
    assert rc == -4 and b"workspace" in _lib.lib.didi_error_string(rc)
    # the largest bench shape fits; at V = 50000 (391 chunks of 128) the boundary is M = floor(2097152 / 391) = 5363
    assert _lib.lib.didi_round_argmax_bytes(16384, 8008) <= limit
    assert _lib.lib.didi_round_argmax_bytes(5363, 50000) <= limit < _lib.lib.didi_round_argmax_bytes(5364, 50000)
