"""GPU parity of the training step against the REFERENCE-generated fixtures, through the C ABI.

tests/golden/train_tiny_*.safetensors hold DiDi.training_step (src/diffusion/model.py:200-243) run on the HF modules with a
tapped generator (oracle/make_golden.py train): the draws, t (one dialog at t == 1, so both branches of the mse `where` are
live), x_t, x_0_hat, the five logged scalars, every parameter gradient and the intermediates of the last decoder layer's
feed-forward block.  Bars: forward tensors rel-L2 1e-5 (bf16x3; 1e-3 fp16x2), scalars 1e-4 relative, gradients from the
backward kernels 1e-4 relative (bf16x3 GEMMs).
"""
import pytest
import torch

from golden_util import load_case, noise_list, spec_to_config, weights_of

pytestmark = pytest.mark.gpu
CASES = ["train_tiny_plain", "train_tiny_anchor_nf2"]


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _setup(name, precision="bf16x3"):
    from didi_amd.model import DiDiSampler
    from didi_amd.synthetic import RawContext

    t, meta = load_case(name)
    w = weights_of(load_case("tiny_weights")[0])
    model = DiDiSampler(w, spec_to_config(meta["spec"]), device="cuda", precision=precision)
    ctx = RawContext(t["in.input_ids"].cuda(), t["in.attention_mask"].cuda())
    tgt = RawContext(t["in.target_ids"].cuda(), t["in.target_mask"].cuda())
    return t, meta, w, model, ctx, tgt


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("precision", ["bf16x3", "fp16x2"])
def test_training_forward_and_losses(name, precision):
    from didi_amd.training import metric_message, training_step_forward

    t, meta, w, model, ctx, tgt = _setup(name, precision)
    run = meta["run"]
    out = training_step_forward(model, ctx, tgt, anchor_loss=run["anchor_loss"], noise_factor=run["noise_factor"],
                                draws=noise_list(t, int(meta["n_noise"])), t=t["t"])
    model.check_fault(name)
    tol = 1e-5 if precision == "bf16x3" else 1e-3
    assert rel_l2(out["emb"], t["out.emb"]) == 0.0
    assert rel_l2(out["x_0"], t["out.x_0"]) <= 1e-6 and rel_l2(out["x_t"], t["out.x_t"]) <= 1e-6
    assert rel_l2(out["x_0_hat"], t["out.x_0_hat"]) <= tol * 3
    got, want = metric_message(out).cpu(), t["out.scalars"]
    print(f"\n[{name} {precision}] mse, ce, t_T, loss, ce_hat = {got.tolist()}  (reference {want.tolist()})  x_0_hat rel-L2 {rel_l2(out['x_0_hat'], t['out.x_0_hat']):.2e}")
    assert torch.allclose(got, want, rtol=1e-4 if precision == "bf16x3" else 2e-3, atol=1e-9)


def test_training_noise_philox_is_reproducible_and_normal():
    """Without injected draws the noise comes from the in-kernel Philox stream keyed by torch's global generator."""
    from didi_amd.training import training_step_forward

    t, meta, w, model, ctx, tgt = _setup("train_tiny_plain")
    torch.manual_seed(7)
    a = training_step_forward(model, ctx, tgt)
    torch.manual_seed(7)
    b = training_step_forward(model, ctx, tgt)
    assert torch.equal(a["x_t"], b["x_t"]) and torch.equal(a["t"], b["t"]) and float(a["loss"]) == float(b["loss"])
    z = (a["x_0"] - a["emb"]) / model.std_0
    assert abs(float(z.mean())) < 0.05 and abs(float(z.std()) - 1.0) < 0.05
    assert int(a["t"].min()) >= 1 and int(a["t"].max()) <= model.diffusion_steps


@pytest.mark.parametrize("name", CASES)
def test_loss_gradient_wrt_x0_hat(name):
    """d loss / d x_0_hat.  Without the anchor loss it is the mse's gradient alone (the cross-entropy reads x_0, not x_0_hat); with it
    the classifier's cross-entropy adds its share, which this round's kernels do not produce yet: there the mse part is checked
    against autograd minus the CE part computed in fp64 from the fixture's own tensors."""
    from didi_amd.training import mse_backward

    t, meta, w, model, ctx, tgt = _setup(name)
    dx = mse_backward(t["out.x_0_hat"].cuda(), t["out.emb"].cuda(), t["out.x_0"].cuda(), t["in.target_mask"].cuda(), t["t"].cuda())
    want = t["grad.x_0_hat"].double()
    if meta["run"]["anchor_loss"]:
        x = t["out.x_0_hat"].double().requires_grad_(True)
        keep = t["in.target_mask"].bool().view(-1)
        logits = (x @ w["classifier.weight"].double().T).view(-1, w["classifier.weight"].shape[0])[keep]
        torch.nn.functional.cross_entropy(logits, t["in.target_ids"].view(-1)[keep]).backward()
        want = want - x.grad
    assert rel_l2(dx, want) <= 1e-5


@pytest.mark.parametrize("name", CASES)
def test_last_layer_ffn_backward_kernels_vs_autograd(name):
    """The last decoder layer's BertOutput (dense + residual + LayerNorm) backward, kernel by kernel, on the fixture's
    intermediates: LayerNorm backward (dx, dgamma, dbeta), then the Linear backward on the MFMA GEMM (dA, dW, db)."""
    from didi_amd import ops

    t, meta, w, model, ctx, tgt = _setup(name)
    L = meta["spec"]["dec_layers"] - 1
    p = f"decoder.encoder.layer.{L}.output."
    H = meta["spec"]["dec_dim"]
    dy = t["grad.x_0_hat"].cuda().view(-1, H)
    ln_in = t["last.ln_in"].cuda().view(-1, H)
    dx, dg, db = ops.layernorm_backward(dy, ln_in, w[p + "LayerNorm.weight"].cuda().float(), 1e-12)
    assert rel_l2(dx, t["last.ln_in.grad"].view(-1, H)) <= 1e-5
    assert rel_l2(dg, t["g." + p + "LayerNorm.weight"]) <= 1e-5 and rel_l2(db, t["g." + p + "LayerNorm.bias"]) <= 1e-5
    ffn = t["last.ffn"].cuda().view(dx.shape[0], -1)
    dA, dW, dbias = ops.linear_backward(dx, ffn, w[p + "dense.weight"].cuda().float())
    print(f"\n[{name}] LayerNorm bwd dx {rel_l2(dx, t['last.ln_in.grad'].view(-1, H)):.2e}; Linear bwd dA {rel_l2(dA, t['last.ffn.grad'].view(dA.shape)):.2e} "
          f"dW {rel_l2(dW, t['g.' + p + 'dense.weight']):.2e} db {rel_l2(dbias, t['g.' + p + 'dense.bias']):.2e}")
    assert rel_l2(dA, t["last.ffn.grad"].view(dA.shape)) <= 1e-4
    assert rel_l2(dW, t["g." + p + "dense.weight"]) <= 1e-4
    assert rel_l2(dbias, t["g." + p + "dense.bias"]) <= 1e-5


@pytest.mark.parametrize("M,H", [(40, 128), (1000, 768), (4099, 1280), (3, 96)])
def test_layernorm_backward_vs_autograd(M, H):
    from didi_amd import ops

    g = torch.Generator().manual_seed(M + H)
    x = (torch.randn(M, H, generator=g) * 3 + 1.5).double().requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(H, generator=g)).double().requires_grad_(True)
    beta = torch.randn(H, generator=g).double().requires_grad_(True)
    dy = torch.randn(M, H, generator=g).double()
    torch.nn.functional.layer_norm(x, (H,), gamma, beta, 1e-12).backward(dy)
    dx, dg, db = ops.layernorm_backward(dy.float().cuda(), x.detach().float().cuda(), gamma.detach().float().cuda(), 1e-12)
    assert rel_l2(dx, x.grad) <= 1e-5 and rel_l2(dg, gamma.grad) <= 1e-5 and rel_l2(db, beta.grad) <= 1e-5


@pytest.mark.parametrize("M,N,K", [(40, 128, 512), (2304, 768, 3072), (250, 600, 96), (64, 100, 64)])
def test_linear_backward_vs_fp64(M, N, K):
    """dgrad / wgrad / bias gradient on ragged and full-size shapes (M = 2304 takes the 256x192 tile in both GEMMs)."""
    from didi_amd import ops

    g = torch.Generator().manual_seed(M * 7 + N)
    dC, A, W = torch.randn(M, N, generator=g), torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    dA, dW, db = ops.linear_backward(dC.cuda(), A.cuda(), W.cuda())
    assert rel_l2(dA, dC.double() @ W.double()) <= 3e-5
    assert rel_l2(dW, dC.double().T @ A.double()) <= 3e-5
    assert rel_l2(db, dC.double().sum(0)) <= 1e-5


@pytest.mark.parametrize("R,Cc", [(130, 72), (1000, 768), (2050, 8008), (61, 4), (64, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_transpose_split_vector_form(R, Cc, dtype):
    """Column counts that are multiples of 4 take the kernel with 16-byte loads and 8-byte stores: same planes as the definition
    (hi = the plane format's rounding of x^T, hi + lo = x^T to the remainder plane's precision, zero padding up to a multiple of
    8 columns), and the fused column sums."""
    from didi_amd import ops

    x = torch.randn(R, Cc, generator=torch.Generator().manual_seed(R + Cc))
    hi, lo = ops.transpose_split(x.cuda(), dtype=dtype)
    ldo = (R + 7) // 8 * 8
    assert tuple(hi.shape) == (Cc, ldo)
    assert torch.equal(hi[:, :R].cpu(), x.T.to(dtype)) and float(ops.join(hi, lo)[:, R:].abs().max() if ldo > R else 0.0) == 0.0
    assert rel_l2(ops.join(hi, lo)[:, :R], x.T) <= (2e-5 if dtype == torch.bfloat16 else 1e-6)
    h2, l2, cs = ops.transpose_split_colsum(x.cuda(), dtype=dtype)
    assert torch.equal(h2, hi) and torch.equal(l2, lo) and rel_l2(cs, x.double().sum(0)) <= 1e-6


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_split_both_equals_the_three_passes(dtype):
    """didi_split_both: the planes of dC, of dC^T and the column sums from one read == split, transpose_split and colsum."""
    from didi_amd import ops

    x = torch.randn(1030, 776, generator=torch.Generator().manual_seed(3)).cuda()
    (p_hi, p_lo), (t_hi, t_lo), cs = ops.split_both(x, dtype=dtype)
    w_hi, w_lo = ops.split(x, dtype=dtype)
    u_hi, u_lo, want_cs = ops.transpose_split_colsum(x, dtype=dtype)
    assert torch.equal(p_hi, w_hi) and torch.equal(p_lo, w_lo) and torch.equal(t_hi, u_hi) and torch.equal(t_lo, u_lo) and torch.equal(cs, want_cs)
    assert ops.split_both(x, dtype=dtype, colsum=False)[2] is None
    # without the transposed planes (the weight gradient reads dC as it lies), and of dropout(x): the mask of dropout_add, applied on the read
    (q_hi, q_lo), (n_hi, n_lo), cs2 = ops.split_both(x, dtype=dtype, transposed=False)
    assert n_hi is None and n_lo is None and torch.equal(q_hi, w_hi) and torch.equal(q_lo, w_lo) and torch.equal(cs2, want_cs)
    seed, off = 1234567, 5 << 34
    xd = ops.dropout_add(x, None, 0.1, seed, off)
    assert 0.05 < float((xd == 0).float().mean()) < 0.15
    u = torch.randn(1030, 776, generator=torch.Generator().manual_seed(4)).cuda()
    (g_hi, g_lo), _, gcs = ops.split_both(x, dtype=dtype, transposed=False, gelu_u=u)  # of x * gelu'(u): act_backward on the read
    (h_hi, h_lo), _, hcs = ops.split_both(ops.act_backward("gelu", x, u), dtype=dtype, transposed=False)
    assert torch.equal(g_hi, h_hi) and torch.equal(g_lo, h_lo) and torch.equal(gcs, hcs)
    (d_hi, d_lo), (dt_hi, dt_lo), dcs = ops.split_both(x, dtype=dtype, drop=(0.1, seed, off))
    (e_hi, e_lo), (et_hi, et_lo), ecs = ops.split_both(xd, dtype=dtype)
    assert torch.equal(d_hi, e_hi) and torch.equal(d_lo, e_lo) and torch.equal(dt_hi, et_hi) and torch.equal(dt_lo, et_lo) and torch.equal(dcs, ecs)


@pytest.mark.parametrize("H", [768, 1280, 128])
def test_dropout_add_layernorm_equals_the_two_passes(H):
    """didi_dropout_add_layernorm == didi_dropout_add followed by didi_layernorm, bit for bit (sum kept for the backward included)."""
    from didi_amd import ops

    g = torch.Generator().manual_seed(11)
    x, r = torch.randn(1031, H, generator=g).cuda(), torch.randn(1031, H, generator=g).cuda()
    gamma, beta = torch.randn(H, generator=g).cuda(), torch.randn(H, generator=g).cuda()
    seed, off = 987654321, 7 << 34
    for dtype in (torch.bfloat16, torch.float16):
        hi, lo, y, s = ops.dropout_add_layernorm(x, r, 0.1, seed, off, gamma, beta, 1e-12, dtype=dtype)
        s_want = ops.dropout_add(x, r, 0.1, seed, off)
        hi_w, lo_w, y_w = ops.layernorm(s_want, gamma, beta, 1e-12, f32=True, dtype=dtype)
        assert torch.equal(s, s_want) and torch.equal(y, y_w) and torch.equal(hi, hi_w) and torch.equal(lo, lo_w)


def test_transpose_split_and_colsum():
    from didi_amd import ops

    x = torch.randn(130, 70, generator=torch.Generator().manual_seed(1))
    hi, lo = ops.transpose_split(x.cuda())
    assert tuple(hi.shape) == (70, 136)
    assert rel_l2(ops.join(hi, lo)[:, :130], x.T) <= 2e-5 and float(ops.join(hi, lo)[:, 130:].abs().max()) == 0.0
    assert torch.equal(hi[:, :130].cpu(), x.T.to(torch.bfloat16))
    assert rel_l2(ops.colsum(x.cuda()), x.double().sum(0)) <= 1e-6
