"""GPU parity of the FULL training step (forward + hand-written backward + AdamW) -- SURVEY.md section 8f.2 / BASELINE configs[4].

Ground truth for the gradients: tests/golden/train_tiny_*.safetensors, written by oracle/make_golden.py from the reference's
``DiDi.training_step`` run under torch autograd on the HF modules (all 63 trainable-parameter gradients).  Kernel-level
tests compare each backward kernel with torch autograd in fp64 on the CPU (a floating-point kernel: torch reference, tolerance
in the test).  The optimiser is checked against torch.optim.AdamW + LambdaLR + clip_grad_norm_ (what Lightning runs for
configs/train.yaml:33-45).
"""
import math

import pytest
import torch

from golden_util import load_case, noise_list, spec_to_config, weights_of

pytestmark = pytest.mark.gpu
CASES = ["train_tiny_plain", "train_tiny_anchor_nf2"]


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


# ------------------------------------------------------------------ kernels
@pytest.mark.parametrize("B,heads,Sq,Sk,D,causal,masked", [(3, 4, 10, 10, 32, True, False), (2, 12, 64, 64, 64, True, False),
                                                           (3, 4, 10, 12, 32, False, True), (2, 12, 64, 61, 64, False, True),
                                                           (1, 2, 96, 96, 64, False, False), (2, 3, 7, 5, 16, False, True),
                                                           (2, 3, 96, 64, 64, False, True), (3, 2, 32, 80, 16, False, True), (2, 2, 48, 48, 48, True, False)])
def test_attention_backward_vs_autograd(B, heads, Sq, Sk, D, causal, masked):
    """Operands as thirds / halves of fused projections (row stride 3H / 2H), like the trainer passes them.  Shapes whose Sq, Sk and D
    are all multiples of 16 take the fp32-MFMA form of the kernel, the others the FMA loops."""
    from didi_amd import ops

    g = torch.Generator().manual_seed(B * 100 + Sq)
    H = heads * D
    fused = Sq == Sk
    if fused:
        qkv = torch.randn(B, Sq, 3 * H, generator=g)
        q, k, v = qkv[:, :, :H], qkv[:, :, H:2 * H], qkv[:, :, 2 * H:]
    else:
        q, kv = torch.randn(B, Sq, H, generator=g), torch.randn(B, Sk, 2 * H, generator=g)
        k, v = kv[:, :, :H], kv[:, :, H:]
    dO = torch.randn(B, Sq, H, generator=g)
    mask = torch.ones(B, Sk, dtype=torch.int32)
    if masked:
        for b in range(B):
            mask[b, max(1, Sk - 1 - 2 * b):] = 0
    qd, kd, vd = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    sh = lambda t, S: t.view(B, S, heads, D).transpose(1, 2)  # noqa: E731
    s = sh(qd, Sq) @ sh(kd, Sk).transpose(-1, -2) / math.sqrt(D)
    neg = torch.finfo(torch.float32).min
    if masked:
        s = s + (1.0 - mask[:, None, None, :].double()) * neg
    if causal:
        s = s + torch.triu(torch.full((Sq, Sk), neg, dtype=torch.float64), 1)
    o = (torch.softmax(s, -1) @ sh(vd, Sk)).transpose(1, 2).reshape(B, Sq, H)
    o.backward(dO.double())
    if fused:
        dev = qkv.cuda()
        dq, dk, dv = dev[:, :, :H], dev[:, :, H:2 * H], dev[:, :, 2 * H:]
        out = torch.zeros_like(dev)
        outs = (out[:, :, :H], out[:, :, H:2 * H], out[:, :, 2 * H:])
    else:
        dq, kvd = q.cuda(), kv.cuda()
        dk, dv = kvd[:, :, :H], kvd[:, :, H:]
        o1, o2 = torch.zeros_like(dq), torch.zeros_like(kvd)
        outs = (o1, o2[:, :, :H], o2[:, :, H:])
    gq, gk, gv = ops.attention_backward(dq, dk, dv, dO.cuda(), heads, causal=causal, key_mask=mask.cuda() if masked else None, out=outs)
    assert rel_l2(gq, qd.grad) <= 2e-5 and rel_l2(gk, kd.grad) <= 2e-5 and rel_l2(gv, vd.grad) <= 2e-5


@pytest.mark.parametrize("act", ["gelu", "tanh"])
def test_activation_forward_backward(act):
    from didi_amd import ops

    x = (torch.randn(333, 512, generator=torch.Generator().manual_seed(2)) * 3).double().requires_grad_(True)
    dy = torch.randn(333, 512, generator=torch.Generator().manual_seed(3)).double()
    y = torch.nn.functional.gelu(x) if act == "gelu" else torch.tanh(x)
    y.backward(dy)
    for dtype in (torch.bfloat16, torch.float16):
        hi, lo, yf = ops.act_forward(act, x.detach().float().cuda(), dtype=dtype)
        assert rel_l2(yf, y) <= 1e-6 and rel_l2(ops.join(hi, lo), y) <= (2e-5 if dtype == torch.bfloat16 else 1e-6)
    saved = x.detach().float().cuda() if act == "gelu" else y.detach().float().cuda()
    assert rel_l2(ops.act_backward(act, dy.float().cuda(), saved), x.grad) <= 2e-6


@pytest.mark.parametrize("M,V", [(40, 600), (257, 8008), (16, 50000)])
def test_ce_backward_vs_autograd(M, V):
    from didi_amd import ops

    g = torch.Generator().manual_seed(V)
    lg = (torch.randn(M, V, generator=g) * 4).double().requires_grad_(True)
    tg = torch.randint(0, V, (M,), generator=g)
    mask = (torch.rand(M, generator=g) > 0.3).long()
    mask[0] = 1
    keep = mask.bool()
    loss = torch.nn.functional.cross_entropy(lg[keep], tg[keep])
    loss.backward()
    dev = lg.detach().float().cuda()
    got, nll = ops.ce_backward(dev, tg.cuda(), mask.cuda(), 1.0 / int(mask.sum()))
    assert got.data_ptr() == dev.data_ptr()
    assert rel_l2(got, lg.grad) <= 1e-5
    assert float(got[~keep.cuda()].abs().max() if (~keep).any() else 0.0) == 0.0
    assert abs(float(nll.sum()) / int(mask.sum()) - float(loss.detach())) <= 1e-5 * max(1.0, float(loss.detach()))


def test_embedding_backward_repeats_and_padding():
    from didi_amd import ops

    g = torch.Generator().manual_seed(4)
    ids = torch.randint(0, 50, (7, 33), generator=g)
    ids[:, -5:] = 0  # padding rows
    dy = torch.randn(7, 33, 128, generator=g)
    emb = torch.nn.Embedding(50, 128, padding_idx=0).double()
    emb(ids).backward(dy.double())
    grad = torch.zeros(50, 128, device="cuda")
    ops.embedding_backward(ids.cuda(), dy.cuda(), grad, skip_id=0)
    assert rel_l2(grad, emb.weight.grad) <= 1e-6 and float(grad[0].abs().max()) == 0.0
    ops.embedding_backward(ids.cuda(), dy.cuda(), grad, scale=-1.0, skip_id=0)  # accumulates: back to ~0
    assert float(grad.abs().max()) <= 1e-5


def test_out_of_range_ids_and_timesteps_touch_nothing():
    """An id outside the table (vocabulary mismatch, ignore_index = -100) or a timestep outside time_embeds must not become an
    out-of-bounds device access: embedding_backward skips the row, ce_backward gives the row +inf and no gradient (the case as
    didi_round_ce defines it), train_embed_sum poisons the dialog with NaN."""
    from didi_amd import ops

    g = torch.Generator().manual_seed(14)
    V, H = 50, 128
    ids = torch.randint(0, V, (4, 9), generator=g)
    dy = torch.randn(4, 9, H, generator=g)
    bad = ids.clone()
    bad[0, 0], bad[1, 3], bad[2, 2] = -100, V, 10 ** 12
    ok_rows = (bad == ids)
    guard = torch.full((V + 8, H), 7.0, device="cuda")  # the table sits inside a larger allocation: writes past it would show
    grad = guard[:V]
    grad.zero_()
    ops.embedding_backward(bad.cuda(), dy.cuda(), grad)
    want = torch.zeros(V, H, dtype=torch.float64)
    want.index_add_(0, ids[ok_rows], dy[ok_rows].double())
    assert rel_l2(grad, want) <= 1e-6 and float((guard[V:] - 7.0).abs().max()) == 0.0
    # cross-entropy backward: rows 0 and 2 carry targets outside [0, V)
    logits = torch.randn(5, V, generator=g)
    targets = torch.tensor([-100, 3, V, 7, 9])
    mask = torch.tensor([1, 1, 1, 0, 1])
    dev, nll = ops.ce_backward(logits.cuda().clone(), targets.cuda(), mask.cuda(), 1.0)
    assert torch.isinf(nll[0]) and torch.isinf(nll[2]) and float(nll[3]) == 0.0 and bool(torch.isfinite(nll[[1, 4]]).all())
    assert float(dev[[0, 2, 3]].abs().max()) == 0.0
    ref = torch.softmax(logits.double(), -1)
    ref[torch.arange(5), targets.clamp(0, V - 1)] -= 1.0
    assert rel_l2(dev[[1, 4]], ref[[1, 4]]) <= 1e-5
    # timesteps
    B, S, T = 3, 5, 20
    xt, te, pos, tok = torch.randn(B, S, H, generator=g), torch.randn(T + 1, H, generator=g), torch.randn(8, H, generator=g), torch.randn(H, generator=g)
    tt = torch.tensor([4, T + 1, -1])
    got = ops.train_embed_sum(xt.cuda(), te.cuda(), tt.cuda(), pos.cuda(), tok.cuda())
    assert bool(torch.isnan(got[1:]).all()) and rel_l2(got[0], xt[0] + te[4] + pos[:S] + tok) <= 1e-7


def test_embed_sum_seq_sum_target_backward_sumsq_axpby():
    from didi_amd import ops

    g = torch.Generator().manual_seed(5)
    B, S, H, T = 5, 9, 64, 20
    xt, te, pos, tok = torch.randn(B, S, H, generator=g), torch.randn(T + 1, H, generator=g), torch.randn(40, H, generator=g), torch.randn(2, H, generator=g)
    t = torch.randint(1, T + 1, (B, 1), generator=g)
    got = ops.train_embed_sum(xt.cuda(), te.cuda(), t.cuda(), pos.cuda(), tok[0].contiguous().cuda())
    want = xt + te[t.view(-1)][:, None, :] + pos[:S][None] + tok[0]
    assert rel_l2(got, want) <= 1e-7
    assert rel_l2(ops.seq_sum(xt.cuda()), xt.double().sum(1)) <= 1e-6
    hat, emb, x0, dxt, extra = (torch.randn(B, S, H, generator=g) for _ in range(5))
    mask = (torch.rand(B, S, generator=g) > 0.3).long()
    cb = torch.rand(B, generator=g)
    m = mask[:, :, None].double()
    want = m * (-0.3 * (hat - emb) - 0.7 * (hat - x0) + 0.11 * x0).double() + cb[:, None, None].double() * dxt.double() + extra.double()
    got = ops.train_target_backward(hat.cuda(), emb.cuda(), x0.cuda(), mask.cuda(), 0.3, 0.7, 0.11, cb.cuda(), dxt.cuda(), extra.cuda())
    assert rel_l2(got, want) <= 1e-6
    got = ops.train_target_backward(hat.cuda(), emb.cuda(), x0.cuda(), mask.cuda(), 0.3, 0.7, 0.11, cb.cuda(), dxt.cuda())
    assert rel_l2(got, want - extra.double()) <= 1e-6
    acc = torch.zeros(1, device="cuda")
    big = torch.randn(300001, generator=g)
    ops.sumsq_accumulate(big.cuda(), acc)
    ops.sumsq_accumulate(xt.cuda().view(-1), acc)
    assert abs(float(acc) - float((big.double() ** 2).sum() + (xt.double() ** 2).sum())) <= 1e-5 * float(acc)
    y = dxt.cuda().clone()
    ops.axpby(2.0, hat.cuda(), -0.5, y)
    assert rel_l2(y, 2.0 * hat.double() - 0.5 * dxt.double()) <= 1e-6


def test_adamw_kernel_vs_torch():
    from didi_amd import ops

    g = torch.Generator().manual_seed(6)
    p0 = torch.randn(1000, 37, generator=g)
    ref = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.AdamW([ref], lr=3e-3, weight_decay=0.01)
    p, m, v = p0.cuda().clone(), torch.zeros_like(p0).cuda(), torch.zeros_like(p0).cuda()
    half = torch.full((1,), 0.5, device="cuda")
    for step in range(1, 5):
        grad = torch.randn(1000, 37, generator=g)
        ref.grad = grad.double() * 0.5
        opt.step()
        ops.adamw_step(p, grad.cuda(), m, v, 3e-3, 0.9, 0.999, 1e-8, 0.01, step, gscale=half)
        assert rel_l2(p, ref.detach()) <= 1e-6, step


# ------------------------------------------------------------------ the whole step against the reference's autograd
def _trainer(name, precision="bf16x3", **kw):
    from didi_amd.synthetic import RawContext
    from didi_amd.trainer import DiDiTrainer

    t, meta = load_case(name)
    w = weights_of(load_case("tiny_weights")[0])
    run = meta["run"]
    kw.setdefault("hidden_dropout", 0.0)  # the gradient fixtures were generated without dropout (the trainer's default is the
    kw.setdefault("attention_dropout", 0.0)  # reference's 0.1 / 0.1, configs/train.yaml:22-23)
    tr = DiDiTrainer(w, spec_to_config(meta["spec"]), device="cuda", precision=precision, anchor_loss=run["anchor_loss"],
                     noise_factor=run["noise_factor"], **kw)
    ctx = RawContext(t["in.input_ids"].cuda(), t["in.attention_mask"].cuda())
    tgt = RawContext(t["in.target_ids"].cuda(), t["in.target_mask"].cuda())
    return t, meta, w, tr, ctx, tgt


@pytest.mark.parametrize("name", CASES)
def test_all_parameter_gradients_vs_reference_autograd(name):
    """Every gradient the reference's ``loss.backward()`` produced (63 tensors), from the hand-written backward.
    bf16x3 GEMMs (fp32-class): rel-L2 <= 2e-4 per tensor, scalars 1e-4 relative."""
    from didi_amd.training import metric_message

    t, meta, w, tr, ctx, tgt = _trainer(name)
    out = tr.forward_backward(ctx, tgt, draws=noise_list(t, int(meta["n_noise"])), t=t["t"], keep=True)
    assert rel_l2(out["x_0_hat"], t["out.x_0_hat"]) <= 3e-5
    assert torch.allclose(metric_message(out).cpu(), t["out.scalars"], rtol=1e-4, atol=1e-9)
    assert rel_l2(out["d_x_0_hat"], t["grad.x_0_hat"]) <= 1e-4
    want = {k[2:]: v for k, v in t.items() if k.startswith("g.")}
    assert len(want) == int(meta["n_grads"]) == 63
    assert set(want) == set(out["grads"]) == set(tr.params), (set(want) ^ set(out["grads"]))
    for k in want:
        assert tuple(out["grads"][k].shape) == tuple(want[k].shape), k
    # key biases: softmax is invariant to a per-query shift of the scores, so these gradients are analytically ZERO and the
    # reference's values are rounding noise (norm ~1e-11): compared in absolute terms against the query bias's scale instead
    key_bias = [k for k in want if k.endswith(".key.bias")]
    assert len(key_bias) == 2 * meta["spec"]["dec_layers"]
    for k in key_bias:
        scale = float(want[k.replace(".key.", ".query.")].norm())
        assert float(want.pop(k).norm()) <= 1e-5 * scale and float(out["grads"][k].norm()) <= 1e-4 * scale, k
    errs = {k: rel_l2(out["grads"][k], v) for k, v in want.items()}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print(f"\n[{name}] worst gradient rel-L2: " + ", ".join(f"{k.split('decoder.')[-1]} {e:.1e}" for k, e in worst))
    for k, e in errs.items():
        assert e <= 2e-4, (k, e)
    # rows that must stay untouched: padding_idx of emb, position rows past S, token type 1
    S = t["in.target_ids"].shape[1]
    assert float(out["grads"]["emb.weight"][0].abs().max()) == 0.0
    assert float(out["grads"]["decoder.embeddings.position_embeddings.weight"][S:].abs().max()) == 0.0
    assert float(out["grads"]["decoder.embeddings.token_type_embeddings.weight"][1].abs().max()) == 0.0


@pytest.mark.parametrize("precision", ["fp16x2", "fp16"])
def test_gradients_in_faster_precisions_stay_close(precision):
    """One- and two-product GEMMs in forward and backward: gradients within 1e-2 (fp16) / 3e-3 (fp16x2) of autograd's in
    rel-L2 over the large tensors -- reported, with a loose bound: mixed-precision training tolerates far more."""
    t, meta, w, tr, ctx, tgt = _trainer("train_tiny_plain", precision)
    out = tr.forward_backward(ctx, tgt, draws=noise_list(t, int(meta["n_noise"])), t=t["t"])
    errs = {k[2:]: rel_l2(out["grads"][k[2:]], v) for k, v in t.items() if k.startswith("g.") and v.numel() >= 128 * 128}
    print(f"\n[{precision}] max weight-gradient rel-L2 {max(errs.values()):.2e}")
    assert max(errs.values()) <= (1e-2 if precision == "fp16" else 3e-3)


def test_optimizer_step_vs_torch_adamw_lambdalr_clip():
    """Three optimiser steps on the fixture's gradients (scaled up so that clipping engages on step 1) against
    torch.optim.AdamW(lr=1) + LambdaLR(linear) + clip_grad_norm_(1.0) on fp64 copies."""
    from didi_amd.trainer import lr_linear

    optk = dict(weight_decay=0.01, scheduler="linear", lr=1e-3, warmup_steps=1, min_lr=0.0, decay_range=4)
    t, meta, w, tr, ctx, tgt = _trainer("train_tiny_plain", opt_kwargs=optk)
    names = list(tr.params)
    ref = {k: torch.nn.Parameter(v.detach().double().cpu().clone()) for k, v in tr.params.items()}
    opt = torch.optim.AdamW(ref.values(), lr=1.0, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: lr_linear(s, 1e-3, 0.0, 1, 4))
    grads0 = {k: t["g." + k].cuda().float().contiguous() for k in names}
    for step, mult in enumerate((400.0, 1.0, 0.01)):
        grads = {k: (g * mult).contiguous() for k, g in grads0.items()}
        for k in names:
            ref[k].grad = grads[k].double().cpu()
        want_norm = torch.nn.utils.clip_grad_norm_(ref.values(), 1.0)
        opt.step()
        sched.step()
        norm = tr.optimizer_step(grads)
        assert abs(float(norm) - float(want_norm)) <= 1e-4 * float(want_norm)
        worst = max(rel_l2(tr.params[k], ref[k].detach()) for k in names)
        assert worst <= 1e-6, (step, worst)
    assert tr.step_count == 3
    # planes follow the masters
    from didi_amd import ops
    L0 = tr._lin[0]["up"]
    assert rel_l2(ops.join(L0.hi, L0.lo), L0.w) <= 2e-5 and torch.equal(L0.w, tr.params[tr._layer_names[0]["up"] + ".weight"])


def test_training_steps_reduce_the_loss_and_sampler_follows():
    """A dozen steps on one fixed batch (fixed noise): the loss must fall, stay finite, and a sampler built from the trainer's
    state dict must see the trained weights."""
    optk = dict(weight_decay=0.01, scheduler="rsqrt", lr=2e-3, warmup_steps=2, min_lr=None)
    t, meta, w, tr, ctx, tgt = _trainer("train_tiny_plain", opt_kwargs=optk)
    draws = noise_list(t, int(meta["n_noise"]))
    losses = []
    for _ in range(14):
        out = tr.training_step(ctx, tgt, draws=draws, t=t["t"])
        losses.append(float(out["loss"]))
    print("\nlosses:", [round(x, 4) for x in losses])
    assert all(math.isfinite(x) for x in losses) and losses[-1] < 0.8 * losses[1]
    sd = tr.state_dict()
    assert set(w) <= set(sd)
    assert not torch.equal(sd["emb.weight"].cpu(), w["emb.weight"]) and torch.equal(sd["encoder.embed_tokens.weight"].cpu(), w["encoder.embed_tokens.weight"])
    s = tr.sampler(precision="bf16x3")
    assert torch.equal(s.emb.weight, tr.params["emb.weight"])


def test_rsqrt_lr_lambda():
    from didi_amd.trainer import lr_linear, lr_rsqrt_with_warmup

    assert lr_rsqrt_with_warmup(25, 1e-4, None, 100) == pytest.approx(1e-4 * 0.5)
    assert lr_rsqrt_with_warmup(400, 1e-4, None, 100) == pytest.approx(1e-4 * 0.5)
    assert lr_rsqrt_with_warmup(10**8, 1e-4, 1e-5, 100) == 1e-5
    assert lr_linear(0, 1e-4, 0.0, 2500, 100000) == 1e-4 and lr_linear(100001, 1e-4, 0.0, 2500, 100000) == 0.0
    assert lr_linear(51250, 1e-4, 0.0, 2500, 100000) == pytest.approx(0.5e-4)
    with pytest.raises(ValueError):
        lr_linear(3, 1e-4, None, 1, 2)


# ------------------------------------------------------------------ dropout (configs/train.yaml:22-23)
def _philox_words(n4, seed, offset):
    from oracle.philox import raw_stream  # numpy Philox4x32-10, pinned by the Random123 known-answer vectors

    return torch.from_numpy(raw_stream(n4, seed, offset).astype("int64"))


def _thr(p):
    import numpy as np

    return int(min(float(np.float32(p) * np.float32(4294967296.0)), 4294967040.0))


def test_hidden_dropout_mask_is_the_documented_philox_stream():
    from didi_amd import ops

    g = torch.Generator().manual_seed(8)
    x, r = torch.randn(37, 64, generator=g), torch.randn(37, 64, generator=g)
    p, seed, off = 0.1, 0x1234567855AA, 5 << 34
    keep = (_philox_words(x.numel() // 4, seed, off).view(-1) >= _thr(p)).view(x.shape)
    got = ops.dropout_add(x.cuda(), r.cuda(), p, seed, off)
    assert rel_l2(got, torch.where(keep, x.double() / (1 - float(torch.tensor(p, dtype=torch.float32))), torch.zeros(())) + r.double()) <= 1e-6
    back = ops.dropout_add(x.cuda(), None, p, seed, off)
    assert torch.equal(back == 0, ~keep.cuda() | (x.cuda() == 0))
    big = ops.dropout_add(torch.ones(1 << 20, device="cuda"), None, 0.2, 7, 0)
    assert abs(float((big == 0).float().mean()) - 0.2) < 3e-3 and abs(float(big.mean()) - 1.0) < 5e-3
    assert torch.equal(ops.dropout_add(x.cuda(), r.cuda(), 0.0, seed, off), (x + r).cuda())


@pytest.mark.parametrize("B,heads,Sq,Sk,D,causal,masked,p", [(3, 4, 10, 10, 32, True, False, 0.1), (2, 12, 64, 64, 64, True, False, 0.2),
                                                             (3, 4, 10, 13, 32, False, True, 0.1), (2, 3, 70, 90, 64, False, True, 0.3),
                                                             (2, 4, 24, 24, 32, True, False, 0.0), (2, 2, 64, 80, 32, False, True, 0.2),
                                                             (1, 3, 96, 96, 64, True, False, 0.1)])
def test_attention_train_forward_backward_with_dropout_vs_autograd(B, heads, Sq, Sk, D, causal, masked, p):
    """O = dropout(softmax(...)) V and its backward against torch autograd in fp64 fed the SAME mask, which is rebuilt on the host
    from the documented Philox stream."""
    from didi_amd import ops

    g = torch.Generator().manual_seed(B * 1000 + Sk)
    H = heads * D
    q, k, v, dO = (torch.randn(B, s_, H, generator=g) for s_ in (Sq, Sk, Sk, Sq))
    mask = torch.ones(B, Sk, dtype=torch.int32)
    if masked:
        for b in range(B):
            mask[b, max(1, Sk - 1 - 3 * b):] = 0
    seed, off = 0xBADB01, 3 << 34
    Sk4 = (Sk + 3) // 4 * 4
    keep = torch.ones(B, heads, Sq, Sk, dtype=torch.bool)
    if p > 0:
        keep = (_philox_words(B * heads * Sq * Sk4 // 4, seed, off).view(B, heads, Sq, Sk4) >= _thr(p))[..., :Sk]
    qd, kd, vd = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    sh = lambda t, S: t.view(B, S, heads, D).transpose(1, 2)  # noqa: E731
    s = sh(qd, Sq) @ sh(kd, Sk).transpose(-1, -2) / math.sqrt(D)
    neg = torch.finfo(torch.float32).min
    if masked:
        s = s + (1.0 - mask[:, None, None, :].double()) * neg
    if causal:
        s = s + torch.triu(torch.full((Sq, Sk), neg, dtype=torch.float64), 1)
    pr = torch.softmax(s, -1) * keep.double() / (1.0 - float(torch.tensor(p, dtype=torch.float32)))
    o = (pr @ sh(vd, Sk)).transpose(1, 2).reshape(B, Sq, H)
    o.backward(dO.double())
    kw = dict(causal=causal, key_mask=mask.cuda() if masked else None, drop_p=p, seed=seed, offset=off)
    got = ops.attention_train(q.cuda(), k.cuda(), v.cuda(), heads, **kw)
    assert rel_l2(got, o.detach()) <= 2e-5
    gq, gk, gv = ops.attention_backward(q.cuda(), k.cuda(), v.cuda(), dO.cuda(), heads, **kw)
    assert rel_l2(gq, qd.grad) <= 2e-5 and rel_l2(gk, kd.grad) <= 2e-5 and rel_l2(gv, vd.grad) <= 2e-5
    if p > 0:  # the mask really dropped something, at about the asked rate on the unmasked region
        rate = 1.0 - keep.float().mean().item()
        assert abs(rate - p) < 0.05


def test_training_step_with_dropout_directional_derivative_and_determinism():
    """With dropout on (hidden 0.1, attention 0.1) the masks cannot match torch's generator, so the whole-step check is a property:
    for a FIXED dropout seed, noise and t the loss is a deterministic function of the parameters, and the hand-written backward
    (which regenerates every mask) must give its directional derivative: (L(theta + e u) - L(theta - e u)) / 2e = <grad, u> for
    u = grad / |grad|.  A backward mask that disagreed with the forward's would be off by O(p) = 10 %; the bar is 1 %."""
    t, meta, w, tr, ctx, tgt = _trainer("train_tiny_plain", hidden_dropout=0.1, attention_dropout=0.1)
    draws = noise_list(t, int(meta["n_noise"]))
    kw = dict(draws=draws, t=t["t"], dropout_seed=1234)
    a = tr.forward_backward(ctx, tgt, **kw)
    b = tr.forward_backward(ctx, tgt, **kw)
    assert float(a["loss"]) == float(b["loss"]) and all(torch.equal(a["grads"][k], b["grads"][k]) for k in a["grads"])
    c = tr.forward_backward(ctx, tgt, draws=draws, t=t["t"], dropout_seed=99)
    assert float(c["loss"]) != float(a["loss"])
    _, _, _, plain, _, _ = _trainer("train_tiny_plain")
    d = plain.forward_backward(ctx, tgt, draws=draws, t=t["t"])
    assert abs(float(a["loss"]) - float(d["loss"])) > 1e-4, "dropout must change the forward"
    flat = tr.flatten_gradients(a["grads"])[: tr._n_flat]
    norm = float(flat.norm())
    u = flat / norm
    eps = 0.02
    theta = tr._flat_p.clone()
    vals = []
    for sgn in (1.0, -1.0):
        tr._flat_p.copy_(theta + sgn * eps * u)
        tr.refresh_planes()
        vals.append(float(tr.forward_backward(ctx, tgt, **kw)["loss"].double()))
    tr._flat_p.copy_(theta)
    tr.refresh_planes()
    fd = (vals[0] - vals[1]) / (2 * eps)
    print(f"\n[dropout step] |grad| {norm:.5f}  finite difference {fd:.5f}  losses {vals}")
    assert abs(fd - norm) <= 0.01 * norm


@pytest.mark.parametrize("precision", ["bf16x3", "fp16x2"])
def test_overlapped_reducer_covers_every_gradient_once(precision):
    """The overlapped all-reduce (launched per bucket while the backward runs) with a stand-in collective that doubles its
    slice in place -- two identical ranks summed -- must leave, after the 1 / world scaling, exactly the gradients of the plain
    path, every element reduced once, buckets launched from the last layer backwards with the head last."""
    from didi_amd.trainer import _OverlappedReducer

    t, meta, w, tr, ctx, tgt = _trainer("train_tiny_plain", precision)
    kw = dict(draws=noise_list(t, int(meta["n_noise"])), t=t["t"])
    plain = tr.flatten_gradients(tr.forward_backward(ctx, tgt, **kw)["grads"])[: tr._n_flat]
    calls = []

    def fake(slice_):
        calls.append((slice_.data_ptr(), slice_.numel()))
        slice_.mul_(2.0)
        return None

    sink = _OverlappedReducer(tr, 2, all_reduce=fake)
    out = tr.forward_backward(ctx, tgt, grad_sink=sink, **kw)
    assert torch.equal(sink.flat[: tr._n_flat], plain)
    base = sink.flat.data_ptr()
    spans = sorted(((p_ - base) // 4, (p_ - base) // 4 + n) for p_, n in calls)
    assert spans[0][0] == 0 and spans[-1][1] == tr._n_flat and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    starts = [(p_ - base) // 4 for p_, _ in calls]
    assert starts[:-1] == sorted(starts[:-1], reverse=True) and starts[-1] == 0  # last layers first, the head at the end
    for k, g_ in out["grads"].items():  # the caller's dict now views the reduced buffer
        off = tr._offsets[k]
        assert g_.data_ptr() == base + 4 * off and tuple(g_.shape) == tuple(tr.params[k].shape)


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_save_and_resume_continues_an_uninterrupted_run(tmp_path, dropout):
    """Checkpoint / resume (src/training.py:27-33,47: ModelCheckpoint every save_interval steps, trainer.fit(ckpt_path=resume)):
    three steps, save, then a FRESH trainer built from the original weights loads the file and takes step four -- parameters,
    both AdamW moments, the step count (LR lambda, bias correction) and the step's loss must equal those of a trainer that took
    the four steps without interruption, to the run-to-run noise of the backward's fp32 atomic adds (embedding gradients:
    two identical runs differ in the last bits; bar 1e-6 relative, a resume WITHOUT the optimiser state is 1e-3 away).  The
    file is plain tensors: torch.load(weights_only=True) reads it,
    and it loads into a DiDiSampler like a reference checkpoint."""
    from didi_amd.checkpoint import load_lightning_checkpoint

    optk = dict(lr=3e-3, warmup_steps=2, decay_range=10, min_lr=1e-4, scheduler="linear")
    kw = dict(opt_kwargs=optk, hidden_dropout=dropout, attention_dropout=dropout)
    t, meta, w, a, ctx, tgt = _trainer("train_tiny_plain", **kw)
    _, _, _, b, _, _ = _trainer("train_tiny_plain", **kw)

    def step(tr, k):
        torch.manual_seed(100 + k)  # Philox keys of the noise, the timesteps and the dropout masks come from the global generator
        return tr.training_step(ctx, tgt)

    for k in range(3):
        step(a, k)
        step(b, k)
    path = tmp_path / "step_3.ckpt"
    b.save_checkpoint(path)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert ck["global_step"] == 3 and set(ck["didi_amd_trainer"]["exp_avg"]) == set(b.params)
    _, _, _, c, _, _ = _trainer("train_tiny_plain", **kw)  # a restart: original weights, zero moments, step 0
    c.load_checkpoint(path)
    assert c.step_count == 3
    oa, oc = step(a, 3), step(c, 3)
    assert torch.allclose(oa["metrics"], oc["metrics"], rtol=1e-5, atol=0) and torch.allclose(oa["grad_norm"], oc["grad_norm"], rtol=1e-5)
    e_p, e_m, e_v = rel_l2(c._flat_p, a._flat_p), rel_l2(c._flat_m, a._flat_m), rel_l2(c._flat_v, a._flat_v)
    print(f"\n[resume, dropout {dropout}] after step 4, resumed vs uninterrupted: parameters {e_p:.1e}, exp_avg {e_m:.1e}, exp_avg_sq {e_v:.1e}")
    assert e_p <= 1e-6 and e_m <= 1e-5 and e_v <= 1e-5
    assert a.step_count == c.step_count == 4
    # without the optimiser state the run does NOT continue: same weights, zero moments, step 0 -> another step four
    _, _, _, d, _, _ = _trainer("train_tiny_plain", **kw)
    for k_, p_ in d.params.items():
        p_.copy_(ck["state_dict"][k_])
    d.refresh_planes()
    step(d, 3)
    assert rel_l2(d._flat_p, a._flat_p) > 100 * max(e_p, 1e-8)
    # mismatched optimiser settings are refused; the file is also a sampler checkpoint in the reference's key layout
    _, _, _, e, _, _ = _trainer("train_tiny_plain", opt_kwargs=dict(optk, lr=1e-3))
    with pytest.raises(ValueError):
        e.load_checkpoint(path)
    cfg = spec_to_config(meta["spec"])
    smp = load_lightning_checkpoint(path, device="cuda", precision="bf16x3", dec_heads=cfg.dec_heads, enc_heads=cfg.enc_heads)
    assert torch.equal(smp.emb.weight, b.params["emb.weight"])


def test_unfrozen_encoder_is_refused():
    from didi_amd.trainer import DiDiTrainer

    _, meta = load_case("train_tiny_plain")
    with pytest.raises(NotImplementedError):
        DiDiTrainer(weights_of(load_case("tiny_weights")[0]), spec_to_config(meta["spec"]), device="cuda", freeze_encoder=False)
