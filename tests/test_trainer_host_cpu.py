"""Host-side logic of the training step that needs no GPU: the two learning-rate lambdas of ``configure_optimizers``
(src/diffusion/model.py:271-308), restated in didi_amd/trainer.py, against values worked out from the reference's formulas, and
their use under torch's LambdaLR exactly as the reference wires them (AdamW(lr = 1) x lambda, model.py:138-157)."""
import math

import pytest
import torch


def test_linear_lambda_values():
    from didi_amd.trainer import lr_linear

    kw = dict(max_lr=1e-4, min_lr=0.0, warmup=2500, decay_range=100_000)  # configs/train.yaml:33-39
    assert lr_linear(0, **kw) == 1e-4 and lr_linear(2499, **kw) == 1e-4   # "warm-up" of the linear lambda is a plateau at max_lr
    assert lr_linear(2500, **kw) == pytest.approx(1e-4)
    assert lr_linear(51250, **kw) == pytest.approx(0.5e-4)
    assert lr_linear(100_000, **kw) == pytest.approx(0.0) and lr_linear(100_001, **kw) == 0.0
    assert lr_linear(60_000, 3e-4, 1e-5, 10_000, 110_000) == pytest.approx((3e-4 - 1e-5) * 0.5 + 1e-5)
    with pytest.raises(ValueError):
        lr_linear(3, 1e-4, None, 1, 2)


def test_rsqrt_lambda_values():
    from didi_amd.trainer import lr_rsqrt_with_warmup as f

    assert f(0, 1e-4, None, 100) == 0.0
    assert f(25, 1e-4, None, 100) == pytest.approx(1e-4 * 0.5)
    assert f(100, 1e-4, None, 100) == pytest.approx(1e-4)
    assert f(400, 1e-4, None, 100) == pytest.approx(1e-4 * 0.5)
    assert f(10**8, 1e-4, 1e-5, 100) == 1e-5             # the floor
    assert f(4, 2e-3, None, 0) == pytest.approx(1e-3)    # warmup 0: max_lr / sqrt(step)
    with pytest.raises(ZeroDivisionError):                # the reference divides by step there too (model.py:291): step 0, warmup 0
        f(0, 1e-4, None, 0)


def test_lambdas_drive_lambdalr_like_the_reference():
    from didi_amd.trainer import DEFAULT_OPT, lr_linear

    p = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.AdamW([p], lr=1.0, weight_decay=DEFAULT_OPT["weight_decay"])
    o = dict(DEFAULT_OPT, warmup_steps=2, decay_range=6)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: lr_linear(s, o["lr"], o["min_lr"], o["warmup_steps"], o["decay_range"]))
    seen = []
    for _ in range(8):
        seen.append(opt.param_groups[0]["lr"])  # the k-th optimizer.step() runs with lambda(k - 1): what DiDiTrainer.lr_at(step_count) returns
        p.grad = torch.ones(3)
        opt.step()
        sched.step()
    want = [lr_linear(k, o["lr"], o["min_lr"], o["warmup_steps"], o["decay_range"]) for k in range(8)]
    assert seen == pytest.approx(want) and seen[0] == 1e-4 and seen[-1] == 0.0 and math.isclose(seen[4], 0.5e-4)


def test_layer_buckets_tile_the_flat_buffer():
    """Bucket plan of the overlapped gradient reduction: launched from the last layer backwards, two layers per bucket, the head
    (everything before layer 0) last; together they cover [0, n_flat) exactly once."""
    from didi_amd.trainer import layer_buckets

    for L, per in ((12, 2), (5, 2), (1, 2), (7, 3), (2, 1)):
        head = 1000
        width = 7_087_872
        ranges = [(head + i * width, head + (i + 1) * width) for i in range(L)]
        buckets, h = layer_buckets(ranges, head + L * width, per)
        assert h == (0, head)
        assert [b[0] for b in buckets] == sorted((b[0] for b in buckets), reverse=True)  # launch order follows the backward
        covered = sorted([(b[1], b[2]) for b in buckets] + [h])
        assert covered[0][0] == 0 and covered[-1][1] == head + L * width
        assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
        assert all((b[2] - b[1]) // width <= per for b in buckets)
        # a bucket is launched when its FIRST (lowest) layer is done: by then the layers above it in the bucket are done too
        for first, start, end in buckets:
            assert start == ranges[first][0]
    with pytest.raises(AssertionError):
        layer_buckets([(0, 10), (12, 20)], 20)
