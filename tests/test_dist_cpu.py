"""world_size-2 gloo test of the multi-GPU plumbing (shard bounds, id gather, disjoint Philox ranges)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from didi_amd.dist import gather_ids, philox_offset, sample_sharded, shard_bounds


def test_shard_bounds_cover_batch():
    for B in (1, 7, 8, 256, 4096, 1023):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_philox_ranges_disjoint():
    per_call = 2001 * 512 * 84 * 768 // 4  # largest config: counters one sample() call can consume
    assert per_call < (1 << 40)
    assert all(philox_offset(r + 1) - philox_offset(r) > per_call for r in range(8))


def _worker(rank, world, port, B, S, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = shard_bounds(B, world, rank)
    full = torch.arange(B * S, dtype=torch.int64).reshape(B, S) * 7 + 3
    lens = (torch.arange(B, dtype=torch.int32) % S) + 1
    ids, lengths = gather_ids(full[a:b].clone(), lens[a:b].clone(), B, world)
    q.put((rank, torch.equal(ids, full), torch.equal(lengths, lens)))
    dist.destroy_process_group()


def test_gather_ids_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    B, S, world = 7, 5, 2  # uneven shards (4 + 3)
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(got) == [(0, True, True), (1, True, True)]


def _stub_sample(local, model, mode, step_freq, noise_factor=1, max_len=-1, tokenizer=None, return_ids=False, philox_offset=0, **kw):
    """Stands in for didi_amd.sampling.sample: ids that encode which dialog and which Philox range produced them."""
    assert return_ids and mode == "euler" and step_freq == 10 and model == "the-model"
    B, C = local.input_ids.shape
    S = len(range(C)[:max_len])
    ids = local.input_ids[:, :1] * 1000 + torch.arange(S)[None, :]
    ids[:, -1] = philox_offset >> 40  # the rank whose counter range was used
    lengths = (local.input_ids[:, 0] % S + 1).to(torch.int32)
    return ids, lengths


def _sharded_worker(rank, world, port, B, C, local_only, q):
    from didi_amd.synthetic import RawContext

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gids = (torch.arange(B * C, dtype=torch.int64).reshape(B, C) % 97) + 3 + torch.arange(B)[:, None] * 100
    gmask = torch.ones(B, C, dtype=torch.int64)
    a, b = shard_bounds(B, world, rank)
    if local_only:  # the rank holds (and would upload) only its own rows
        ids, lengths = sample_sharded(RawContext(gids[a:b], gmask[a:b]), "the-model", "euler", 10, rank=rank, world=world,
                                      max_len=C, eos_id=2, global_batch=B, sample_fn=_stub_sample)
    else:
        ids, lengths = sample_sharded(RawContext(gids, gmask), "the-model", "euler", 10, rank=rank, world=world, max_len=C,
                                      eos_id=2, sample_fn=_stub_sample)
    want = gids[:, :1] * 1000 + torch.arange(C)[None, :]
    owner = torch.tensor([r for r in range(world) for _ in range(*shard_bounds(B, world, r))])
    want[:, -1] = owner
    ok = tuple(ids.shape) == (B, C) and torch.equal(ids, want) and torch.equal(lengths, (gids[:, 0] % C + 1).to(torch.int32))
    bad_shard = False
    if local_only and b - a > 0:
        try:
            sample_sharded(RawContext(gids[a:b][:-1], gmask[a:b][:-1]), "the-model", "euler", 10, rank=rank, world=world, max_len=C,
                           global_batch=B, sample_fn=_stub_sample)
        except ValueError:
            bad_shard = True
    q.put((rank, bool(ok), bad_shard or not local_only or b - a == 0))
    dist.destroy_process_group()


def _run(target, args, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(got)


def test_sample_sharded_world2_gloo_global_batch_uneven():
    """sample_sharded itself under world_size 2: uneven shards (4 + 3), rows come back in global order on every rank and each
    rank sampled with its own Philox range."""
    assert _run(_sharded_worker, (7, 6, False), 2) == [(0, True, True), (1, True, True)]


def test_sample_sharded_world2_gloo_local_shards():
    """Each rank passes only its own rows (what bench.py uploads); a shard of the wrong size is rejected."""
    assert _run(_sharded_worker, (5, 4, True), 2) == [(0, True, True), (1, True, True)]


def test_sample_sharded_world2_gloo_empty_shard():
    """B = 1 < world: rank 1 has nothing to sample and still takes part in the gather."""
    assert _run(_sharded_worker, (1, 4, False), 2) == [(0, True, True), (1, True, True)]
    assert _run(_sharded_worker, (1, 4, True), 2) == [(0, True, True), (1, True, True)]


def _grad_worker(rank, world, port, bucket_bytes, q):
    from didi_amd.dist import allreduce_gradients

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = [(7, 5), (3,), (64, 33), (1,), (130,)]
    g = torch.Generator().manual_seed(100 + rank)
    grads = [torch.randn(*s, generator=g) for s in shapes]
    metrics = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0]) * (rank + 1)
    want = []
    for s in shapes:  # the mean over ranks of what each rank drew
        acc = torch.zeros(*s)
        for r in range(world):
            gr = torch.Generator().manual_seed(100 + r)
            per = [torch.randn(*s2, generator=gr) for s2 in shapes]
            acc += per[shapes.index(s)]
        want.append(acc / world)
    m = allreduce_gradients(grads, world, metrics=metrics, bucket_bytes=bucket_bytes)
    ok = all(torch.allclose(a, b, atol=1e-6) for a, b in zip(grads, want))
    ok = ok and torch.allclose(m, torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0]) * (sum(range(1, world + 1)) / world))
    q.put((rank, bool(ok), True))
    dist.destroy_process_group()


def test_gradient_allreduce_world2_gloo_one_bucket_and_many():
    """The training step's gradient all-reduce + the five metric scalars in the same message: one bucket, and buckets smaller
    than a tensor (every tensor its own bucket) give the same averages on both ranks."""
    assert _run(_grad_worker, (64 << 20,), 2) == [(0, True, True), (1, True, True)]
    assert _run(_grad_worker, (64,), 2) == [(0, True, True), (1, True, True)]


def _flat_worker(rank, world, port, bucket_bytes, q):
    from didi_amd.dist import allreduce_flat

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, tail = 1000, 8
    mine = torch.randn(n + tail, generator=torch.Generator().manual_seed(7 + rank))
    want = sum(torch.randn(n + tail, generator=torch.Generator().manual_seed(7 + r)) for r in range(world)) / world
    calls = []
    real = dist.all_reduce

    def counting(t, **kw):
        calls.append(t.numel())
        return real(t, **kw)

    dist.all_reduce = counting
    try:
        got = allreduce_flat(mine, world, bucket_bytes=bucket_bytes, tail=tail)
    finally:
        dist.all_reduce = real
    per = max(1, bucket_bytes // 4)
    ok = got.data_ptr() == mine.data_ptr() and torch.allclose(got, want, atol=1e-6)
    # buckets tile the buffer exactly, and the metric tail never travels alone
    ok = ok and sum(calls) == n + tail and calls[-1] > tail and all(c == per for c in calls[:-1])
    q.put((rank, bool(ok), True))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2_gloo():
    """The trainer's contiguous gradient buffer: buckets are slices (no packing), the metric tail rides in the last one."""
    assert _run(_flat_worker, (64 << 20,), 2) == [(0, True, True), (1, True, True)]
    assert _run(_flat_worker, (400,), 2) == [(0, True, True), (1, True, True)]    # 100-float buckets: 10 of them, last = 100 + 8
    assert _run(_flat_worker, (4 * 252,), 2) == [(0, True, True), (1, True, True)]  # remainder 244 + 8 <= 252 + 8: folded into one


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the parent starts two ranks itself (bench.py spawn_ranks),
    rank 0 prints ONE JSON line, both ranks took part (world_size_seen 2, gloo), exit code 0.  --device cpu swaps the sampler for a
    stub: what is rehearsed is the launch path the driver's N > 1 command takes."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--device", "cpu", "--steps", "2", "--warmup", "1", "--batch", "5"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["world_size_seen"] == 2 and res["config"]["backend"] == "gloo"
    assert res["config"]["global_batch"] == 10 and res["steps"] == 2 and res["valid"] is False


def test_bench_spawn_relays_a_failing_rank():
    """A rank that fails makes the bare `--gpus N` command exit non-zero instead of hanging on the survivors' collectives."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--device", "cpu", "--steps", "1", "--warmup", "0", "--batch", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
