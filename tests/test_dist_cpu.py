"""world_size-2 gloo test of the multi-GPU plumbing (shard bounds, id gather, disjoint Philox ranges)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from didi_amd.dist import gather_ids, philox_offset, shard_bounds


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
