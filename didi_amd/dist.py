"""Multi-GPU sampling: shard the dialog batch over ranks, gather token ids at the end.

Dialogs are independent (no cross-sample op anywhere in the loop: LayerNorm is per token,
attention per dialog), so the batch is partitioned contiguously across the ranks of one node, one
process per GPU, weights replicated.  There is NO collective inside the sampling loop; the only
exchange is one all-gather of the decoded ids [B/world, S] (+ lengths) over RCCL at the end
(SURVEY.md section 2, C4).  The reference samples on a single device (scripts/sample.py:27,55)
and has no counterpart for this file; its DDP (src/training.py:37) is a training-only mechanism.

Backend: "nccl" (= RCCL on ROCm) when CUDA tensors are used, "gloo" for the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(device_type="cuda"):
    """-> (rank, world, local_rank).  Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if device_type == "cuda":
        torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        kw = {}
        if device_type == "cuda":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group("nccl" if device_type == "cuda" else "gloo", rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def shard_bounds(batch: int, world: int, rank: int):
    """Contiguous, balanced split: the first batch % world ranks get one extra dialog."""
    base, extra = divmod(batch, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def philox_offset(rank: int) -> int:
    """Disjoint Philox counter ranges per rank (one sample() call uses < 2^36 counters)."""
    return rank << 40


def gather_ids(ids: torch.Tensor, lengths: torch.Tensor, batch: int, world: int):
    """All-gather per-rank (ids [b_r, S] int64, lengths [b_r] int32) -> ([batch, S], [batch]) on every rank.

    Shards may differ by one row; every rank pads to the largest shard so that one
    all_gather_into_tensor moves everything (ids and lengths packed into one int32 message:
    config 3 is 512 x 65 x 4 B = 133 KB per GPU, latency-bound).
    """
    S = ids.shape[1]
    if world == 1:
        return ids, lengths
    rows = -(-batch // world)
    packed = torch.zeros((rows, S + 1), dtype=torch.int32, device=ids.device)
    packed[: ids.shape[0], :S] = ids.to(torch.int32)
    packed[: ids.shape[0], S] = lengths.to(torch.int32)
    out = torch.empty((world * rows, S + 1), dtype=torch.int32, device=ids.device)
    dist.all_gather_into_tensor(out, packed)
    keep = []
    for r in range(world):
        a, b = shard_bounds(batch, world, r)
        keep.append(out[r * rows: r * rows + (b - a)])
    out = torch.cat(keep, 0)
    return out[:, :S].to(torch.int64), out[:, S].contiguous()


def sample_sharded(raw_context, model, mode, step_freq, *, rank, world, noise_factor=1, max_len=-1, eos_id=None,
                   global_batch=None, sample_fn=None, **kw):
    """Sample this rank's shard and all-gather the ids: every rank returns (ids [B, S], lengths [B]) of the WHOLE batch.

    ``raw_context`` is the global batch (every rank passes the same tensors; the rank's rows are sliced out here) or, with
    ``global_batch=B``, only this rank's shard -- rows ``shard_bounds(B, world, rank)`` of the global batch -- so that a rank
    never uploads dialogs it does not sample (bench.py).  A rank whose shard is empty (B < world) samples nothing and
    contributes no rows.  ``sample_fn`` replaces ``didi_amd.sampling.sample`` (the CPU gloo tests pass a stub).
    """
    from .synthetic import RawContext

    if sample_fn is None:
        from .sampling import sample as sample_fn
    if global_batch is None:
        B = raw_context.input_ids.shape[0]
        a, b = shard_bounds(B, world, rank)
        local = RawContext(raw_context.input_ids[a:b].contiguous(), raw_context.attention_mask[a:b].contiguous())
    else:
        B = int(global_batch)
        a, b = shard_bounds(B, world, rank)
        local = raw_context
        if local.input_ids.shape[0] != b - a:
            raise ValueError(f"rank {rank} of {world}: shard of a batch of {B} has {b - a} rows, got {local.input_ids.shape[0]}")

    class _Eos:
        eos_token_id, sep_token_id = eos_id, None

    if b > a:
        ids, lengths = sample_fn(local, model, mode, step_freq, noise_factor=noise_factor, max_len=max_len,
                                 tokenizer=_Eos() if eos_id is not None else None, return_ids=True,
                                 philox_offset=philox_offset(rank), **kw)
    else:  # empty shard: the reply length still has to be known for the gather
        S = kw.get("target_len") or len(range(local.input_ids.shape[1])[:max_len])  # src/sampling.py:11
        dev = local.input_ids.device
        ids, lengths = torch.empty((0, S), dtype=torch.int64, device=dev), torch.empty((0,), dtype=torch.int32, device=dev)
    return gather_ids(ids, lengths, B, world)


# ------------------------------------------------------------------ training: gradient all-reduce
def allreduce_gradients(grads, world, metrics=None, bucket_bytes=64 << 20, average=True):
    """Data-parallel gradient reduction for the training step: replaces Lightning's ``DDPStrategy`` (src/training.py:37) and the
    five per-step ``log_dict(sync_dist=True)`` scalar reductions (src/diffusion/model.py:241-242).

    ``grads``: list of fp32 tensors (one per trainable parameter, same order on every rank), reduced IN PLACE.  They are packed
    into flat buckets of ``bucket_bytes`` (default 64 MiB: xGMI is point-to-point, 7 links x ~153 GB/s per GPU, so a ring step
    is per-link bound and a few large messages beat DDP's 25 MiB default -- 544 MB of fp32 gradients = 9 buckets) and each bucket
    is ONE ``all_reduce`` (RCCL on GPUs, gloo in the CPU tests).  ``metrics`` (fp32 [5], ``training.metric_message``) rides at the
    tail of the LAST bucket instead of five collectives of 4 bytes; it is returned averaged.  No collective when world == 1.
    """
    if world <= 1:
        return metrics
    flat_sizes = [g.numel() for g in grads]
    dev = grads[0].device if grads else metrics.device
    per_bucket = max(1, bucket_bytes // 4)
    buckets, cur, cur_n = [], [], 0
    for i, n in enumerate(flat_sizes):
        if cur and cur_n + n > per_bucket:
            buckets.append(cur)
            cur, cur_n = [], 0
        cur.append(i)
        cur_n += n
    if cur or not buckets:
        buckets.append(cur)
    out_metrics = None
    for bi, idx in enumerate(buckets):
        parts = [grads[i].reshape(-1) for i in idx]
        tail = bi == len(buckets) - 1 and metrics is not None
        if tail:
            parts.append(metrics.reshape(-1).to(torch.float32))
        flat = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.float32, device=dev)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= world
        off = 0
        for i in idx:
            n = flat_sizes[i]
            grads[i].copy_(flat[off: off + n].view_as(grads[i]))
            off += n
        if tail:
            out_metrics = flat[off: off + metrics.numel()].clone()
    return out_metrics


def allreduce_flat(flat, world, bucket_bytes=64 << 20, tail=0):
    """Mean over ranks of ONE flat fp32 buffer, in place: the trainer keeps its gradients contiguous (didi_amd/trainer.py), so the
    buckets are slices -- no packing copies -- and each is one ``all_reduce``.  The last ``tail`` floats (the five logged scalars
    of ``log_dict(sync_dist=True)``, src/diffusion/model.py:241-242) always travel with the last bucket, never alone."""
    if world <= 1:
        return flat
    assert flat.dtype == torch.float32 and flat.is_contiguous() and flat.dim() == 1
    per = max(1, bucket_bytes // 4)
    n = flat.numel()
    start = 0
    while start < n:
        end = n if n - start <= per + tail else start + per
        dist.all_reduce(flat[start:end], op=dist.ReduceOp.SUM)
        start = end
    flat /= world
    return flat
