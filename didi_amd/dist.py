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


def sample_sharded(raw_context, model, mode, step_freq, *, rank, world, noise_factor=1, max_len=-1, eos_id=None, **kw):
    """Every rank passes the SAME global batch; each samples its shard and all ranks get all ids."""
    from .sampling import sample
    from .synthetic import RawContext

    B = raw_context.input_ids.shape[0]
    a, b = shard_bounds(B, world, rank)
    local = RawContext(raw_context.input_ids[a:b].contiguous(), raw_context.attention_mask[a:b].contiguous())

    class _Eos:
        eos_token_id, sep_token_id = eos_id, None

    ids, lengths = sample(local, model, mode, step_freq, noise_factor=noise_factor, max_len=max_len,
                          tokenizer=_Eos() if eos_id is not None else None, return_ids=True,
                          philox_offset=philox_offset(rank), **kw)
    return gather_ids(ids, lengths, B, world)
