#!/usr/bin/env python3
"""Benchmark of the DiDi sampler hot path on MI355X (contract: see the task's bench.py section).

One "step" = one full `sample()` over one synthetic CommonSense-shaped batch: context encoder +
adapter, hoisted cross-K/V, T denoiser calls with the fused DDPM update, rounding argmax, eos
lengths and (N > 1) the RCCL gather of token ids.  Default workload = BASELINE.json configs[1]:
blenderbot-400M-distill-sized encoder + BertConfig-default decoder, T = 2000, step_freq = 1,
batch 256 per GPU, context/target 64.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision fp16-sched|bf16x3|...] [--mode ddpm|euler] [--freq F]

N > 1: either under a launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: RANK / WORLD_SIZE come
from the environment) or bare (`python bench.py --gpus N`: this process spawns the N ranks itself, see spawn_ranks).

Prints ONE JSON line on rank 0: metric/value (sampled dialogs/s, whole job), plus
  roofline     : the dominant kernel (MFMA GEMM), algorithmic FLOPs / launch duration measured with
                 HIP events around every GEMM launch of one denoiser call inside the timed region
  cpu_baseline : the CPU oracle (port of the reference path, fp32 torch eager) timed on a bounded
                 sample of the same workload on the host cores (rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


class GemmProf:
    """HIP events for the engine's per-GEMM timing hook (didi_sample_args.prof_*): one block of events per profiled call."""

    def __init__(self, calls, per_call=256):
        from didi_amd._lib import check, lib  # events are created inside libdidi_hip's HIP runtime

        self.lib, self.check = lib, check
        self.per_call, self.capacity = per_call, per_call * len(calls)
        self.calls = (C.c_int * len(calls))(*calls)
        self.used = (C.c_int * len(calls))()
        self.events = (C.c_void_p * self.capacity)()
        for i in range(self.capacity):
            ev = C.c_void_p()
            check(lib.didi_event_create(C.byref(ev)), "event_create")
            self.events[i] = ev

    def durations_ms(self, j=0):
        """Durations of the GEMM launches of profiled call j, in launch order."""
        out = []
        for i in range(0, self.used[j], 2):
            ms = C.c_float()
            a, b = self.events[j * self.per_call + i], self.events[j * self.per_call + i + 1]
            self.check(self.lib.didi_event_elapsed_ms(a, b, C.byref(ms)), "event_elapsed")
            out.append(ms.value)
        return out

    def close(self):
        for ev in self.events:
            self.lib.didi_event_destroy(ev)


def gemm_flops_per_call(cfg, M, attn_self=0.0, attn_cross=0.0):
    """Algorithmic FLOPs (2*M*N*K, NOT x3 for the split-bf16 passes) of the 6 decoder GEMMs x L, launch order.  ``attn_self`` /
    ``attn_cross``: algorithmic attention FLOPs that ride in the qkv / cross_q launch when the engine fuses them (fused_attention_flops)."""
    H, I = cfg.dec_dim, cfg.dec_ffn
    per_layer = [(3 * H, H, attn_self), (H, H, 0.0), (H, H, attn_cross), (H, H, 0.0), (I, H, 0.0), (H, I, 0.0)]
    return [2.0 * M * n * k + extra for _ in range(cfg.dec_layers) for n, k, extra in per_layer]


def fused_attention_flops(cfg, B, S, ctx_len, precision_name, ctx_mask):
    """(self, cross) algorithmic attention FLOPs per layer that the engine runs INSIDE the qkv / cross_q GEMM launch for this shape
    and precision (DIDI_OPT_FUSED_ATTN; 0.0 where it runs didi_attention as a launch of its own, which the GEMM events do not
    cover): QK^T and PV over the causal half (self) / over each dialog's visible context keys (cross)."""
    from didi_amd import _lib, ops

    opt = C.c_int()
    _lib.check(_lib.lib.didi_get_option(_lib.OPTIONS["FUSED_ATTN"], C.byref(opt)), "get_option")
    H, heads = cfg.dec_dim, cfg.dec_heads
    D, M, prec = H // heads, B * S, ops.PRECISIONS[precision_name]
    f_self = f_cross = 0.0
    if (opt.value & 1) and _lib.lib.didi_qkv_self_attention_supported(M, H, heads, S, H, prec):
        f_self = 2 * 2.0 * B * heads * D * (S * (S + 1) / 2)
    if (opt.value & 2) and _lib.lib.didi_q_cross_attention_supported(M, H, heads, S, ctx_len, H, prec):
        f_cross = 2 * 2.0 * heads * S * D * float(ctx_mask.sum())
    return f_self, f_cross



def host_cpu():
    """(model string, physical cores visible) from /proc/cpuinfo."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return model, len(cores) or os.cpu_count()


def cpu_baseline(cfg, sd, ids, mask, S, n_calls_total, slice_dialogs=32, slice_calls=20, full_calls=2):
    """Time the CPU oracle on a bounded sample of the same workload (>= 20 consecutive denoiser calls
    after warm-up, scaled by the call count, plus the measured one-off encoder and rounding):
      (a) `slice_calls` calls on the first `slice_dialogs` dialogs of the batch (M = 2048 rows: the CPU GEMMs are already
          compute-bound there) -> seconds per dialog per call;
      (b) `full_calls` calls on the full batch as a cross-check of the per-dialog rate at full size.
    The baseline uses the FASTER per-dialog rate of the two (the one that favours the CPU)."""
    from golden_util import config_to_oracle_config
    from oracle.model import OracleDiDi

    threads = torch.get_num_threads()
    cpu_model, phys = host_cpu()
    oracle = OracleDiDi(sd, config_to_oracle_config(cfg))
    B = ids.shape[0]
    Bs = min(slice_dialogs, B)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, S, cfg.dec_dim, generator=g)
    ones = torch.ones(B, 1, dtype=torch.long)

    def run_calls(nb, n):
        xx, ctx_b, mask_b, ones_b = x[:nb], ctx[:nb], mask[:nb], ones[:nb]
        oracle(encoder_attention_mask=mask_b, decoder_inputs_embeds=xx, time_ids=ones_b, context=ctx_b)  # warm-up call
        t0 = time.perf_counter()
        for k in range(n):
            xx, _ = oracle(encoder_attention_mask=mask_b, decoder_inputs_embeds=xx, time_ids=(k + 2) * ones_b, context=ctx_b)
        return (time.perf_counter() - t0) / n, xx

    with torch.no_grad():
        t0 = time.perf_counter()
        ctx = oracle.encode_context(ids, mask)
        t_enc = time.perf_counter() - t0
        t_slice, _ = run_calls(Bs, slice_calls)
        t_full, x0 = run_calls(B, full_calls)
        t0 = time.perf_counter()
        oracle.classifier(x0).argmax(-1)
        t_round = time.perf_counter() - t0
    per_dialog_call = min(t_slice / Bs, t_full / B)
    total = t_enc + n_calls_total * per_dialog_call * B + t_round
    return {
        "value": B / total, "unit": "dialogs/s", "cores": threads, "kind": "port",
        "cpu_model": cpu_model, "physical_cores_visible": phys,
        "sample": (f"CPU oracle (fp32 torch eager, {threads} threads on '{cpu_model}') on the batch of {B}: encoder once ({t_enc:.2f}s) + "
                   f"{slice_calls} consecutive denoiser calls on a {Bs}-dialog slice ({t_slice:.3f}s each = {1e3 * t_slice / Bs:.2f} ms per dialog) "
                   f"and {full_calls} on the full batch ({t_full:.2f}s each = {1e3 * t_full / B:.2f} ms per dialog) + rounding ({t_round:.2f}s); "
                   f"per-batch time = encoder + {n_calls_total} calls x {B} dialogs x the faster per-dialog rate + rounding = {total:.0f}s"),
    }


def main_training(args):
    """BASELINE configs[4]: 'Training step (noised-target fwd+bwd, MSE+rounding loss) DDP over 8 GPUs with RCCL gradient
    all-reduce' -- one step = DiDi.training_step + backward + bucketed gradient all-reduce (the five logged scalars folded in)
    + clip + AdamW on a fresh synthetic batch of 256 dialogs per GPU (configs/train.yaml:9)."""
    from didi_amd import dist as ddist
    from didi_amd.config import DiDiConfig
    from didi_amd.synthetic import RawContext, synthetic_batch, synthetic_state_dict
    from didi_amd.trainer import DiDiTrainer

    rank, world, local_rank = ddist.init_from_env("cuda")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dev = torch.device("cuda", local_rank)
    cfg = DiDiConfig(diffusion_steps=args.diffusion_steps, **({"vocab": args.vocab} if args.vocab else {}))
    sd = synthetic_state_dict(cfg, seed=1234)
    tr = DiDiTrainer(sd, cfg, device=dev, precision=args.train_precision, hidden_dropout=0.1, attention_dropout=0.1)  # train.yaml:22-23
    B, S = args.batch, args.seq
    n_batches = 4
    batches = []
    for i in range(n_batches):  # per-rank shards of distinct global batches, resident in HBM before the timed region
        ids, mask = synthetic_batch(B * world, S, cfg.vocab, seed=0xBADB01 + i)
        tids, tmask = synthetic_batch(B * world, S, cfg.vocab, seed=0xBADB01 + 100 + i)
        lo = rank * B
        batches.append((RawContext(ids[lo:lo + B].to(dev), mask[lo:lo + B].to(dev)), RawContext(tids[lo:lo + B].to(dev), tmask[lo:lo + B].to(dev))))
    torch.manual_seed(0xBADB01 + rank)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def step(k):
        c, t = batches[k % n_batches]
        return tr.training_step(c, t, world=world)

    losses = []
    for k in range(max(args.warmup, 1 if world > 1 else 0)):
        losses.append(step(k))
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        losses.append(step(k))
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()
    vals = [float(o["loss"]) for o in losses]
    assert all(v == v and abs(v) < 1e6 for v in vals), f"non-finite loss: {vals}"
    assert all(bool(torch.isfinite(p_).all()) for p_ in tr.params.values()), "non-finite parameters after the timed steps"
    # phase breakdown of one more (untimed) step
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    c, t = batches[0]
    ev[0].record()
    out = tr.forward_backward(c, t)
    ev[1].record()
    tr.optimizer_step(out["grads"])
    ev[2].record()
    torch.cuda.synchronize()
    M, H, I, L, V = B * S, cfg.dec_dim, cfg.dec_ffn, cfg.dec_layers, cfg.vocab
    fwd = L * 2.0 * M * (3 * H * H + 3 * H * H + 2 * H * I) + L * 2.0 * (B * S) * 2 * H * H + 2.0 * M * V * H  # + ckv on context rows
    gemm_flops = 3.0 * fwd  # forward + dgrad + wgrad
    n_params = sum(p_.numel() for p_ in tr.params.values())
    value = B * world * args.steps / elapsed
    res = {
        "metric": "training dialogs/sec (fwd + bwd + gradient all-reduce + AdamW, whole job)", "value": round(value, 2), "unit": "dialogs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 1),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"bf16x3": "bf16", "fp16x2": "fp16"}.get(args.train_precision, args.train_precision), "data": "synthetic",
        "config": {
            "workload": (f"BASELINE configs[4] (training step): blenderbot-400M-distill-sized frozen encoder + BertConfig-default decoder "
                         f"(12x768/12h/3072), V={cfg.vocab}, T={cfg.diffusion_steps}, batch {B}/GPU, context {S}, target {S}, AdamW + linear LR + clip 1.0, "
                         f"decoder dropout 0.1 / 0.1 (configs/train.yaml:22-23)"),
            "global_batch": B * world, "seq_len": S, "parallelism": f"dp{world} (gradient all-reduce in 57 MB buckets launched under the backward; the 5 metric scalars in the buffer's tail)",
            "trainable_parameters": n_params, "precision": args.train_precision,
            "world_size_seen": torch.distributed.get_world_size() if world > 1 else 1,
            "backend": torch.distributed.get_backend() if world > 1 else None,
            "losses_first_last": [round(vals[0], 4), round(vals[-1], 4)],
            "phase_ms": {"forward_backward": round(ev[0].elapsed_time(ev[1]), 2), "clip_adamw_resplit": round(ev[1].elapsed_time(ev[2]), 2)},
        },
        "roofline": {"bound": "mfma", "kernel": "all GEMMs of the step (forward, dgrad, wgrad): whole-step figure, not a per-kernel one",
                     "achieved": round(gemm_flops / (elapsed / args.steps) / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(gemm_flops / (elapsed / args.steps) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                     "note": "algorithmic GEMM FLOPs of one step (3 x forward) / the whole step time, attention, LayerNorm, dropout, the operand "
                             "splits and the optimiser included; bf16x3 executes three MFMA products per algorithmic one "
                             "(per kernel: DESIGN.md section 6, profiles/r03_train_kernel_stats_v7.csv)"},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_training(cfg, sd, batches[0], tr)
        res["gpu_over_cpu"] = round(value / res["cpu_baseline"]["value"], 1)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def cpu_baseline_training(cfg, sd, batch, tr, slice_dialogs=64, passes=3):
    """The CPU oracle's training step (fp32 torch eager forward + autograd backward, oracle/training.py) on the first
    `slice_dialogs` dialogs of the first batch (M = 4096 rows: the CPU GEMMs are compute-bound there), one warm-up + `passes` timed
    passes, the fastest kept; the optimiser step is NOT timed (both choices favour the CPU)."""
    from golden_util import config_to_oracle_config
    from oracle.training import training_step_with_grads

    from didi_amd.synthetic import RawContext

    threads = torch.get_num_threads()
    cpu_model, phys = host_cpu()
    sd = dict(sd)
    sd.setdefault("sigmas", tr.sigmas.detach().cpu())
    sd.setdefault("std_0", torch.tensor(tr.std_0))
    if "classifier.weight" not in sd:
        sd["classifier.weight"] = sd["emb.weight"]
    n = slice_dialogs
    ctx = RawContext(batch[0].input_ids[:n].cpu(), batch[0].attention_mask[:n].cpu())
    tgt = RawContext(batch[1].input_ids[:n].cpu(), batch[1].attention_mask[:n].cpu())
    S, H = tgt.input_ids.shape[1], cfg.dec_dim
    g = torch.Generator().manual_seed(3)
    draws = [torch.randn(n, S, H, generator=g), torch.randn(n, S, H, generator=g)]
    t = torch.randint(1, cfg.diffusion_steps + 1, (n, 1), generator=g)
    ocfg = config_to_oracle_config(cfg)
    times = []
    for _ in range(1 + passes):
        t0 = time.perf_counter()
        training_step_with_grads(sd, ocfg, ctx, tgt, draws, t)
        times.append(time.perf_counter() - t0)
    return {"value": round(n / min(times[1:]), 3), "unit": "dialogs/s", "cores": threads, "kind": "port", "cpu_model": cpu_model,
            "physical_cores_visible": phys,
            "sample": (f"CPU oracle training step (fp32 torch eager forward + autograd backward, no optimiser step) on a {n}-dialog slice of the "
                       f"batch, {threads} threads on '{cpu_model}': warm-up {times[0]:.2f}s, timed passes {', '.join(f'{x:.2f}s' for x in times[1:])} (fastest kept)")}


def spawn_ranks(n):
    """`python bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): this process becomes the launcher.
    It has not touched the GPU (torch is imported, no device call is made, didi_amd._lib is not loaded) and never will: it starts
    N fresh children of this same command line, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as
    torch.distributed.run would, lets rank 0's single JSON line through on stdout, and exits non-zero if any rank fails (the
    survivors are stopped: their collectives would wait for ever).  What Lightning does for the reference's training
    (`Trainer(devices=8, strategy=DDPStrategy)`, src/training.py:37-47)."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, text=True,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))

    def relay():  # stdout carries the JSON line and nothing else: library chatter on rank 0's stdout (gloo prints there) goes to stderr
        for line in procs[0].stdout:
            print(line, end="", file=sys.stdout if line.lstrip().startswith("{") else sys.stderr, flush=True)

    import threading

    pump = threading.Thread(target=relay, daemon=True)
    pump.start()
    rc, left = 0, set(range(n))
    while left:
        for r in sorted(left):
            code = procs[r].poll()
            if code is None:
                continue
            left.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for q in left:
                    procs[q].terminate()
        time.sleep(0.2)
    pump.join(timeout=30)
    return rc


def main_rehearsal(args):
    """`--device cpu`: the launch path of a multi-rank run (rank spawn, rendezvous, shard bounds, per-rank Philox ranges, the id
    gather, barrier + max-over-ranks timing, one JSON line from rank 0) over gloo with a STUB in place of the sampler -- no kernel
    runs and the line says so ("valid": false).  tests/test_dist_cpu.py runs it with --gpus 2; it exists so that the N > 1 command
    the driver will use is exercised where there is one GPU or none."""
    from didi_amd import dist as ddist
    from didi_amd.synthetic import RawContext, synthetic_batch

    rank, world, _ = ddist.init_from_env("cpu")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    B_total, S = args.batch * world, args.seq
    ids, mask = synthetic_batch(B_total, S, 8008, seed=0xBADB01)
    lo, hi = ddist.shard_bounds(B_total, world, rank)
    ctx = RawContext(ids[lo:hi], mask[lo:hi])

    def stub(local, model, mode, step_freq, philox_offset=0, max_len=-1, **kw):  # ids say which rank's counter range produced them
        b = local.input_ids.shape[0]
        return torch.full((b, S), philox_offset >> 40, dtype=torch.int64), torch.full((b,), S, dtype=torch.int32)

    def step():
        return ddist.sample_sharded(ctx, None, args.mode, args.freq, rank=rank, world=world, max_len=S, eos_id=2, global_batch=B_total,
                                    sample_fn=stub)

    def sync():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_ids, _ = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()
    owners = [int(out_ids[ddist.shard_bounds(B_total, world, r)[0], 0]) for r in range(world)]
    assert tuple(out_ids.shape) == (B_total, S) and owners == list(range(world)), owners
    if rank == 0:
        print(json.dumps({"metric": "sampled dialogs/sec (whole node, T steps)", "value": round(B_total * args.steps / elapsed, 3), "unit": "dialogs/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "synthetic", "valid": False,
                          "config": {"workload": "launch-path rehearsal on CPU: stub sampler, no kernels (--device cpu)", "global_batch": B_total,
                                     "seq_len": S, "parallelism": f"dp{world} (batch shard, id all-gather)",
                                     "world_size_seen": torch.distributed.get_world_size() if world > 1 else 1,
                                     "backend": torch.distributed.get_backend() if world > 1 else None}}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default="fp16-sched", choices=["bf16", "fp16", "fp16x2", "fp16-sched", "fp16x3", "bf16x3"],
                    help="fp16-sched (default): fp16 planes with the precision schedule of didi_amd.model.DEFAULT_SCHEDULE (ddpm: one MFMA product, "
                         "cycling through 4 differently rounded weight copies, for t > 0.15 T; the weights' remainder for t > 0.064 T; all three "
                         "products for the last steps and at least the last 64 calls; euler: two products, then three): the cheapest "
                         "configuration that holds ids >= 99.9 %% and x_t <= 1e-3 over the 2000-call loop at THIS model size; fp16 / fp16x2: "
                         "one / two products throughout (miss the bars at 2000 calls); fp16x3 / bf16x3: fp32-class, three products")
    ap.add_argument("--mode", default="ddpm", choices=["ddpm", "euler"])
    ap.add_argument("--freq", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="dialogs per GPU")
    ap.add_argument("--seq", type=int, default=64, help="context length = target length")
    ap.add_argument("--diffusion-steps", type=int, default=2000)
    ap.add_argument("--vocab", type=int, default=None, help="vocabulary size (BASELINE config 4: 50000)")
    ap.add_argument("--clamp", action="store_true", help="config 4's per-step clamp x0 <- emb[argmax(classifier(x0))] (not in the reference)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-precision", default="bf16x3", choices=["bf16x3", "fp16x2", "fp16", "bf16"],
                    help="--preset configs4 only: GEMM precision of the training step's forward and backward")
    ap.add_argument("--preset", default=None, choices=["configs1", "configs2", "configs3", "configs4"],
                    help="BASELINE.json configs[i]: configs1 = the default (ddpm, f=1, 256/GPU); configs2 = 'DDIM 200-step, batch 4096 over "
                         "8 GPUs' = euler, step_freq 10, 512 dialogs/GPU; configs3 = V=50000, seq 84, batch 1024, per-step clamp")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                    help="cpu: rehearsal of the multi-rank launch path over gloo with a stub sampler (no kernels, the line is marked invalid)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # no launcher: start the ranks ourselves, before anything touches the GPU
        return spawn_ranks(args.gpus)
    if args.device == "cpu":
        return main_rehearsal(args)
    if args.preset == "configs4":  # BASELINE configs[4]: the training step (secondary line; the headline metric is the sampler's)
        return main_training(args)
    workload = "BASELINE configs[1]"
    if args.preset == "configs2":
        args.mode, args.freq, args.batch = "euler", 10, 512
        workload = "BASELINE configs[2] (DDIM = euler with s_churn 0, 200 + 1 calls, 512 dialogs per GPU = 4096 on 8)"
    elif args.preset == "configs3":
        args.vocab, args.seq, args.batch, args.clamp = 50000, 84, 1024, True
        workload = "BASELINE configs[3] (large-vocab clamp stress)"
    elif (args.mode, args.freq, args.batch, args.seq, args.diffusion_steps, args.vocab, args.clamp) != ("ddpm", 1, 256, 64, 2000, None, False):
        workload = "custom (flags override BASELINE configs[1])"

    from didi_amd import dist as ddist
    from didi_amd.config import DiDiConfig
    from didi_amd.model import DiDiSampler
    from didi_amd.schedule import ddpm_plan, euler_plan
    from didi_amd.synthetic import RawContext, synthetic_batch, synthetic_state_dict

    rank, world, local_rank = ddist.init_from_env("cuda")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dev = torch.device("cuda", local_rank)

    cfg = DiDiConfig(diffusion_steps=args.diffusion_steps, **({"vocab": args.vocab} if args.vocab else {}))
    sd = synthetic_state_dict(cfg, seed=1234)
    if args.precision == "fp16x3":  # fp16 planes, all three products in every call
        model = DiDiSampler(sd, cfg, device=dev, precision="fp16x2", schedule=((0, "fp16x3"),))
    else:
        model = DiDiSampler(sd, cfg, device=dev, precision=args.precision)
    B_total, S = args.batch * world, args.seq
    ids, mask = synthetic_batch(B_total, args.seq, cfg.vocab, seed=0xBADB01)  # the global batch, on the host
    lo, hi = ddist.shard_bounds(B_total, world, rank)
    ctx = RawContext(ids[lo:hi].to(dev), mask[lo:hi].to(dev))  # only this rank's shard is uploaded; resident in HBM before the timed region
    plan = (ddpm_plan(model._sigmas_host, cfg.diffusion_steps, args.freq) if args.mode == "ddpm" else
            euler_plan(model._sigmas_host, cfg.diffusion_steps, args.freq, 1, cfg.s_churn, cfg.s_tmin, cfg.s_tmax))
    n_calls = plan.n_calls
    # precision of every call (a precision schedule has up to three regimes); one representative call per regime is instrumented
    names = {v: k for k, v in __import__("didi_amd.ops", fromlist=["PRECISIONS"]).PRECISIONS.items()}
    per_call = model.call_precisions(plan.time_row, args.mode) or [model.prec] * n_calls
    regimes = []  # (precision name, number of calls, index of its middle call)
    for pv in dict.fromkeys(per_call):
        idx = [k for k, q in enumerate(per_call) if q == pv]
        regimes.append((names[pv], len(idx), idx[len(idx) // 2]))

    def step(prof=None):
        return ddist.sample_sharded(ctx, model, args.mode, args.freq, rank=rank, world=world, max_len=S, eos_id=2, global_batch=B_total,
                                    clamp=args.clamp, _prof=prof)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    torch.manual_seed(0xBADB01)
    if world > 1:  # RCCL builds its communicator on the first collective: keep that out of the timed region even with --warmup 0
        probe = torch.zeros((1, S + 1), dtype=torch.int32, device=dev)
        torch.distributed.all_gather_into_tensor(torch.empty((world, S + 1), dtype=torch.int32, device=dev), probe)
    t_prog = [time.perf_counter()]

    def progress(what, i, n):  # stderr only (stdout carries the one JSON line): a 25-step run is minutes of silence otherwise
        if rank == 0:
            now = time.perf_counter()
            print(f"[bench] {what} {i + 1}/{n}: {now - t_prog[0]:.2f} s", file=sys.stderr, flush=True)  # host time since the last line
            t_prog[0] = now

    for k in range(args.warmup):
        step()
        progress("warm-up step", k, args.warmup)
    prof = GemmProf(calls=sorted(r[2] for r in regimes))
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        out_ids, out_len = step(prof if k == 0 else None)
        progress("timed step", k, args.steps)  # after the step's own id read-back: no extra synchronisation
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()
    # validity of what was timed: device-side faults (LayerNorm exchange timeout, fp16 range), finite x0, sane ids
    model.check_fault("bench")
    x0 = next(iter(model._bufs.values()))["x0"]
    assert bool(torch.isfinite(x0).all()), "non-finite x0 after the timed loop"
    assert tuple(out_ids.shape) == (B_total, S) and int(out_len.min()) >= 1 and int(out_len.max()) <= S
    assert int(out_ids.min()) >= 0 and int(out_ids.max()) < cfg.vocab
    counts = out_ids.flatten().bincount(minlength=cfg.vocab)
    top_share, distinct = counts.max().item() / out_ids.numel(), int((counts > 0).sum())
    # random-init weights give near-uniform ids; a poisoned batch rounds to one token (NaN rows -> index 0)
    assert top_share < 0.5 and distinct >= min(64, cfg.vocab // 4), f"degenerate ids: top share {top_share:.3f}, {distinct} distinct"

    # roofline of the dominant kernel (MFMA GEMM) from the HIP events around every GEMM launch of one denoiser call per precision
    # regime, weighted by the number of calls each regime has in the loop
    M = args.batch * S
    flops = gemm_flops_per_call(cfg, M)
    fused = {}
    order = sorted(range(len(regimes)), key=lambda i: regimes[i][2])  # GemmProf holds the calls in increasing index
    roofline = None
    per_regime, w_time, w_flop, w_launch = {}, 0.0, 0.0, 0
    for slot, ri in enumerate(order):
        name, n_reg, call = regimes[ri]
        dur = prof.durations_ms(slot)
        # the attention a fused launch runs is algorithmic work of that launch (about 1 % of the layer's FLOPs); un-fused, it is a
        # launch of its own outside the GEMM events
        fa = fused_attention_flops(cfg, args.batch, S, args.seq, name, mask[lo:hi])
        flops = gemm_flops_per_call(cfg, M, *fa)
        fused[name] = {"self_attention_in_qkv_launch": fa[0] > 0, "cross_attention_in_cross_q_launch": fa[1] > 0}
        if len(dur) != len(flops):
            per_regime = None
            break
        mult = {"bf16x3": 3, "fp16x3": 3, "fp16x2": 2}.get(name, 1)
        tf = sum(flops) / (sum(dur) * 1e-3) / 1e12
        per_regime[name] = {
            "calls": n_reg, "profiled_call": call, "gemm_ms_per_call": round(sum(dur), 3), "avg_launch_us": round(1e3 * sum(dur) / len(dur), 2),
            "algorithmic_tflops": round(tf, 1), "mfma_products_per_product": mult, "executed_mfma_tflops": round(tf * mult, 1),
            "plain (qkv, cross_q, ffn_up) tflops": round(sum(flops[0::2]) / (sum(dur[0::2]) * 1e-3) / 1e12, 1),
            "with LayerNorm epilogue (attn_out, cross_out, ffn_down) tflops": round(sum(flops[1::2]) / (sum(dur[1::2]) * 1e-3) / 1e12, 1),
        }
        w_time += n_reg * sum(dur) * 1e-3
        w_flop += n_reg * sum(flops)
        w_launch += n_reg * len(dur)
    if per_regime:
        achieved = w_flop / w_time / 1e12
        # HBM (fabric-side) bytes per GEMM launch: PMC passes are separate runs per precision (profiles/README.md); a schedule's
        # figure is the average over its regimes weighted by their launch counts
        traffic, traffic_src = None, None
        if args.batch == 256 and args.seq == 64:
            paths = {name: os.path.join(ROOT, "profiles", f"gemm_traffic_{name}.json") for name in per_regime}
            if all(os.path.exists(q) for q in paths.values()):
                tot = sum(per_regime[name]["calls"] for name in per_regime)
                traffic = round(sum(per_regime[name]["calls"] * json.load(open(q))["hbm_bytes_per_launch"] for name, q in paths.items()) / tot)
                traffic_src = [os.path.relpath(q, ROOT) for q in paths.values()]
        roofline = {
            "bound": "mfma", "kernel": ("gemm_pp_kernel<NPASS, F16, LN, false> (256x192 tile, staggered wave groups; NPASS = MFMA products per "
                                         "algorithmic product: 1 fp16, 2 fp16x2, 3 fp16x3/bf16x3; LN = true on the three output projections per "
                                         "layer, which carry bias + residual + LayerNorm in the epilogue)"),
            "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "fused_into_gemm_launches": fused,
            "launches_per_call": len(flops), "avg_launch_us": round(1e6 * w_time / w_launch, 2),
            "avg_gflop_per_launch": round(sum(flops) / len(flops) / 1e9, 2),
            "gemm_share_of_step": round(w_time / (elapsed / args.steps), 3),
            "by_precision_regime": per_regime,
            "note": ("achieved = algorithmic FLOPs (2*M*N*K per launch, the extra MFMA products of fp16x2 / x3 NOT counted) of all GEMM launches of the "
                     "loop / their summed durations: HIP events around every GEMM launch of one denoiser call per precision regime inside the "
                     "timed region, weighted by the regime's number of calls.  Since round 3 the qkv launch also runs the causal self-attention "
                     "and the cross_q launch the cross-attention over the hoisted K / V (fused_into_gemm_launches; their ~1 % of FLOPs is counted), "
                     "and the three output projections run bias + residual + LayerNorm in their epilogues: the durations cover all of that, so "
                     "this figure is lower than a GEMM-only one would be while the step is faster (profiles/r03_ab_cross_fp16.log)"),
        }
    prof.close()

    value = B_total * args.steps / elapsed
    res = {
        "metric": "sampled dialogs/sec (whole node, T steps)", "value": round(value, 3), "unit": "dialogs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 1),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp16-sched": "fp16", "fp16x2": "fp16", "fp16x3": "fp16", "bf16x3": "bf16"}.get(args.precision, args.precision), "data": "synthetic",
        "config": {
            "workload": (f"{workload}: blenderbot-400M-distill-sized encoder (2x1280/32h/5120) + BertConfig-default "
                         f"decoder (12x768/12h/3072), V={cfg.vocab}, {args.mode} T={cfg.diffusion_steps} step_freq={args.freq} "
                         f"({n_calls} denoiser calls), batch {args.batch}/GPU, context {args.seq}, target {S}"
                         + (", per-step clamp extension ON" if args.clamp else "")),
            "global_batch": B_total, "seq_len": S, "denoiser_calls": n_calls, "parallelism": f"dp{world} (batch shard, id all-gather)",
            "weights": "random-init N(0,0.02), seed 1234", "noise": "in-kernel Philox",
            "precision": args.precision,
            "precision_schedule": ({"timestep_thresholds": [list(x) for x in (model.schedule_euler if args.mode == "euler" else model.schedule)],
                                    "tail_calls": [list(x) for x in model.tail_calls], "calls": {r[0]: r[1] for r in regimes},
                                    "weight_dither": model.weight_dither}
                                   if model.schedule else None),
            "precision_note": {
                "fp16-sched": ("fp16 operand planes where BASELINE configs[1] says bf16: 16-bit MFMA operands with fp32 accumulation either way; the "
                               "parity bar (ids >= 99.9 %, every x_t <= 1e-3 against the fp32 reference) decides what is multiplied: one-product "
                               "bf16 / fp16 miss it at this model size (weight rounding adds up coherently over the loop), so ddpm calls run one "
                               "product, cycling through 4 differently rounded copies of the weights, while t > 0.15 T, add the weights' "
                               "remainder plane below that and multiply all three products for the last 0.064 T steps: 0 of 1024 ids differ, "
                               "max x_t rel-L2 6.3e-4 over 2000 calls on two seed sets (profiles/r02_dither_probe_ddpm_f1*.log, "
                               "tests/test_gpu_fullsize.py); euler loops run two products, then three"),
                "fp16x2": "fp16 planes, A_hi*(W_hi+W_lo) in every call: holds the bars over 201 calls (step_freq 10), misses them over 2000",
                "fp16": "one-product fp16: misses the parity bars at this model size (tests/test_gpu_fullsize.py); not a valid headline mode",
                "bf16": "one-product bf16: misses the parity bars (99.46 % ids on the config-1 model); not a valid headline mode",
                "fp16x3": "fp16 planes, all three products in every call: fp32-class (max x_t rel-L2 2.9e-5 over 2000 calls, 0 of 1024 ids differ)",
                "bf16x3": "bf16 planes, all three products: fp32-class with fp32's exponent range (max x_t rel-L2 1.0e-4 over 2000 calls)",
            }.get(args.precision),
            "world_size_seen": torch.distributed.get_world_size() if world > 1 else 1,
            "backend": torch.distributed.get_backend() if world > 1 else None,
            "checks": {"device_faults": 0, "x0_finite": True, "ids_top_share": round(top_share, 4), "ids_distinct": distinct},
        },
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(cfg, sd, ids[: args.batch], mask[: args.batch], S, n_calls)
        res["gpu_over_cpu"] = round(value / res["cpu_baseline"]["value"], 1)
        res["cpu_baseline"]["value"] = round(res["cpu_baseline"]["value"], 5)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
