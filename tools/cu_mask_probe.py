"""Experiment: one GPU's batch as two halves on two HIP streams with disjoint CU masks, so that one half's
HBM-bound phases (epilogues, attention) run under the other half's MFMA loops.  Usage: cu_mask_probe.py [T] [layout]"""
import ctypes as C
import sys
import threading
import time

import torch

from didi_amd import _lib
from didi_amd.config import DiDiConfig
from didi_amd.model import DiDiSampler
from didi_amd.sampling import sample
from didi_amd.synthetic import RawContext, synthetic_batch, synthetic_state_dict

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
layout = sys.argv[2] if len(sys.argv) > 2 else "split"


def _loaded_hip_runtime():
    """The HIP runtime torch (and libdidi_hip) already run on: a second copy loaded by name would not share their streams."""
    for line in open("/proc/self/maps"):
        path = line.split()[-1]
        if "libamdhip64" in path:
            return C.CDLL(path)
    raise RuntimeError("no libamdhip64 mapped into this process")


torch.cuda.init()
hip = _loaded_hip_runtime()


def masked_stream(words):
    st = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), len(words), arr)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(st.value)


# two samplers share the GPU here: the fused projection + LayerNorm needs the device to itself (its band mates must be co-resident)
_lib.set_option("GEMM_LN", 0)
cfg = DiDiConfig(diffusion_steps=T)
sd = synthetic_state_dict(cfg, seed=1234)
ids, mask = synthetic_batch(256, 64, cfg.vocab, seed=3)
torch.cuda.set_device(0)


def run(models, ctxs, streams):
    def one(m, c, s):
        with torch.cuda.stream(s):
            sample(c, m, "ddpm", 1, return_ids=True)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        th = [threading.Thread(target=one, args=a) for a in zip(models, ctxs, streams)]
        [t.start() for t in th]
        [t.join() for t in th]
        torch.cuda.synchronize()
        dt = time.time() - t0
    return dt


m0 = DiDiSampler(sd, cfg, device="cuda:0", precision="fp16")
full = RawContext(ids.cuda(), mask.cuda())
dt = run([m0], [full], [torch.cuda.current_stream()])
print(f"one stream, B=256: {dt * 1e3:.1f} ms  -> {256 / dt * T / 2000:.2f} dialogs/s at T=2000", flush=True)
m1 = DiDiSampler(sd, cfg, device="cuda:0", precision="fp16")
halves = [RawContext(ids[:128].cuda(), mask[:128].cuda()), RawContext(ids[128:].cuda(), mask[128:].cuda())]
if layout == "split":      # CUs 0-127 / 128-255 in the mask's bit order
    masks = [[0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4]
elif layout == "even":     # alternate bits
    masks = [[0x55555555] * 8, [0xAAAAAAAA] * 8]
elif layout == "none":
    masks = None
if masks is None:
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
else:
    streams = [masked_stream(w) for w in masks]
dt = run([m0, m1], halves, streams)
print(f"two streams ({layout}), 2 x B=128: {dt * 1e3:.1f} ms  -> {256 / dt * T / 2000:.2f} dialogs/s at T=2000", flush=True)
dt = run([m0], [halves[0]], [streams[0]])
print(f"one masked stream alone, B=128: {dt * 1e3:.1f} ms", flush=True)
