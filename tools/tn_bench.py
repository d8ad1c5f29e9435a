"""NT (didi_gemm on a transposed copy of W) against TN = 2 (didi_gemm_dgrad on W as it lies) on the same products, per shape.
Usage: python tools/tn_bench.py [precision]"""
import sys
import torch
from didi_amd import ops

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
pd = ops.plane_dtype(prec)
g = torch.Generator().manual_seed(1)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for M, N, K in [(16384, 768, 3072), (16384, 768, 768), (16384, 3072, 768), (16384, 2304, 768)]:
    dc = torch.randn(M, N, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).cuda()
    d_hi, d_lo = ops.split(dc, dtype=pd)
    w_hi, w_lo = ops.split(w, dtype=pd)
    wt_hi, wt_lo = ops.transpose_split(w, dtype=pd)
    t_nt = timed(lambda: ops.gemm(d_hi, d_lo, wt_hi, wt_lo, precision=prec, out_f32=True, out_planes=False))
    t_tn = timed(lambda: ops.gemm_dgrad(d_hi, d_lo, w_hi, w_lo, precision=prec))
    fl = 2.0 * M * N * K * (3 if prec.endswith("x3") else 2)
    print(f"{prec} M={M} contraction={N} out={K}: NT {t_nt:7.1f} us ({fl / t_nt / 1e6:6.0f} TF executed)   TN2 {t_tn:7.1f} us ({fl / t_tn / 1e6:6.0f} TF)   ratio {t_nt / t_tn:.3f}")
