#!/usr/bin/env python3
"""Micro-benchmark of didi_gemm on the decoder's shapes (M = B*S rows), both precisions.

    python tools/gemm_bench.py [--M 16384] [--iters 30]

Times each (shape, epilogue) as the decoder issues it, with HIP events around `iters`
back-to-back launches on random data (cdna_hip_programming.md rule 25), and prints algorithmic
TFLOP/s (2*M*N*K; the bf16x3 passes are not counted) and the executed-MFMA TFLOP/s.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from didi_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=16384)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--H", type=int, default=768)
    ap.add_argument("--I", type=int, default=3072)
    ap.add_argument("--precisions", default="fp16,fp16x2,bf16x3")
    args = ap.parse_args()
    precs = args.precisions.split(",")
    MULT = {"bf16x3": 3, "fp16x3": 3, "fp16x2": 2}
    M, H, I = args.M, args.H, args.I
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
    shapes = [
        ("qkv      planes", 3 * H, H, dict()),
        ("attn_out f32+res", H, H, dict(resid=True)),
        ("cross_q  planes", H, H, dict()),
        ("ffn_up   gelu", I, H, dict(act="gelu")),
        ("ffn_down f32+res", H, I, dict(resid=True)),
    ]
    total = {p: 0.0 for p in precs}
    for name, N, K, kw in shapes:
        a32, w32, r32 = rnd(M, K), rnd(N, K) * 0.02, rnd(M, N)
        bias = rnd(N)
        for prec in precs:
            pd = ops.plane_dtype(prec)
            a_hi, a_lo = ops.split(a32, dtype=pd)
            w_hi, w_lo = ops.split(w32, dtype=pd)
            r = ops.split(r32, dtype=pd) if kw.get("resid") else None
            call = lambda: ops.gemm(a_hi, a_lo, w_hi, w_lo, bias=bias, act=kw.get("act"), precision=prec,  # noqa: E731
                                    residual_planes=r, out_f32=bool(r), out_planes=not r, out_lo=prec == "bf16x3")
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                call()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            fl = 2.0 * M * N * K
            mult = MULT.get(prec, 1)
            total[prec] += us * (2 if name.startswith('attn_out') else 1)
            print(f"{name:18s} N={N:5d} K={K:5d} {prec:7s} {us:8.1f} us  alg {fl / us / 1e6:7.1f} TF  exec {mult * fl / us / 1e6:7.1f} TF")
    # output projection + residual + LayerNorm: fused launch vs GEMM + LayerNorm
    if H == 768:
        gam, bet = torch.ones(H, device="cuda"), torch.zeros(H, device="cuda")
        for prec in precs:
            pd = ops.plane_dtype(prec)
            for K in (H, I):
                a_hi, a_lo = ops.split(rnd(M, K), dtype=pd)
                w_hi, w_lo = ops.split(rnd(H, K) * 0.02, dtype=pd)
                r = ops.split(rnd(M, H), dtype=pd)
                bias = rnd(H)

                def unfused():
                    _, _, t = ops.gemm(a_hi, a_lo, w_hi, w_lo, bias=bias, precision=prec, residual_planes=r, out_f32=True, out_planes=False)
                    ops.layernorm(t, gam, bet, 1e-12, dtype=pd)

                for name, call in (("gemm_ln fused", lambda: ops.gemm_ln(a_hi, a_lo, w_hi, w_lo, bias, r, gam, bet, 1e-12, precision=prec)),
                                   ("gemm + layernorm", unfused)):
                    for _ in range(3):
                        call()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.iters):
                        call()
                    e1.record()
                    torch.cuda.synchronize()
                    print(f"{name:18s} N={H:5d} K={K:5d} {prec:7s} {e0.elapsed_time(e1) * 1e3 / args.iters:8.1f} us")
    layer_fl = 2.0 * M * (3 * H * H + 3 * H * H + 2 * H * I)
    for prec, us in total.items():
        print(f"sum of 6 GEMMs per layer ({prec}): {us + 0:8.1f} us (cross_out counted as attn_out) -> alg {layer_fl / (us) / 1e6:7.1f} TF")


if __name__ == "__main__":
    main()
