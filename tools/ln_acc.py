"""Accuracy of didi_gemm_ln vs didi_gemm + didi_layernorm against an fp64 reference (same rounded operands)."""
import torch
from didi_amd import ops

def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm()).item()

torch.manual_seed(0)
M, N = 4096, 768
for K in (768, 3072):
    for outlier in (0.0, 30.0):
        a, w = torch.randn(M, K), torch.randn(N, K) * 0.05
        bias, r = torch.randn(N), torch.randn(M, N)
        r[:, 5] += outlier
        g, b = 1 + 0.1 * torch.randn(N), 0.1 * torch.randn(N)
        for prec in ("bf16x3", "fp16"):
            pd = ops.plane_dtype(prec)
            a_hi, a_lo = ops.split(a.cuda(), dtype=pd)
            w_hi, w_lo = ops.split(w.cuda(), dtype=pd)
            rp = ops.split(r.cuda(), dtype=pd)
            ru = ops.join(*rp).cpu().double()
            if prec == "bf16x3":
                ar, wr = ops.join(a_hi, a_lo).cpu().double(), ops.join(w_hi, w_lo).cpu().double()
            else:
                ar, wr = a_hi.cpu().double(), w_hi.cpu().double()
            pre = ar @ wr.T + bias.double() + ru
            want = torch.nn.functional.layer_norm(pre, (N,), g.double(), b.double(), 1e-12)
            _, _, yf = ops.gemm_ln(a_hi, a_lo, w_hi, w_lo, bias.cuda(), rp, g.cuda(), b.cuda(), 1e-12, precision=prec, f32=True)
            tmp = ops.gemm(a_hi, a_lo, w_hi, w_lo, bias.cuda(), precision=prec, residual_planes=rp, out_f32=True, out_planes=False)[2]
            _, _, yu = ops.layernorm(tmp, g.cuda(), b.cuda(), 1e-12, dtype=pd, f32=True)
            print(f"K={K} outlier={outlier} {prec}: fused {rel(yf, want):.3e}  unfused {rel(yu, want):.3e}  pre-norm gemm {rel(tmp, pre):.3e}")
