"""How fast is the vendor library (hipBLASLt through torch.matmul, fp16, bias only) on the decoder's GEMM shapes?
A yardstick for gemm_pp_kernel, not a code path: nothing in didi_amd calls it."""
import torch

M = 16384
for name, N, K in [("qkv", 2304, 768), ("proj", 768, 768), ("ffn_up", 3072, 768), ("ffn_down", 768, 3072), ("square 8192", 8192, 8192)]:
    m = 8192 if name.startswith("square") else M
    a = torch.randn(m, K, device="cuda", dtype=torch.float16)
    w = torch.randn(N, K, device="cuda", dtype=torch.float16) * 0.02
    b = torch.randn(N, device="cuda", dtype=torch.float16)
    for _ in range(5):
        torch.nn.functional.linear(a, w, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 30
    e0.record()
    for _ in range(iters):
        torch.nn.functional.linear(a, w, b)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name:12s} M={m} N={N:5d} K={K:5d}  {us:8.1f} us  {2 * m * N * K / us / 1e6:7.1f} TF", flush=True)
