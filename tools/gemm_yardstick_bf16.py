"""Yardstick only (never used by the product path): hipBLASLt bf16 GEMM (torch.addmm on bf16 tensors, fp32 accumulate) vs
conv_bf16x on the token-GEMM shapes of the e2fgvi_hq 720x1296 T=10 forward, and the fp32 shapes of the headline against the
library's fp32 GEMM.   python tools/gemm_yardstick_bf16.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, M, K, N in (("qkv", 66240, 512, 1536), ("proj", 64800, 512, 512), ("fc1", 64800, 512, 1960), ("fc2", 64800, 1960, 512),
                      ("sc", 64800, 512, 6272)):
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05); b = torch.randn(N, device=dev)
    wt = w.bfloat16().t().contiguous()
    lin = ops.PackedLinearX(w, b); lin.tune = True
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    lin(x, out=out)                      # tunes
    b16 = b.bfloat16()
    t_lib = timeit(lambda: torch.addmm(b16, x, wt, out=out))
    t_ours = timeit(lambda: lin(x, out=out))
    gf = 2.0 * M * K * N / 1e9
    print("%-5s M=%d K=%d N=%d  %7.1f GF | hipBLASLt bf16 %7.1f us %7.1f TF/s | conv_bf16x %7.1f us %7.1f TF/s" % (
        name, M, K, N, gf, t_lib, gf / t_lib * 1e3, t_ours, gf / t_ours * 1e3), flush=True)
