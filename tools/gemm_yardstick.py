"""Yardstick only (never used by the product path): rocBLAS / hipBLASLt fp32 GEMM (torch.mm) vs conv_igemm on the token-GEMM
shapes of the 432x240 T=10 forward.   python tools/gemm_yardstick.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, M, K, N in (("qkv", 7360, 512, 1536), ("proj", 7200, 512, 512), ("fc1", 7200, 512, 1960), ("fc2", 7200, 1960, 512),
                      ("sc", 7200, 512, 6272)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    wt = w.t().contiguous()
    lin = ops.PackedLinear(w, b); lin.tune = True
    out = torch.empty(M, N, device=dev)
    lin(x, out=out)                      # tunes
    t_lib = timeit(lambda: torch.addmm(b, x, wt, out=out))
    t_ours = timeit(lambda: lin(x, out=out))
    gf = 2.0 * M * K * N / 1e9
    print("%-5s M=%d K=%d N=%d  %6.1f GF | rocBLAS/hipBLASLt %7.1f us %6.1f TF/s | conv_igemm %7.1f us %6.1f TF/s" % (
        name, M, K, N, gf, t_lib, gf / t_lib * 1e3, t_ours, gf / t_ours * 1e3), flush=True)
