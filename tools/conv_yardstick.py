"""Yardstick only (never used by the product path): MIOpen (torch F.conv2d, channels_last, benchmark mode) vs this library on
the 3x3 layer shapes of the forward -- fp32 at 432x240 T=10 and bf16 at 720x1296 T=10.   python tools/conv_yardstick.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from e2fgvi_amd import ops
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
F32 = [("encoder.10 640->512 g2", 10, 60, 108, [128, 192], 2, 512), ("encoder.8 256->384", 10, 60, 108, [256], 1, 384),
       ("encoder.16 512->128", 10, 60, 108, [256, 256], 1, 128), ("decoder.4 64->64 @240x432", 10, 240, 432, [64], 1, 64),
       ("prop 128->128 one frame", 1, 60, 108, [128], 1, 128), ("conv_offset.6 128->432 one frame", 1, 60, 108, [128], 1, 432)]
B16 = [("encoder.10 640->512 g2", 10, 180, 324, [128, 192], 2, 512), ("encoder.8 256->384", 10, 180, 324, [256], 1, 384),
       ("decoder.4 64->64 @720x1296", 10, 720, 1296, [64], 1, 64), ("prop 128->128 one frame", 1, 180, 324, [128], 1, 128)]
for tag, cases, dt in (("fp32", F32, torch.float32), ("bf16", B16, torch.bfloat16)):
    for name, N, H, W, cpg, g, Cout in cases:
        cin = sum(cpg) * g
        w = torch.randn(Cout, sum(cpg), 3, 3, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        srcs = [torch.randn(N, H, W, c * g, device=dev).to(dt) for c in cpg]
        # the library takes the virtual concat; MIOpen gets the materialised tensor (its copy is not timed)
        xcat = torch.cat([s.view(N, H, W, g, c) for s, c in zip(srcs, cpg)], 4).reshape(N, H, W, cin).permute(0, 3, 1, 2)
        xcat = xcat.contiguous(memory_format=torch.channels_last)
        wt = w.to(dt).contiguous(memory_format=torch.channels_last); bt = b.to(dt)
        gf = 2.0 * N * H * W * Cout * sum(cpg) * 9 / 1e9
        try:
            t_lib = timeit(lambda: F.leaky_relu(F.conv2d(xcat, wt, bt, padding=1, groups=g), 0.2))
        except Exception as e:
            t_lib = float("nan")
        if dt == torch.float32:
            layer = ops.PackedConv(w, b, cpg, groups=g, pad=1, algo="auto")
            out = torch.empty(N, H, W, Cout, device=dev)
            t_ours = timeit(lambda: layer(srcs, out=out, act=ops.ACT_LRELU, slope=0.2))
        else:
            layer = ops.PackedConvX(w, b, cpg, groups=g, pad=1); layer.tune = True
            out = torch.empty(N, H, W, Cout, device=dev, dtype=dt)
            layer(srcs, out=out, act=ops.ACT_LRELU, slope=0.2)
            t_ours = timeit(lambda: layer(srcs, out=out, act=ops.ACT_LRELU, slope=0.2))
        print("%-4s %-34s %8.1f GF | MIOpen conv + leaky_relu %8.1f us %7.1f TF/s | this library (fused) %8.1f us %7.1f TF/s" % (
            tag, name, gf, t_lib, gf / t_lib * 1e3, t_ours, gf / t_ours * 1e3), flush=True)
