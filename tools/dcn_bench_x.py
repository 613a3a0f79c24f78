"""Micro-benchmark of the bf16-MFMA deformable conv at the 720p propagation shape (1x180x324, 2x128 ch, dg 16):
fp32 sources vs bf16 sources, every tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(); g.manual_seed(0)
H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "180x324").split("x"))
a = torch.randn(1, H, W, 128, generator=g).to(dev); c = torch.randn(1, H, W, 128, generator=g).to(dev)
offs = torch.cat([torch.randn(1, H, W, 288, generator=g) * 3, torch.rand(1, H, W, 144, generator=g)], -1).to(dev)
w = (torch.randn(128, 256, 3, 3, generator=g) / 48).to(dev); b = torch.randn(128, generator=g).to(dev)
layer = ops.PackedDcn(w, b, 16, pad=1, mfma="bf16")
gf = 2 * H * W * 128 * 2304 * 1e-9
for name, srcs in (("fp32 src", [a, c]), ("bf16 src", [a.bfloat16(), c.bfloat16()])):
    for tile in (1, 2, 4, 6, 101, 102, 104, 106):
        out = layer(srcs, offs, tile=tile, out_dtype=torch.bfloat16)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            layer(srcs, offs, out=out, tile=tile)
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 10
        print("dcn %dx%d %s tile %d: %7.1f us  %6.1f TF" % (H, W, name, tile, us, gf / us * 1e3), flush=True)
