"""Run one Winograd layer a few times (profiling target).  python tools/wino_one.py [tile] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops
dev = torch.device("cuda:0")
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
x = torch.randn(10, 60, 108, 256, device=dev)
w = torch.randn(384, 256, 3, 3, device=dev) * 0.05
b = torch.randn(384, device=dev)
wi = ops.PackedConv(w, b, [256], pad=1, algo="winograd")
d = ops.PackedConv(w, b, [256], pad=1)
out = torch.empty(10, 60, 108, 384, device=dev)
for _ in range(iters):
    wi([x], out=out, act=ops.ACT_LRELU, slope=0.2, tile=tile)
    d([x], out=out, act=ops.ACT_LRELU, slope=0.2)
torch.cuda.synchronize()
