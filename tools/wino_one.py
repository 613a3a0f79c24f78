"""Run the dominant launch of the forward (encoder.layers.10) with the given Winograd tile codes a few times -- profiling
target for tools/pmc_raw.sh / pmc_cmd.sh.   python tools/wino_one.py [iters] [tile codes ...]   (default: 5, 0)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
tiles = [int(a) for a in sys.argv[2:]] or [0]
x0 = torch.randn(10, 60, 108, 256, device=dev)
x1 = torch.randn(10, 60, 108, 384, device=dev)
w = torch.randn(512, 320, 3, 3, device=dev) * 0.05
b = torch.randn(512, device=dev)
wi = ops.PackedConv(w, b, [128, 192], groups=2, pad=1, algo="winograd")
out = torch.empty(10, 60, 108, 512, device=dev)
for t in tiles:
    for _ in range(iters):
        wi([x0, x1], out=out, act=ops.ACT_LRELU, slope=0.2, tile=t)
torch.cuda.synchronize()
