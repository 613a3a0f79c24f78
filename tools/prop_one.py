"""The one-frame propagation launches as a profiling target (tools/pmc_raw.sh): the split-operand deformable conv (tile 4) and the
128 -> 128 / 388 -> 128 split-operand Winograd convs on one 60x108 frame, `iters` launches each.
    python tools/prop_one.py [iters=30] [dcn|wino|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
what = sys.argv[2] if len(sys.argv) > 2 else "all"
g = torch.Generator(); g.manual_seed(0)
H, W = 60, 108
if what in ("dcn", "all"):
    a = torch.randn(1, H, W, 128, generator=g).to(dev); c = torch.randn(1, H, W, 128, generator=g).to(dev)
    raw = (torch.randn(1, H, W, 432, generator=g) * 0.5).to(dev); fl = (torch.randn(1, H, W, 4, generator=g) * 2).to(dev)
    w = (torch.randn(128, 256, 3, 3, generator=g) / 48).to(dev); b = torch.randn(128, generator=g).to(dev)
    layer = ops.PackedDcn(w, b, 16, pad=1, mfma="x3")
    out = layer([a, c], raw, flows=fl, tile=4)
    for _ in range(iters):
        layer([a, c], raw, flows=fl, out=out, tile=4)
if what in ("wino", "all"):
    for cpg in ([128], [128, 128, 128, 4]):
        w = (torch.randn(128, sum(cpg), 3, 3, generator=g) * 0.05).to(dev); b = torch.randn(128, generator=g).to(dev)
        srcs = [torch.randn(1, H, W, c_, generator=g).to(dev) for c_ in cpg]
        layer = ops.PackedConv(w, b, cpg, pad=1, algo="winograd")
        out = torch.empty(1, H, W, 128, device=dev)
        for _ in range(iters + 1):
            layer(srcs, out=out, act=ops.ACT_LRELU, slope=0.1, tile=ops.W3_BASE + 132)
torch.cuda.synchronize()
