"""Micro-benchmark of the deformable-conv kernel tiles at the propagation shape (1x60x108, 2x128 ch, dg 16)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(); g.manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H, W = 60, 108
a = torch.randn(N, H, W, 128, generator=g).to(dev); c = torch.randn(N, H, W, 128, generator=g).to(dev)
raw = (torch.randn(N, H, W, 432, generator=g) * 0.5).to(dev); fl = (torch.randn(N, H, W, 4, generator=g) * 2).to(dev)
w = (torch.randn(128, 256, 3, 3, generator=g) / 48).to(dev); b = torch.randn(128, generator=g).to(dev)
layer = ops.PackedDcn(w, b, 16, pad=1)
ref = layer([a, c], raw, flows=fl, tile=2)
for tile in (0, 1, 2, 3, 4, 5, 6, 105, 104, 102, 5):
    out = layer([a, c], raw, flows=fl, tile=tile)
    diff = (out - ref).abs().max().item()
    iters = 20
    gr = torch.cuda.CUDAGraph(); st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        layer([a, c], raw, flows=fl, out=out, tile=tile)
    torch.cuda.current_stream().wait_stream(st)
    with torch.cuda.graph(gr):
        for _ in range(iters):
            layer([a, c], raw, flows=fl, out=out, tile=tile)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    gf = 2 * N * H * W * 128 * 2304 * 1e-9
    print("dcn N=%d tile %d: %7.1f us  %5.1f TF (diff %.1e)" % (N, tile, us, gf / us * 1e3 / 1e3, diff), flush=True)
# the same layer on the split-operand MFMA (mfma="x3")
layer3 = ops.PackedDcn(w, b, 16, pad=1, mfma="x3")
for tile in (0, 1, 2, 3, 4):
    out = layer3([a, c], raw, flows=fl, tile=tile)
    diff = (out - ref).abs().max().item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        layer3([a, c], raw, flows=fl, out=out, tile=tile)
    e0.record()
    for _ in range(20):
        layer3([a, c], raw, flows=fl, out=out, tile=tile)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    print("dcn x3 N=%d tile %d: %7.1f us  %5.1f TF fp32-equivalent (diff to the fp32 kernel %.1e)" % (N, tile, us, gf / us, diff), flush=True)
