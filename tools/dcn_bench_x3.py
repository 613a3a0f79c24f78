"""Micro-benchmark of the split-operand deformable conv (mfma="x3", the fp32 headline's kernel) at the propagation shape
(1x60x108, 2x128 ch, dg 16, fused offset post-processing), every tile; error against the fp32-MFMA kernel of the same call.
    python tools/dcn_bench_x3.py [HxW=60x108] [smooth]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import lib as L, ops
dev = torch.device("cuda:0")
g = torch.Generator(); g.manual_seed(0)
H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "60x108").split("x"))
a = torch.randn(1, H, W, 128, generator=g).to(dev); c = torch.randn(1, H, W, 128, generator=g).to(dev)
raw = (torch.randn(1, H, W, 432, generator=g) * 0.5).to(dev); fl = (torch.randn(1, H, W, 4, generator=g) * 2).to(dev)
w = (torch.randn(128, 256, 3, 3, generator=g) / 48).to(dev); b = torch.randn(128, generator=g).to(dev)
ref = ops.PackedDcn(w, b, 16, pad=1)([a, c], raw, flows=fl, tile=2)
rms = ref.double().pow(2).mean().sqrt().item()
layer = ops.PackedDcn(w, b, 16, pad=1, mfma="x3")
print("library", L.library_key(), "shape 1x%dx%d" % (H, W))
for tile in (0, 1, 2, 3, 4, 5, 6, 7):
    try:
        out = layer([a, c], raw, flows=fl, tile=tile)
    except L.HipError as e:
        print("tile %d: %s" % (tile, str(e)[:90])); continue
    err = (out.double() - ref.double()).abs().max().item() / rms
    again = layer([a, c], raw, flows=fl, tile=tile)
    iters = 40
    gr = torch.cuda.CUDAGraph(); st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        layer([a, c], raw, flows=fl, out=out, tile=tile)
    torch.cuda.current_stream().wait_stream(st)
    with torch.cuda.graph(gr):
        for _ in range(iters):
            layer([a, c], raw, flows=fl, out=out, tile=tile)
    gr.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / iters)
    print("tile %d: %7.1f us   max|x3 - fp32 kernel| / rms %.1e   rerun bit-identical %s" % (tile, best, err, bool(torch.equal(out, again))), flush=True)
