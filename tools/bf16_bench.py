"""fp32-MFMA vs bf16-MFMA conv kernels on large layers (HQ 720p shapes and the base shapes)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops
dev = torch.device("cuda:0")
SHAPES = {
    "enc8_base": (10, 60, 108, [256], 1, 384, 3, 1, 1),
    "enc8_720p": (10, 180, 324, [256], 1, 384, 3, 1, 1),
    "dec0_720p": (4, 360, 648, [128], 1, 128, 3, 1, 1),
    "qkv_720p": (64800, 1, 1, [512], 1, 1536, 1, 1, 0),
    "fc2_720p": (64800, 1, 1, [1960], 1, 512, 1, 1, 0),
    "off0_720p": (1, 180, 324, [128, 128, 128, 4], 1, 128, 3, 1, 1),
    "prop128_base": (1, 60, 108, [128], 1, 128, 3, 1, 1),
}
g = torch.Generator(); g.manual_seed(0)
def timeit(layer, srcs, tile):
    out = layer(srcs, tile=tile)
    gr = torch.cuda.CUDAGraph(); st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        layer(srcs, out=out, tile=tile)
    torch.cuda.current_stream().wait_stream(st)
    with torch.cuda.graph(gr):
        for _ in range(10):
            layer(srcs, out=out, tile=tile)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / 10
for name, (N, H, W, cpg, groups, Cout, k, stride, pad) in SHAPES.items():
    cin = sum(cpg)
    w = (torch.randn(Cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    srcs = [torch.randn(N, H, W, groups * c, generator=g).to(dev) for c in cpg]
    l32 = ops.PackedConv(w, b, cpg, groups=groups, stride=stride, pad=pad)
    lbf = ops.PackedConv(w, b, cpg, groups=groups, stride=stride, pad=pad, precision="bf16")
    Ho, Wo = l32.out_hw(H, W)
    gflop = 2 * N * Ho * Wo * Cout * cin * k * k / groups * 1e-9
    t32 = timeit(l32, srcs, 0)
    res = ["fp32 %8.1f us %6.1f TF" % (t32, gflop / t32 * 1e3 / 1e3)]
    for tile in (0, 1, 5, 6, 2, 3):
        try:
            t = timeit(lbf, srcs, tile)
            res.append("bf16[t%d] %7.1f us %6.1f TF" % (tile, t, gflop / t * 1e3 / 1e3))
        except Exception as e:
            res.append("bf16[t%d] err" % tile)
    print("%-13s %s" % (name, " | ".join(res)), flush=True)
