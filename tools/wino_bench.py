"""Winograd F(2x2,3x3) vs implicit-GEMM conv on the 3x3 stride-1 layers of the 432x240 T=10 forward.
    python tools/wino_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops

dev = torch.device("cuda:0")
LAYERS = [  # name, N, H, W, cpg, groups, Cout
    ("enc.2   64->64  @120x216", 10, 120, 216, [64], 1, 64),
    ("enc.6  128->256 @60x108", 10, 60, 108, [128], 1, 256),
    ("enc.8  256->384", 10, 60, 108, [256], 1, 384),
    ("enc.10 640->512 g2", 10, 60, 108, [128, 192], 2, 512),
    ("enc.12 768->384 g4", 10, 60, 108, [64, 128], 4, 384),
    ("enc.14 640->256 g8", 10, 60, 108, [32, 48], 8, 256),
    ("enc.16 512->128", 10, 60, 108, [256, 256], 1, 128),
    ("dec.0  128->128 @120x216", 10, 120, 216, [128], 1, 128),
    ("dec.2  128->64  @120x216", 10, 120, 216, [128], 1, 64),
    ("dec.4   64->64  @240x432", 10, 240, 432, [64], 1, 64),
    ("prop   128->128 @60x108 x1", 1, 60, 108, [128], 1, 128),
    ("prop off.0 388->128 x1", 1, 60, 108, [128, 128, 128, 4], 1, 128),
    ("prop off.6 128->432 x1", 1, 60, 108, [128], 1, 432),
    ("prop bb.0 256->128 x1", 1, 60, 108, [128, 128], 1, 128),
    ("prop bb.0 384->128 x1", 1, 60, 108, [128, 128, 128], 1, 128),
    ("prop   128->128 x2", 2, 60, 108, [128], 1, 128),
    ("prop   128->128 x4", 4, 60, 108, [128], 1, 128),
    ("prop off.0 388->128 x2", 2, 60, 108, [128, 128, 128, 4], 1, 128),
]
if os.environ.get("WINO_ONLY_PROP"):
    LAYERS = [l for l in LAYERS if l[0].startswith("prop")]

def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

tiles = [int(a) for a in sys.argv[1:]] or [0]
for name, N, H, W, cpg, g, Cout in LAYERS:
    srcs = [torch.randn(N, H, W, g * c, device=dev) for c in cpg]
    w = torch.randn(Cout, sum(cpg), 3, 3, device=dev) * 0.05
    b = torch.randn(Cout, device=dev)
    d = ops.PackedConv(w, b, cpg, groups=g, pad=1)
    wi = ops.PackedConv(w, b, cpg, groups=g, pad=1, algo="winograd")
    out = torch.empty(N, H, W, Cout, device=dev)
    gf = 2.0 * N * H * W * Cout * sum(cpg) * 9 / 1e9
    td = timeit(lambda: d(srcs, out=out, act=ops.ACT_LRELU, slope=0.2))
    line = "%-28s %7.1f GF  igemm %8.1f us %6.1f TF/s |" % (name, gf, td, gf / td * 1e3)
    for t in tiles:
        tw = timeit(lambda: wi(srcs, out=out, act=ops.ACT_LRELU, slope=0.2, tile=t))
        line += "  wino[%d] %8.1f us %6.1f TF/s x%.2f" % (t, tw, gf / tw * 1e3, td / tw)
    print(line, flush=True)
