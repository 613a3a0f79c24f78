"""Micro-benchmark of the fp32 conv tile codes on SPyNet's pyramid levels (18 frame pairs of the 432x240 T=10 clip):
7x7 convs 8->32->64->32->16->2 on 64x128 (level 5) ... 2x4 (level 0) images.   python tools/spynet_bench.py [levels]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops, lib
dev = torch.device("cuda:0")
levels = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4, 3, 2]
LAYERS = [(8, 32), (32, 64), (64, 32), (32, 16), (16, 2)]
HALO = [10000 + i for i in (7, 8, 9, 10, 27, 28, 30, 51, 54, 41, 43, 46)]
IGEMM = [223, 213, 222, 212, 1222, 1212, 3225, 3215, 2223, 3223, 1223, 3213]
g = torch.Generator(); g.manual_seed(0)
for lv in levels:
    H, W = 2 << lv, 4 << lv
    for j, (cin, cout) in enumerate(LAYERS):
        w = (torch.randn(cout, cin, 7, 7, generator=g) / math.sqrt(cin * 49)).to(dev)
        layer = ops.PackedConv(w, torch.randn(cout, generator=g).to(dev), [cin], pad=3)
        x = torch.randn(18, H, W, cin, generator=g).to(dev)
        out = layer([x], act=ops.ACT_RELU)
        gf = 2 * 18 * H * W * cout * cin * 49 * 1e-9
        row = []
        for code in [0] + HALO + IGEMM:
            try:
                layer([x], out=out, act=ops.ACT_RELU, tile=code)
            except Exception:
                continue
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph(); st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                layer([x], out=out, act=ops.ACT_RELU, tile=code)
            torch.cuda.current_stream().wait_stream(st)
            with torch.cuda.graph(gr):
                for _ in range(20):
                    layer([x], out=out, act=ops.ACT_RELU, tile=code)
            gr.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            row.append((1e3 * e0.elapsed_time(e1) / 20, code))
        base = [u for u, c in row if c == 0][0]
        best = sorted(row)[:4]
        print("level %d %dx%d  .%d %2d->%2d  %6.2f GF  auto %7.1f us | best: %s" % (
            lv, H, W, j, cin, cout, gf, base, "  ".join("%d: %.1f" % (c, u) for u, c in best)), flush=True)
