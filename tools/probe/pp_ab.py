"""Alternating A/B of the split-operand GEMM's shipped tiles against their ping-pong forms (the library of tools/probe/conv_bf16x_pingpong.patch):
python tools/probe/pp_ab.py      -- medians of 7 alternating rounds of 20 launches per tile, one process"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from e2fgvi_amd import ops
dev = torch.device("cuda:0")
# (name, rows, Cin, Cout, (shipped tile, ping-pong tile)): the table's split-operand entries on tiles 7 / 8, by size class (1 / 2 / 8 clips)
LAYERS = [("fc1 c1", 7200, 512, 1960, (7, 107)), ("fc1 c2", 14400, 512, 1960, (7, 107)), ("fc1 c8", 57600, 512, 1960, (7, 107)),
          ("qkv c1", 7360, 512, 1536, (8, 108)), ("qkv c2", 14720, 512, 1536, (8, 108)), ("qkv c2", 14720, 512, 1536, (7, 107)), ("qkv c8", 58880, 512, 1536, (7, 107)),
          ("sc half", 3600, 512, 6272, (8, 108)), ("sc c1", 7200, 512, 6272, (7, 107)), ("sc c2", 14400, 512, 6272, (7, 107)), ("sc c8", 57600, 512, 6272, (7, 107)),
          ("proj c8", 57600, 512, 512, (7, 107)), ("ss c8", (80, 60, 108), 128, 512, (7, 107))]
for name, N, Cin, Cout, tiles in LAYERS:
    torch.manual_seed(1)
    if isinstance(N, tuple):                       # SoftSplit: 7x7 stride 3 pad 3 over [n, h, w, 128]
        n, h, w_ = N
        w = torch.randn(Cout, Cin, 7, 7, device=dev) * (2.0 / (Cin * 49)) ** 0.5
        b = torch.randn(Cout, device=dev) * 0.1
        x = torch.randn(n, h, w_, Cin, device=dev)
        x3 = ops.PackedConvX(w, b, [Cin], groups=1, stride=3, pad=3, dtype=torch.float32, x3=True)
        outs = {t: torch.empty(n, h // 3, w_ // 3, Cout, device=dev) for t in tiles}
    else:
        w = torch.randn(Cout, Cin, 1, 1, device=dev) * (2.0 / Cin) ** 0.5
        b = torch.randn(Cout, device=dev) * 0.1
        x = torch.randn(N, 1, 1, Cin, device=dev)
        x3 = ops.PackedConvX(w, b, [Cin], groups=1, stride=1, pad=0, dtype=torch.float32, x3=True)
        outs = {t: torch.empty(N, 1, 1, Cout, device=dev) for t in tiles}
    for t in tiles:
        x3([x], out=outs[t], tile=t)
    torch.cuda.synchronize()
    same = torch.equal(outs[tiles[0]], outs[tiles[1]])
    res = {t: [] for t in tiles}
    for rnd in range(7):
        for t in tiles:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                x3([x], out=outs[t], tile=t)
            e1.record(); torch.cuda.synchronize()
            res[t].append(1e3 * e0.elapsed_time(e1) / 20)
    m = {t: statistics.median(v) for t, v in res.items()}
    print("%-8s t%-3d %7.1f us (min %6.1f)   t%-3d %7.1f us (min %6.1f)   ratio %.3f   bit-identical %s"
          % (name, tiles[0], m[tiles[0]], min(res[tiles[0]]), tiles[1], m[tiles[1]], min(res[tiles[1]]), m[tiles[1]] / m[tiles[0]], same), flush=True)
