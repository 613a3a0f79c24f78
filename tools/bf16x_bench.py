"""Micro-benchmark of the bf16-data-path conv / linear kernel on the e2fgvi_hq 720x1296 T=10 layer shapes.
    python tools/bf16x_bench.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops

dev = torch.device("cuda:0")
# name, N, H, W, cpg, groups, Cout, k, stride, pad
LAYERS = [("encoder.10", 10, 180, 324, [128, 192], 2, 512, 3, 1, 1), ("encoder.8", 10, 180, 324, [256], 1, 384, 3, 1, 1),
          ("encoder.16", 10, 180, 324, [256, 256], 1, 128, 3, 1, 1), ("encoder.2", 10, 360, 648, [64], 1, 64, 3, 1, 1),
          ("decoder.4", 10, 720, 1296, [64], 1, 64, 3, 1, 1), ("decoder.6", 10, 720, 1296, [64], 1, 3, 3, 1, 1),
          ("decoder.2", 10, 360, 648, [128], 1, 64, 3, 1, 1), ("encoder.14", 10, 180, 324, [32, 48], 8, 256, 3, 1, 1),
          ("encoder.12", 10, 180, 324, [64, 128], 4, 384, 3, 1, 1), ("conv_offset.0", 1, 180, 324, [128, 128, 128, 8], 1, 128, 3, 1, 1),
          ("conv_offset.2", 1, 180, 324, [128], 1, 128, 3, 1, 1), ("conv_offset.6", 1, 180, 324, [128], 1, 432, 3, 1, 1),
          ("soft split", 10, 180, 324, [128], 1, 512, 7, 3, 3),
          ("qkv", 64800 + 1440, 1, 1, [512], 1, 1536, 1, 1, 0), ("proj", 64800, 1, 1, [512], 1, 512, 1, 1, 0),
          ("fc1", 64800, 1, 1, [512], 1, 1960, 1, 1, 0), ("fc2", 64800, 1, 1, [1960], 1, 512, 1, 1, 0),
          ("sc", 64800, 1, 1, [512], 1, 6272, 1, 1, 0),
          ("spynet.5.1", 18, 192, 352, [32], 1, 64, 7, 1, 3), ("spynet.5.2", 18, 192, 352, [64], 1, 32, 7, 1, 3),
          ("spynet.5.3", 18, 192, 352, [32], 1, 16, 7, 1, 3), ("decoder.0", 10, 360, 648, [128], 1, 128, 3, 1, 1)]
only = set(sys.argv[1].split(",")) if len(sys.argv) > 1 and sys.argv[1] else None
tiles = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 6, 21, 26]
for name, N, H, W, cpg, groups, Cout, k, s, p in LAYERS:
    if only and name not in only:
        continue
    w = torch.randn(Cout, sum(cpg), k, k, device=dev) * 0.02
    layer = ops.PackedConvX(w, torch.zeros(Cout, device=dev), cpg, groups=groups, stride=s, pad=p)
    srcs = [torch.randn(N, H, W, c * groups, device=dev).bfloat16() for c in cpg]
    Ho, Wo = layer.out_hw(H, W)
    out = torch.empty(N, Ho, Wo, Cout, dtype=torch.bfloat16, device=dev)
    gflop = 2e-9 * N * Ho * Wo * Cout * sum(cpg) * k * k
    res = {}
    for tile in tiles:
        try:
            layer(srcs, out=out, act=ops.ACT_LRELU, slope=0.2, tile=tile)
        except Exception as e:
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            layer(srcs, out=out, act=ops.ACT_LRELU, slope=0.2, tile=tile)
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 5
        res[tile] = (round(us, 1), round(gflop / us * 1e3, 1))
    print("%-14s %8.1f GFLOP  " % (name, gflop) + "  ".join("t%d: %8.1f us %6.1f TF" % (t, u, f) for t, (u, f) in res.items()), flush=True)
