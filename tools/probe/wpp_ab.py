"""Alternating A/B of the wide-tile split-operand Winograd kernel (block-shape code 6064) against its ping-pong form (7064: the two waves
of a SIMD one interval apart, a barrier per interval):  [E2FGVI_LIB=<library>] python tools/probe/wpp_ab.py
medians of 7 alternating rounds of 10 launches per code in one process; bit-identity of the outputs."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from e2fgvi_amd import ops
dev = torch.device("cuda:0")
# name, N, H, W, cpg, groups, Cout
LAYERS = [("encoder.10", 10, 60, 108, [128, 192], 2, 512), ("encoder.8", 10, 60, 108, [256], 1, 384), ("encoder.16", 10, 60, 108, [256, 256], 1, 128),
          ("encoder.6", 10, 60, 108, [128], 1, 256), ("decoder.0", 10, 120, 216, [128], 1, 128), ("decoder.2", 10, 120, 216, [128], 1, 64),
          ("decoder.4", 10, 240, 432, [64], 1, 64), ("conv_offset.6", 1, 60, 108, [128], 1, 432), ("encoder.10 x8", 80, 60, 108, [128, 192], 2, 512)]
codes = (ops.W3_BASE + 6064, ops.W3_BASE + 7064)
for name, N, H, W, cpg, groups, Cout in LAYERS:
    torch.manual_seed(3)
    cin_g = sum(cpg)
    w = torch.randn(Cout, cin_g, 3, 3, device=dev) * (2.0 / (cin_g * 9)) ** 0.5
    b = torch.randn(Cout, device=dev) * 0.1
    srcs = [torch.randn(N, H, W, c * groups, device=dev) for c in cpg]
    layer = ops.PackedConv(w, b, cpg, groups=groups, pad=1, algo="winograd")
    outs = {c: torch.empty(N, H, W, Cout, device=dev) for c in codes}
    for c in codes:
        layer(srcs, out=outs[c], act=ops.ACT_LRELU, slope=0.2, tile=c)
    torch.cuda.synchronize()
    same = torch.equal(outs[codes[0]], outs[codes[1]])
    res = {c: [] for c in codes}
    for rnd in range(7):
        for c in codes:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                layer(srcs, out=outs[c], act=ops.ACT_LRELU, slope=0.2, tile=c)
            e1.record(); torch.cuda.synchronize()
            res[c].append(1e3 * e0.elapsed_time(e1) / 10)
    m = {c: statistics.median(v) for c, v in res.items()}
    print("%-14s 6064 %8.1f us (min %7.1f)   7064 %8.1f us (min %7.1f)   ratio %.3f   bit-identical %s"
          % (name, m[codes[0]], min(res[codes[0]]), m[codes[1]], min(res[codes[1]]), m[codes[1]] / m[codes[0]], same), flush=True)
