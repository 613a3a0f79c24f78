"""Wide-tile split-operand Winograd kernel (tile code 46064): time against the number of 16-channel K stages on fixed output shapes,
to separate the per-workgroup fixed cost (prologue + epilogue) from the per-stage cost.   python tools/x3w_fixed_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops

dev = torch.device("cuda:0")


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best


CODE = int(sys.argv[1]) if len(sys.argv) > 1 else 6064
for name, N, H, W, Cout in (("decoder.4 shape", 10, 240, 432, 64), ("decoder.0 shape", 10, 120, 216, 128), ("encoder shape", 10, 60, 108, 512),
                            ("one frame", 1, 60, 108, 448)):
    wgs = N * ((H + 15) // 16) * ((W + 15) // 16) * ((Cout + 63) // 64)
    row = []
    for cin in (16, 32, 64, 128, 256):
        g = torch.Generator().manual_seed(cin)
        layer = ops.PackedConv((torch.randn(Cout, cin, 3, 3, generator=g) * 0.05).to(dev), torch.randn(Cout, generator=g).to(dev), [cin], pad=1, algo="winograd")
        x = torch.randn(N, H, W, cin, generator=g).to(dev)
        out = torch.empty(N, H, W, Cout, device=dev)
        us = timed(lambda: layer([x], out=out, act=ops.ACT_LRELU, slope=0.2, tile=ops.W3_BASE + CODE))
        row.append((cin // 16, us))
    (s0, t0), (s1, t1) = row[1], row[-1]
    slope = (t1 - t0) / (s1 - s0)
    print("%-16s %5d workgroups (%.1f per CU): " % (name, wgs, wgs / 256.0) + "  ".join("%d stages %.1f us" % r for r in row)
          + "  | per stage %.1f us, at zero stages %.1f us" % (slope, t0 - slope * s0), flush=True)
