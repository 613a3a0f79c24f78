"""fp32 on the bf16 matrix pipe (conv_bf16x.hip MODE 2, "x3": exact three-way operand split, six bf16 MFMA terms per product)
against the fp32 kernels of the product path, on the e2fgvi 432x240 T=10 layer shapes: device time per tile code and the error
of each kernel family against an fp64 reference of the same call (max |err| / rms of the reference).
    python tools/x3_bench.py [layer,layer,...] [x3 tiles, comma separated]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from e2fgvi_amd import ops

dev = torch.device("cuda:0")
# name, N, H, W, cpg, groups, Cout, k, stride, pad
LAYERS = [("fc1", 7200, 1, 1, [512], 1, 1960, 1, 1, 0), ("fc2", 7200, 1, 1, [1960], 1, 512, 1, 1, 0),
          ("qkv", 7360, 1, 1, [512], 1, 1536, 1, 1, 0), ("proj", 7200, 1, 1, [512], 1, 512, 1, 1, 0),
          ("sc", 7200, 1, 1, [512], 1, 6272, 1, 1, 0), ("ss", 10, 60, 108, [128], 1, 512, 7, 3, 3),
          ("encoder.10", 10, 60, 108, [128, 192], 2, 512, 3, 1, 1), ("encoder.8", 10, 60, 108, [256], 1, 384, 3, 1, 1),
          ("encoder.12", 10, 60, 108, [64, 128], 4, 384, 3, 1, 1), ("encoder.16", 10, 60, 108, [256, 256], 1, 128, 3, 1, 1),
          ("encoder.6", 10, 60, 108, [128], 1, 256, 3, 1, 1), ("encoder.14", 10, 60, 108, [32, 48], 8, 256, 3, 1, 1),
          ("encoder.2", 10, 120, 216, [64], 1, 64, 3, 1, 1), ("encoder.4", 10, 120, 216, [64], 1, 128, 3, 2, 1),
          ("decoder.4", 10, 240, 432, [64], 1, 64, 3, 1, 1), ("decoder.0", 10, 120, 216, [128], 1, 128, 3, 1, 1),
          ("decoder.2", 10, 120, 216, [128], 1, 64, 3, 1, 1),
          ("conv_offset.0", 1, 60, 108, [128, 128, 128, 4], 1, 128, 3, 1, 1), ("conv_offset.2", 1, 60, 108, [128], 1, 128, 3, 1, 1),
          ("conv_offset.6", 1, 60, 108, [128], 1, 432, 3, 1, 1), ("backbone.0", 1, 60, 108, [128, 128, 128], 1, 128, 3, 1, 1),
          ("spynet.5.1", 18, 64, 128, [32], 1, 64, 7, 1, 3), ("spynet.5.2", 18, 64, 128, [64], 1, 32, 7, 1, 3),
          ("fusion", 10, 60, 108, [128, 128], 1, 128, 1, 1, 0)]
only = set(sys.argv[1].split(",")) if len(sys.argv) > 1 and sys.argv[1] else None
tiles = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 5, 6, 7]


def timed(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best


for name, N, H, W, cpg, groups, Cout, k, s, p in LAYERS:
    if only and name not in only:
        continue
    torch.manual_seed(1)
    cin_g = sum(cpg)
    w = torch.randn(Cout, cin_g, k, k, device=dev) * (2.0 / (cin_g * k * k)) ** 0.5
    b = torch.randn(Cout, device=dev) * 0.1
    srcs = [torch.randn(N, H, W, c * groups, device=dev) for c in cpg]
    # fp64 reference: channel order of the virtual concat is (group, source, channel)
    xs = torch.cat([torch.cat([t[..., g * c:(g + 1) * c] for t, c in zip(srcs, cpg)], 3) for g in range(groups)], 3)
    ref = F.conv2d(xs.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=s, padding=p, groups=groups).permute(0, 2, 3, 1)
    rms = ref.pow(2).mean().sqrt().item()
    gflop = 2e-9 * ref.numel() * cin_g * k * k
    prod = ops.PackedConv(w, b, cpg, groups=groups, stride=s, pad=p, algo="auto" if (k == 3 and s == 1) else "igemm")
    prod.tune = True
    f32x = ops.PackedConvX(w, b, cpg, groups=groups, stride=s, pad=p, dtype=torch.float32)
    x3 = ops.PackedConvX(w, b, cpg, groups=groups, stride=s, pad=p, dtype=torch.float32, x3=True)
    out = torch.empty(N, ref.shape[1], ref.shape[2], Cout, device=dev)
    line = "%-14s %7.2f GF K=%5d " % (name, gflop, cin_g * k * k)
    prod(srcs, out=out)                          # tunes on the first call
    us = timed(lambda: prod(srcs, out=out))
    e = (out.double() - ref).abs().max().item() / rms
    line += "| product %7.1f us %6.1f TF err %.1e " % (us, gflop / us * 1e3, e)
    best = None
    for t in tiles:
        try:
            f32x(srcs, out=out, tile=t)
        except Exception:
            continue
        us = timed(lambda: f32x(srcs, out=out, tile=t))
        if best is None or us < best[0]:
            best = (us, t, (out.double() - ref).abs().max().item() / rms)
    if best:
        line += "| f32x t%d %7.1f us err %.1e " % (best[1], best[0], best[2])
    res = []
    for t in tiles:
        try:
            out.zero_()
            x3(srcs, out=out, tile=t)
        except Exception as ex:
            res.append("t%d: %s" % (t, str(ex).splitlines()[0][:40]))
            continue
        e = (out.double() - ref).abs().max().item() / rms
        us = timed(lambda: x3(srcs, out=out, tile=t))
        res.append("t%d %7.1f us %6.1f TF err %.1e" % (t, us, gflop / us * 1e3, e))
    if k == 3 and s == 1 and H % 2 == 0 and W % 2 == 0:
        wres = []
        for shape in (132, 164, 32, 5132, 6064):
            try:
                out.zero_()
                prod(srcs, out=out, tile=ops.W3_BASE + shape)
            except Exception as ex:
                wres.append("w%d: %s" % (shape, str(ex).splitlines()[0][:60]))
                continue
            e = (out.double() - ref).abs().max().item() / rms
            us = timed(lambda: prod(srcs, out=out, tile=ops.W3_BASE + shape))
            wres.append("w%d %7.1f us %6.1f TF err %.1e" % (shape, us, gflop / us * 1e3, e))
        res.append("| wino-x3 " + "  ".join(wres))
    print(line + "| x3 " + "  ".join(res), flush=True)
