"""Micro-benchmark of conv_igemm tile codes on the layer shapes of the north-star clip.
    python tools/conv_bench.py [filter]"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops, lib

dev = torch.device("cuda:0")
SHAPES = {
    # name: (N,H,W, cpg, groups, Cout, k, stride, pad, pack_bk)
    "prop128": (1, 60, 108, [128], 1, 128, 3, 1, 1, 32),
    "off0_bk16": (1, 60, 108, [128, 128, 128, 4], 1, 128, 3, 1, 1, 16),
    "off0_bk32": (1, 60, 108, [128, 128, 128, 4], 1, 128, 3, 1, 1, 32),
    "off6": (1, 60, 108, [128], 1, 432, 3, 1, 1, 32),
    "bbf0": (1, 60, 108, [128, 128, 128], 1, 128, 3, 1, 1, 32),
    "enc8": (10, 60, 108, [256], 1, 384, 3, 1, 1, 32),
    "enc2": (10, 120, 216, [64], 1, 64, 3, 1, 1, 32),
    "qkv": (7200, 1, 1, [512], 1, 1536, 1, 1, 0, 32),
    "proj": (7200, 1, 1, [512], 1, 512, 1, 1, 0, 32),
    "fc2_bk16": (7200, 1, 1, [1960], 1, 512, 1, 1, 0, 16),
    "fc2_bk32": (7200, 1, 1, [1960], 1, 512, 1, 1, 0, 32),
    "fc1": (7200, 1, 1, [512], 1, 1960, 1, 1, 0, 32),
    "spy2": (18, 64, 128, [32], 1, 64, 7, 1, 3, 32),
    "spy3": (18, 64, 128, [64], 1, 32, 7, 1, 3, 32),
    "dec4": (10, 240, 432, [64], 1, 64, 3, 1, 1, 32),
    "dec0": (10, 120, 216, [128], 1, 128, 3, 1, 1, 32),
    "ss": (10, 60, 108, [128], 1, 512, 7, 3, 3, 32),
    "sc": (7200, 1, 1, [512], 1, 6272, 1, 1, 0, 32),
    "dec2": (10, 120, 216, [128], 1, 64, 3, 1, 1, 32),
    "dec6": (10, 240, 432, [64], 1, 3, 3, 1, 1, 32),
    "spy1": (18, 64, 128, [8], 1, 32, 7, 1, 3, 16),
    "spy4": (18, 64, 128, [32], 1, 16, 7, 1, 3, 32),
    "spy1b8": (18, 64, 128, [8], 1, 32, 7, 1, 3, 8),
    "enc0_b8": (10, 240, 432, [4], 1, 64, 3, 2, 1, 8),
    "enc0_b16": (10, 240, 432, [4], 1, 64, 3, 2, 1, 16),
    "spy5": (18, 64, 128, [16], 1, 2, 7, 1, 3, 16),
}
ALLW = [0, 223, 213, 222, 212, 122, 1122, 1222, 224, 214, 1224, 225, 215, 1225, 3225, 227, 217, 1227, 2227, 2123, 2223, 3123, 1233, 1133, 1213, 2213, 3213, 228, 118, 211, 219]
HALO3 = [10000 + i for i in (1, 2, 3, 4, 5, 11, 6, 12, 22, 24, 31, 25, 32, 26, 52, 53, 42, 44, 45)]
CODES = {"proj": ALLW, "fc2_bk32": ALLW, "fc2_bk16": ALLW, "sc": ALLW}
if os.environ.get("CONV_BENCH_SET") == "narrow":          # the non-Winograd 3x3 layers of the fp32 forward
    SHAPES["enc4"] = (10, 120, 216, [64], 1, 128, 3, 2, 1, 32)
    CODES = {"dec6": [0] + HALO3 + ALLW, "enc0_b8": [0] + ALLW, "enc0_b16": [0] + ALLW, "enc4": [0] + ALLW}
SHAPES = {k: v for k, v in SHAPES.items() if k in CODES}
flt = sys.argv[1] if len(sys.argv) > 1 else ""
g = torch.Generator(); g.manual_seed(0)
res = {}
for name, (N, H, W, cpg, groups, Cout, k, stride, pad, bk) in SHAPES.items():
    if flt and flt not in name:
        continue
    cin = sum(cpg)
    w = (torch.randn(Cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    layer = ops.PackedConv(w, b, cpg, groups=groups, stride=stride, pad=pad, bk=bk)
    srcs = [torch.randn(N, H, W, groups * c, generator=g).to(dev) for c in cpg]
    ref = layer(srcs, act=ops.ACT_LRELU, slope=0.1)
    Ho, Wo = layer.out_hw(H, W)
    gflop = 2 * N * Ho * Wo * Cout * cin * k * k / groups * 1e-9
    for code in CODES[name]:
        try:
            out = layer(srcs, act=ops.ACT_LRELU, slope=0.1, tile=code)
        except lib.HipError as e:
            print("%-10s code %3d: %s" % (name, code, str(e)[:90])); continue
        diff = (out - ref).abs().max().item()
        torch.cuda.synchronize()
        iters = 20
        # device time: capture the launches in a HIP graph so the (Python) host cost per launch does not count
        gr = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            layer(srcs, out=out, act=ops.ACT_LRELU, slope=0.1, tile=code)
        torch.cuda.current_stream().wait_stream(st)
        with torch.cuda.graph(gr):
            for _ in range(iters):
                layer(srcs, out=out, act=ops.ACT_LRELU, slope=0.1, tile=code)
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / iters
        print("%-10s code %4d: %8.1f us  %6.1f TF  (diff %.1e)" % (name, code, us, gflop / us * 1e3 / 1e3, diff), flush=True)
        res["%s/%d" % (name, code)] = us
print(json.dumps(res))
