"""Where the F(2x2,3x3) Winograd kernel's cycles go: conv_wino.hip built with -DE2_WINO_TIMING accumulates, per wave, the
s_memtime cycles of five K-loop segments (see the source).  Diagnostics only.
    python tools/wino_timing.py --build        (CPU: compiles tools/probe/libe2fgvi_timing.so from the product objects)
    python tools/wino_timing.py                (GPU: runs encoder.layers.10 / .8 and a propagation conv, prints the segment sums)"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from e2fgvi_amd import build as B
SO = os.path.join(ROOT, "tools", "probe", "libe2fgvi_timing.so")

if "--variant" in sys.argv:
    # an un-instrumented build of the library with -DE2_WINO_VARIANT=<n> (experiment switches of conv_wino.hip) for A/B runs:
    #   E2FGVI_LIB=tools/probe/libe2fgvi_v<n>.so python tools/wino_bench.py ...
    n = int(sys.argv[sys.argv.index("--variant") + 1])
    B.build()
    obj = os.path.join(ROOT, "tools", "probe", "conv_wino_v%d.o" % n)
    subprocess.check_call([B._hipcc()] + B.FLAGS + ["-DE2_WINO_VARIANT=%d" % n, "-c", os.path.join(B.CSRC, "conv_wino.hip"), "-o", obj])
    objs = [obj if o == "conv_wino.o" else os.path.join(B.CSRC, "build", o) for _, o, _ in B.UNITS]
    out = os.path.join(ROOT, "tools", "probe", "libe2fgvi_v%d.so" % n)
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    print("built", out)
    sys.exit(0)
X3 = "--x3" in sys.argv      # the split-operand build (conv_wino_x3.o): segments = weights wait / fragments (LDS reads + transform +
#                              split) / MFMA issue / park + next patch loads / stage barrier
if X3:
    SO = os.path.join(ROOT, "tools", "probe", "libe2fgvi_timing_x3.so")
if "--build" in sys.argv and X3:
    B.build()
    obj = os.path.join(ROOT, "tools", "probe", "conv_wino_x3_timing.o")
    subprocess.check_call([B._hipcc()] + B.FLAGS + B.NOPK + ["-DE2_WINO_X3=1", "-DE2_WINO_TIMING", "-c", os.path.join(B.CSRC, "conv_wino.hip"), "-o", obj])
    objs = [obj if o == "conv_wino_x3.o" else os.path.join(B.CSRC, "build", o) for _, o, _ in B.UNITS]
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs)
    print("built", SO)
    sys.exit(0)
if "--build" in sys.argv:
    B.build()
    obj = os.path.join(ROOT, "tools", "probe", "conv_wino_timing.o")
    subprocess.check_call([B._hipcc()] + B.FLAGS + ["-DE2_WINO_TIMING", "-c", os.path.join(B.CSRC, "conv_wino.hip"), "-o", obj])
    objs = [obj if o == "conv_wino.o" else os.path.join(B.CSRC, "build", o) for _, o, _ in B.UNITS]
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs)
    print("built", SO)
    sys.exit(0)

import numpy as np, torch
from e2fgvi_amd import lib as L
L.LIB_PATH = SO
from e2fgvi_amd import ops
dev = torch.device("cuda:0")
raw = C.CDLL(SO)
raw.e2fgvi_wino_timing_read.argtypes = [C.c_void_p, C.c_int32]
CASES = [("encoder.10 <2,64>", 10, [128, 192], 2, 512, 64), ("encoder.10 <1,32>", 10, [128, 192], 2, 512, 132),
         ("encoder.8 <2,64>", 10, [256], 1, 384, 64), ("prop 128->128 x1 <1,32>", 1, [128], 1, 128, 132),
         ("prop 128->128 x10 <1,32>", 10, [128], 1, 128, 132)]
NAMES = ["weights wait", "transform+MFMA issue", "park next stage (store_raw)", "issue stage after (load_raw)", "stage barrier"]
if X3:
    CASES = [("encoder.10 x3 <1,64>", 10, [128, 192], 2, 512, ops.W3_BASE + 164), ("encoder.10 x3 <1,32>", 10, [128, 192], 2, 512, ops.W3_BASE + 132),
             ("prop 128->128 x1 x3 <1,32>", 1, [128], 1, 128, ops.W3_BASE + 132), ("prop 384->128 x1 x3 <1,32>", 1, [128, 128, 128], 1, 128, ops.W3_BASE + 132),
             ("prop 128->128 x10 x3 <1,32>", 10, [128], 1, 128, ops.W3_BASE + 132)]
    NAMES = ["issue next weights + wait for this stage's", "fragments: LDS reads + transform + split", "MFMA issue",
             "park next patch + issue the loads after", "stage barrier"]
for name, N, cpg, g, Cout, tile in CASES:
    srcs = [torch.randn(N, 60, 108, c * g, device=dev) for c in cpg]
    w = torch.randn(Cout, sum(cpg), 3, 3, device=dev) * 0.05
    layer = ops.PackedConv(w, torch.randn(Cout, device=dev), cpg, groups=g, pad=1, algo="winograd")
    out = torch.empty(N, 60, 108, Cout, device=dev)
    for _ in range(3):
        layer(srcs, out=out, act=ops.ACT_LRELU, slope=0.2, tile=tile)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); layer(srcs, out=out, act=ops.ACT_LRELU, slope=0.2, tile=tile); e1.record(); torch.cuda.synchronize()
    buf = np.zeros(64 * 8 * 8, np.uint64)
    raw.e2fgvi_wino_timing_read(buf.ctypes.data_as(C.c_void_p), buf.size)
    t = buf.reshape(64, 8, 8).astype(np.float64)
    nst = t[0, 0, 7]
    chunks = 2 * nst
    kl, ep = t[..., 5].mean(), t[..., 6].mean()
    shape = tile % 1000 if X3 else tile
    mt = 2 if shape < 100 else 1
    tn = (shape % 100) // 32
    mfma_wave = (nst * 12 * mt * tn * 32) if X3 else chunks * 8 * mt * tn * 64            # cycles of this wave's own MFMAs
    print("%-26s %7.1f us (instrumented)  K loop %8.0f cyc / wave, epilogue %6.0f; %d chunks; own MFMA cycles %8.0f (x2 waves per SIMD = %.0f %% of the K loop)"
          % (name, 1e3 * e0.elapsed_time(e1), kl, ep, chunks, mfma_wave, 200 * mfma_wave / kl))
    for k in range(5):
        print("      %-32s %8.0f cyc  %5.1f %% of the K loop   (per chunk %6.0f)" % (NAMES[k], t[..., k].mean(), 100 * t[..., k].mean() / kl, t[..., k].mean() / chunks))
    print("      wave-to-wave spread of the K loop: min %.0f max %.0f" % (t[..., 5].min(), t[..., 5].max()))
