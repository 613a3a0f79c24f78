"""Round 4: A/B builds of the split-operand Winograd object (conv_wino_x3.o) with -DE2_WINO_VARIANT=<n> and / or -DE2_WINO_TIMING.
    python tools/r4_variants.py build 0 8 t0 t8     (CPU: tools/probe/libe2fgvi_x3v<n>.so, tools/probe/libe2fgvi_x3t<n>.so)
    python tools/r4_variants.py timing 0 8          (GPU: per-phase s_memtime sums of conv_wino_x3w_kernel on encoder.layers.10)
    E2FGVI_LIB=tools/probe/libe2fgvi_x3v8.so python tools/x3_bench.py ...   (GPU: device times of a variant)
E2_WINO_VARIANT bits used by conv_wino_x3w_kernel: 8 = phase skew between SIMD partners, 16 = s_setprio 3 around the MFMA phases."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from e2fgvi_amd import build as B
PROBE = os.path.join(ROOT, "tools", "probe")


def so_path(tag):
    return os.path.join(PROBE, "libe2fgvi_x3%s.so" % tag)


if sys.argv[1] == "build":
    B.build()
    for tag in sys.argv[2:]:
        timing, n = tag.startswith("t"), int(tag.lstrip("tv"))
        tag = ("t%d" if timing else "v%d") % n
        obj = os.path.join(PROBE, "conv_wino_x3%s.o" % tag)
        subprocess.check_call([B._hipcc()] + B.FLAGS + B.NOPK + ["-DE2_WINO_X3=1", "-DE2_WINO_VARIANT=%d" % n] + (["-DE2_WINO_TIMING"] if timing else [])
                              + ["-c", os.path.join(B.CSRC, "conv_wino.hip"), "-o", obj])
        objs = [obj if o == "conv_wino_x3.o" else os.path.join(B.CSRC, "build", o) for _, o, _ in B.UNITS]
        subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so_path(tag)] + objs)
        print("built", so_path(tag))
    sys.exit(0)

import numpy as np, torch
NAMES = ["P0 fragments row-tile 0", "M0 plane waits + 24 MFMAs", "P1 fragments row-tile 1", "LDS-DMA pieces of stage + 2",
         "M1 24 MFMAs + plane reloads", "stage barrier"]
CASES = [("encoder.10", 10, 60, 108, [128, 192], 2, 512), ("encoder.8", 10, 60, 108, [256], 1, 384), ("decoder.4", 10, 240, 432, [64], 1, 64),
         ("conv_offset.6 x1", 1, 60, 108, [128], 1, 432)]
for n in sys.argv[2:]:
    so = so_path("t%d" % int(n))
    # one library per process image: run each variant in a child
    if os.environ.get("R4_CHILD") != n:
        subprocess.check_call([sys.executable, __file__, "timing", n], env=dict(os.environ, R4_CHILD=n, E2FGVI_LIB=so))
        continue
    from e2fgvi_amd import ops
    dev = torch.device("cuda:0")
    raw = C.CDLL(so)
    raw.e2fgvi_wino_timing_read.argtypes = [C.c_void_p, C.c_int32]
    raw.e2fgvi_wino_timing_read2.argtypes = [C.c_void_p, C.c_int32]
    print("==== variant %s" % n)
    for name, N, H, W, cpg, g, Cout in CASES:
        srcs = [torch.randn(N, H, W, c * g, device=dev) for c in cpg]
        w = torch.randn(Cout, sum(cpg), 3, 3, device=dev) * 0.05
        layer = ops.PackedConv(w, torch.randn(Cout, device=dev), cpg, groups=g, pad=1, algo="winograd")
        out = torch.empty(N, H, W, Cout, device=dev)
        tile = ops.W3_BASE + 6064
        for _ in range(3):
            layer(srcs, out=out, act=ops.ACT_LRELU, slope=0.2, tile=tile)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); layer(srcs, out=out, act=ops.ACT_LRELU, slope=0.2, tile=tile); e1.record(); torch.cuda.synchronize()
        buf = np.zeros(64 * 8 * 8, np.uint64)
        raw.e2fgvi_wino_timing_read(buf.ctypes.data_as(C.c_void_p), buf.size)
        t = buf.reshape(64, 8, 8).astype(np.float64)
        nst = t[0, 0, 7]
        kl = t[..., 6].mean()
        print("%-18s %7.1f us (instrumented)  K loop %8.0f cyc / wave = %6.0f per stage (%d stages); own MFMA cycles per stage 1536 (x2 waves per SIMD = %.0f %% of the K loop)"
              % (name, 1e3 * e0.elapsed_time(e1), kl, kl / nst, nst, 100 * 3072 * nst / kl))
        buf2 = np.zeros(64 * 8 * 4, np.uint64)
        raw.e2fgvi_wino_timing_read2(buf2.ctypes.data_as(C.c_void_p), buf2.size)
        t2 = buf2.reshape(64, 8, 4).astype(np.float64)
        print("   whole kernel per wave: before the K loop %6.0f  K loop %7.0f  epilogue %6.0f  total %7.0f cycles" % tuple(t2[..., k].mean() for k in range(4)))
        for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
            print("   %s:" % grp + "".join("  %s %5.0f" % (NAMES[k].split()[0], t[:, sl, k].mean() / nst) for k in range(6))
                  + "   K loop %6.0f" % (t[:, sl, 6].mean() / nst))
