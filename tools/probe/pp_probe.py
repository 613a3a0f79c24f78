"""Interval timeline of the ping-pong split-operand GEMM (csrc/conv_bf16x.hip, PP).  Build the stamps in:
    patch -p0 < tools/probe/conv_bf16x_pingpong_probe.patch && python -c "from e2fgvi_amd import build; build.build_variant('pp', '-DE2_PP_PROBE=1')" && git checkout e2fgvi_amd/csrc/conv_bf16x.hip
    E2FGVI_LIB=e2fgvi_amd/csrc/libe2fgvi_hip_pp.so python tools/probe/pp_probe.py [tile]
Workgroup 0's waves 0 (leading half) and 4 (trailing half, same SIMD) stamp s_memtime around every phase of the K loop into the
layer's bias buffer (the output of that launch is garbage).  Prints, per K-step, the cycles of P0 | bar | M0 | bar | P1 | bar | M1 | bar."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from e2fgvi_amd import ops

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 107
dev = torch.device("cuda:0")
torch.manual_seed(1)
N, Cin, Cout = 7200, 512, 1960
w = torch.randn(Cout, Cin, 1, 1, device=dev) * 0.06
b = torch.zeros(Cout, device=dev)
x = torch.randn(N, 1, 1, Cin, device=dev)
x3 = ops.PackedConvX(w, b, [Cin], groups=1, stride=1, pad=0, dtype=torch.float32, x3=True)
out = torch.empty(N, 1, 1, Cout, device=dev)
for _ in range(3):
    x3.bias.zero_()
    x3([x], out=out, tile=tile)
torch.cuda.synchronize()
ts = x3.bias.view(torch.int64).cpu()
names = ["P0", "bar", "M0", "bar", "P1", "bar", "M1", "bar"]
for wave, base in ((0, 0), (4, 400)):
    t = ts[base:base + 128].tolist()
    print("wave %d: first stamp %d, loop %d cycles" % (wave, t[0], t[127] - t[0]))
    for s in range(16):
        seg = t[8 * s:8 * s + 8] + ([t[8 * s + 8]] if s < 15 else [t[8 * s + 7]])
        print("  step %2d: " % s + "  ".join("%s %5d" % (names[k], seg[k + 1] - seg[k]) for k in range(8 if s < 15 else 7)))
w0, w4 = ts[0:128].tolist(), ts[400:528].tolist()
print("trailing - leading at step 0 P0 start: %d cycles" % (w4[0] - w0[0]))
