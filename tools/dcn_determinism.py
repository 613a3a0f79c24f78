"""Re-run check of the deformable conv: every tile, NHWC and planar bf16 sources, 20 launches each must be bit-identical
(guards the packed-fp32 hazard of DESIGN.md "Stream overlap" inside mdcn.hip's sampler waves).   python tools/dcn_determinism.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(91)
N, H, W, Co, dg = 2, 14, 22, 128, 16
a = torch.randn(N, H, W, 128, generator=g).bfloat16().to(dev)
c = torch.randn(N, H, W, 128, generator=g).bfloat16().to(dev)
raw = (torch.randn(N, H, W, 432, generator=g) * 0.7).to(dev)
fl = (torch.randn(N, H, W, 4, generator=g) * 2.5).to(dev)
w = (torch.randn(Co, 256, 3, 3, generator=g) / 48).to(dev)
b = torch.randn(Co, generator=g).to(dev)
layer = ops.PackedDcn(w, b, dg, pad=1, mfma="bf16")
l32 = ops.PackedDcn(w, b, dg, pad=1)
pa, pc = ops.to_planar16(a), ops.to_planar16(c)
bad = 0
for tile in (1, 2, 3, 4, 5, 6, 106):
    for name, fn in (("bf16 nhwc", lambda: layer([a, c], raw, flows=fl, tile=tile)), ("bf16 planar", lambda: layer([pa, pc], raw, flows=fl, tile=tile, planar=True)),
                     ("fp32 src bf16 mfma", lambda: layer([a.float(), c.float()], raw, flows=fl, tile=tile)),
                     ("fp32", lambda: l32([a.float(), c.float()], raw, flows=fl, tile=tile))):
        ref = fn()
        n = sum(0 if torch.equal(fn(), ref) else 1 for _ in range(20))
        bad += n
        print("tile %3d %-20s: %d / 20 reruns differ" % (tile, name, n), flush=True)
print("TOTAL differing reruns:", bad)
