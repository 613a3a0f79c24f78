"""Micro-benchmark of the bf16-MFMA deformable conv at the 720p propagation shape (1x180x324, 2x128 ch, dg 16):
fp32 sources vs bf16 sources, every tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(); g.manual_seed(0)
H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "180x324").split("x"))
a = torch.randn(1, H, W, 128, generator=g).to(dev); c = torch.randn(1, H, W, 128, generator=g).to(dev)
offs = torch.cat([torch.randn(1, H, W, 288, generator=g) * 3, torch.rand(1, H, W, 144, generator=g)], -1).to(dev)
w = (torch.randn(128, 256, 3, 3, generator=g) / 48).to(dev); b = torch.randn(128, generator=g).to(dev)
layer = ops.PackedDcn(w, b, 16, pad=1, mfma="bf16")
gf = 2 * H * W * 128 * 2304 * 1e-9
a16, c16 = a.bfloat16(), c.bfloat16()
# "smooth": what the forward sees (offset = 10 tanh(conv) + flow: a slowly varying field + a small residual)
yy, xx = torch.meshgrid(torch.arange(H, device=dev).float(), torch.arange(W, device=dev).float(), indexing="ij")
field = torch.stack([2.0 * torch.sin(yy / 37.0) + 1.5 * torch.cos(xx / 53.0), 1.7 * torch.cos(yy / 41.0) - 2.2 * torch.sin(xx / 29.0)], -1)
smooth = torch.cat([(field.repeat(1, 1, 144) + torch.randn(H, W, 288, device=dev) * 0.3)[None], offs[..., 288:]], -1).contiguous()
for oname, of in (("random 3 px", offs), ("smooth field", smooth)):
    for name, srcs, planar in (("fp32 src", [a, c], False), ("bf16 src", [a16, c16], False),
                               ("bf16 planar", [ops.to_planar16(a16), ops.to_planar16(c16)], True)):
        ref = layer([a16, c16], of, tile=6, out_dtype=torch.bfloat16)
        for tile in [int(v) for v in os.environ.get("DCN_TILES", "1,6,106").split(",")]:
            out = layer(srcs, of, tile=tile, out_dtype=torch.bfloat16, planar=planar)
            same = bool(torch.equal(out, ref)) if name != "fp32 src" else None
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                layer(srcs, of, out=out, tile=tile, planar=planar)
            e1.record(); torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / 10
            print("dcn %dx%d %-12s %-11s tile %3d: %7.1f us  %6.1f TF  identical to NHWC bf16: %s" % (H, W, oname, name, tile, us, gf / us * 1e3, same), flush=True)
