"""Kernel-level reproducer for DESIGN.md C4: conv_wino_x3w_kernel (tile code 46064) on one stream, an aggressor on another,
every output compared bit for bit with the kernel's result alone.  Seconds per configuration instead of whole forwards.

    python tools/c4_repro.py [rounds=40] [launches per round=6]
    E2FGVI_LIB=tools/probe/libe2fgvi_x3v32.so python tools/c4_repro.py      # a variant build (tools/r4_variants.py build 32 64)

Aggressors: `spynet` = the side stream's own kernels (7x7 convs 32 -> 64 -> 32 on 18 images of 64x128, the packed-math-free
builds), `copy` = plain device copies of 256 MB (memory system only, no matrix pipe), `p4` = round 3's split-operand Winograd
kernel on a second layer, `none` = the control.  Victims: encoder.layers.8 (256 -> 384, one source), encoder.layers.10 (two
sources, two groups).  Hypotheses the variants test (conv_wino.hip, dma_piece):
    32  M0 is not restored behind an LDS-DMA piece      64  it is restored 32 idle cycles behind it
If 32 is clean where the product form is not, the piece reads M0 later than at issue when the vector-memory queue is stalled."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
per = int(sys.argv[2]) if len(sys.argv) > 2 else 6
# further arguments: Winograd tile codes of more victims on the encoder.8 shape (45132 = four positions per wave, 40164 / 40132 = round
# 3's eight-wave split-operand shapes, 2464 = fp32 F(2x4), 64 = fp32 F(2x2)): every kernel with asm-issued weight prefetches
more = [int(a) for a in sys.argv[3:]]
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(11)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
WIDE = ops.W3_BASE + ops.W3_WIDE

victims = {
    "encoder.8": (ops.PackedConv(rnd(384, 256, 3, 3) * 0.05, rnd(384), [256], pad=1, algo="winograd"), [rnd(10, 60, 108, 256)], 384),
    "encoder.10": (ops.PackedConv(rnd(512, 320, 3, 3) * 0.05, rnd(512), [128, 192], groups=2, pad=1, algo="winograd"),
                   [rnd(10, 60, 108, 256), rnd(10, 60, 108, 384)], 512),
}
sp1 = ops.PackedConv(rnd(64, 32, 7, 7) * 0.05, rnd(64), [32], pad=3); sp1.nopk = True
sp2 = ops.PackedConv(rnd(32, 64, 7, 7) * 0.05, rnd(32), [64], pad=3); sp2.nopk = True
sx, sy, sz = rnd(18, 64, 128, 32), torch.empty(18, 64, 128, 64, device=dev), torch.empty(18, 64, 128, 32, device=dev)
big_a, big_b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)
p4 = ops.PackedConv(rnd(256, 128, 3, 3) * 0.05, rnd(256), [128], pad=1, algo="winograd")
p4x, p4y = rnd(10, 60, 108, 128), torch.empty(10, 60, 108, 256, device=dev)


def aggress(kind, n):
    for _ in range(n):
        if kind == "spynet":
            sp1([sx], out=sy, act=ops.ACT_RELU)
            sp2([sy], out=sz, act=ops.ACT_RELU)
        elif kind == "copy":
            big_b.copy_(big_a)
        elif kind == "p4":
            p4([p4x], out=p4y, act=ops.ACT_LRELU, slope=0.2, tile=ops.W3_BASE + 5132)


for code in more:
    victims["encoder.8 tile %d" % code] = victims["encoder.8"][:2] + (384, code)
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream()
print("library:", os.environ.get("E2FGVI_LIB", "in-tree"))
for vname, spec in victims.items():
    layer, srcs, cout = spec[:3]
    WIDE = spec[3] if len(spec) > 3 else ops.W3_BASE + ops.W3_WIDE
    ref = torch.empty(10, 60, 108, cout, device=dev)
    layer(srcs, out=ref, act=ops.ACT_LRELU, slope=0.2, tile=WIDE)
    torch.cuda.synchronize()
    outs = [torch.empty_like(ref) for _ in range(per)]
    for kind in ("none", "spynet", "copy", "p4"):
        bad, blocks = 0, set()
        for _ in range(rounds):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                aggress(kind, 6 * per)
            for o in outs:
                layer(srcs, out=o, act=ops.ACT_LRELU, slope=0.2, tile=WIDE)
            main.wait_stream(side)
            torch.cuda.synchronize()
            for o in outs:
                if not torch.equal(o, ref):
                    bad += 1
                    d = (o != ref).any(dim=3)                                  # [n, y, x]
                    idx = d.nonzero()
                    for n_, y, x in idx[:: max(1, len(idx) // 8)].tolist():
                        blocks.add((n_, y // 16, x // 16))
        print("%-22s beside %-7s: %4d of %4d launches differ from the launch alone%s"
              % (vname, kind, bad, rounds * per, ("   16x16 blocks (image, by, bx): %s" % sorted(blocks)[:8]) if bad else ""), flush=True)
