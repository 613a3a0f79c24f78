"""Probe for cross-stream interference between the SPyNet side stream and conv kernels on the main stream.

Round-1 finding (DESIGN.md "Stream overlap"): with the bf16 conv tiles that carry 2x2 MFMA accumulators per wave (tiles
1/5/6) running on the main stream, `spynet_level_input` on the side stream occasionally produced wrong warped-supp values
in lanes 48-63 of a wave (a packed FMA consumed a just-loaded operand before it had landed).  fp32 kernels, rocBLAS /
hipBLASLt GEMMs and the other bf16 tiles never triggered it.  bf16 mode therefore does not overlap streams; this probe
re-measures the incident rate for both modes and checks that bf16 kernels are self-consistent.

    python tools/overlap_probe.py [trials]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops
from e2fgvi_amd.synth import synth_clip, synth_state_dict
from e2fgvi_amd.engine import Engine

dev = torch.device("cuda:0")
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sd = synth_state_dict("e2fgvi", "stress", 0)
x = synth_clip(1, 3, 240, 432, seed=1, moving=True)[0].to(dev)
engs = {p: Engine(sd, "e2fgvi", dev, precision=p) for p in ("fp32", "bf16")}
ref_idx = torch.tensor([0, 1, 1, 2], dtype=torch.int32, device=dev)
supp_idx = torch.tensor([1, 2, 0, 1], dtype=torch.int32, device=dev)
bt = 3
feat = {"x0": torch.randn(bt, 60, 108, 256, device=dev), "x4": torch.randn(bt, 60, 108, 384, device=dev)}
side = torch.cuda.Stream(device=dev)


def spynet_levels(eng, frames, rec):
    """The level loop of Engine.flows with every level input kept."""
    b, t, c, H, W = frames.shape
    small = ops.resize_bilinear(frames.reshape(b * t, c, H, W), (H // 4, W // 4), True, src_nchw=True, out_ld=4,
                                scale=eng.half, shift=eng.half)
    pyr = [ops.resize_bilinear(small, (64, 128), False, channels=3, out_ld=4, scale=eng.spy_scale, shift=eng.spy_shift)]
    for _ in range(5):
        pyr.append(ops.avgpool2(pyr[-1]))
    pyr = pyr[::-1]
    flow = None
    for lv in range(6):
        inp = ops.spynet_level_input(pyr[lv], ref_idx, supp_idx, flow)
        rec.append((pyr[lv], flow, inp))
        cv = eng.spy[lv]
        y = inp
        for k in range(4):
            y = cv[k]([y], act=ops.ACT_RELU)
        flow = cv[4]([y], residual=inp, res_coff=6)


def incident(prec, tile):
    eng = engs[prec]
    rec = []
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        spynet_levels(eng, x, rec)
    keep = [eng.enc[5]([feat["x0"], feat["x4"]], act=ops.ACT_LRELU, slope=0.2, tile=tile) for _ in range(6)]
    main.wait_stream(side)
    torch.cuda.synchronize()
    for pyr, flow, inp in rec[1:]:
        again = ops.spynet_level_input(pyr, ref_idx, supp_idx, flow)
        if not torch.equal(again, inp):
            return True
    return False


for prec, tile in (("fp32", 0), ("fp32", 228), ("bf16", 2), ("bf16", 5)):
    n = sum(incident(prec, tile) for _ in range(trials))
    print("main stream: %s encoder layer 5 tile %-4d -> side-stream incidents %d / %d" % (prec, tile, n, trials), flush=True)

# self-consistency of the bf16 kernels on one stream (bias + residual + activation epilogue while other waves run MFMAs)
eng = engs["bf16"]
res = torch.randn(bt, 60, 108, 512, device=dev)
for tile in (1, 2, 3, 5, 6):
    outs = [eng.enc[5]([feat["x0"], feat["x4"]], residual=res, act=ops.ACT_LRELU, slope=0.2, tile=tile) for _ in range(trials)]
    torch.cuda.synchronize()
    bad = sum(1 for o in outs[1:] if not torch.equal(o, outs[0]))
    print("bf16 tile %d with residual, single stream: %d / %d launches differ from the first" % (tile, bad, trials - 1), flush=True)
