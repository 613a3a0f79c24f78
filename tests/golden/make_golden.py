"""Generate the golden fixtures from the REAL reference (run in the build container only).

    python tests/golden/make_golden.py

Imports /root/reference/model/*.py read-only through oracle/ref_import.py (mmcv shim), loads the
deterministic synthetic weights of e2fgvi_amd.synth into the reference InpaintGenerator, runs its CPU
forward on the deterministic synthetic clips and stores strided sub-samples of the outputs plus
whole-tensor statistics.  The fixtures travel to the GPU box; the reference does not.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from e2fgvi_amd.synth import synth_clip, synth_state_dict  # noqa: E402
from oracle import ref_import  # noqa: E402

# name, model, weights kind, (H, W), t, l_t, clips, clip seed
CASES = [
    ("g1_e2fgvi_default_t5", "e2fgvi", "default", (240, 432), 5, 5, 1, 11),
    ("g2_e2fgvi_stress_t4_lt3", "e2fgvi", "stress", (240, 432), 4, 3, 1, 12),
    ("g3_hq_stress_120x216_t4_lt3", "e2fgvi_hq", "stress", (120, 216), 4, 3, 1, 13),
    ("g4_hq_default_60x108_t3_lt2_b2", "e2fgvi_hq", "default", (60, 108), 3, 2, 2, 14),
    # BASELINE.json configs 4 / 5 resolutions (720x1280 -> 720x1296, 1080x1920 -> 1080x1944 after test.py's padding):
    # 12x12 / 18x18 window grids, i.e. fully valid pooled neighbourhoods (210 keys per frame) and circular wrap-around
    # over the whole grid -- branches the small fixtures never reach.  Few frames keep the CPU run to minutes.
    ("g5_hq_stress_720x1296_t3_lt2", "e2fgvi_hq", "stress", (720, 1296), 3, 2, 1, 15),
    ("g6_hq_stress_1080x1944_t2_lt2", "e2fgvi_hq", "stress", (1080, 1944), 2, 2, 1, 16),
    # round 4: the headline configuration itself (BASELINE.json configs[1]: 432x240, T = 10, all frames local) and SURVEY.md
    # 8(d)'s second split (T = 10, 5 local + 5 reference frames, configs/train_e2fgvi.json:9-10) from the REAL reference,
    # stress weights -- the kernels the engine selects only at full size (10-frame batches) are then checked against the
    # reference's own output, not only against the port
    ("g7_e2fgvi_stress_t10_lt10", "e2fgvi", "stress", (240, 432), 10, 10, 1, 17),
    ("g8_e2fgvi_stress_t10_lt5", "e2fgvi", "stress", (240, 432), 10, 5, 1, 18),
    # round 6: the HQ model at the clip lengths its bench lines are timed at (BASELINE.json configs[3]: 720x1296, T = l_t = 10;
    # configs[4]'s resolution with an 8-step recurrence -- T = 20 at 1080p needs > 60 GB for the reference's materialised
    # attention scores), so the bf16 data path's error growth over the recurrent chain is measured against the REAL reference
    ("g9_hq_stress_720x1296_t10_lt10", "e2fgvi_hq", "stress", (720, 1296), 10, 10, 1, 19),
    ("g10_hq_stress_1080x1944_t8_lt8", "e2fgvi_hq", "stress", (1080, 1944), 8, 8, 1, 20),
    # the "peaked" weights (synth.py: sharp attention, saturated DCN offsets / masks, non-uniform pooling): the stand-in
    # for trained weights while the released checkpoints are unreachable
    ("g11_e2fgvi_peaked_t10_lt10", "e2fgvi", "peaked", (240, 432), 10, 10, 1, 21),
    ("g12_hq_peaked_240x432_t6_lt4", "e2fgvi_hq", "peaked", (240, 432), 6, 4, 1, 22),
    ("g13_hq_peaked_720x1296_t4_lt3", "e2fgvi_hq", "peaked", (720, 1296), 4, 3, 1, 23),
    # bench.py's own configs[3] clip (synth_clip seed 0, smooth=False, static box; default-init weights): the frames of the
    # TIMED engine of that secondary line are compared with this fixture (its `parity` field)
    ("g14_hq_default_720x1296_t10_lt10_benchclip", "e2fgvi_hq", "default", (720, 1296), 10, 10, 1, 0),
]
BENCH_CLIPS = {"g14_hq_default_720x1296_t10_lt10_benchclip"}       # synth_clip(..., smooth=False, moving=False)
OUT_STRIDE, FLOW_STRIDE = 8, 4


def stats(t):
    t = t.double()
    return np.array([t.mean().item(), t.abs().mean().item(), t.pow(2).mean().sqrt().item(), t.abs().max().item()])


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    only = set(sys.argv[1:])
    for name, model, kind, (H, W), t, lt, b, seed in CASES:
        if only and name not in only:
            continue
        sd = synth_state_dict(model, kind, 0)
        net = ref_import.build_reference_model(model, sd)
        x, _ = synth_clip(b, t, H, W, seed=seed, smooth=False) if name in BENCH_CLIPS else synth_clip(b, t, H, W, seed=seed, moving=True)
        with torch.no_grad():
            out, (ff, fb) = net(x, lt)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            out_sub=out[:, :, ::OUT_STRIDE, ::OUT_STRIDE].numpy(), out_stats=stats(out),
            out_frame_mean=out.double().mean(dim=(1, 2, 3)).numpy(),
            flow_fwd_sub=ff[..., ::FLOW_STRIDE, ::FLOW_STRIDE].numpy(), flow_bwd_sub=fb[..., ::FLOW_STRIDE, ::FLOW_STRIDE].numpy(),
            flow_fwd_stats=stats(ff), flow_bwd_stats=stats(fb),
            meta=np.array([H, W, t, lt, b, seed, OUT_STRIDE, FLOW_STRIDE]), model=model, kind=kind)
        print(name, tuple(out.shape), "out stats", stats(out), flush=True)


if __name__ == "__main__":
    main()
