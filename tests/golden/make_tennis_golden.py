"""Golden fixture for SURVEY.md 8(f) rank 2: the reference's demo clip through the reference's own loop.

    python tests/golden/make_tennis_golden.py          (build container only: needs /root/reference)

Reads the first N_FRAMES frames of /root/reference/examples/tennis (432x240 RGB PNG) and their masks
(examples/tennis_mask, 'L' PNG, 0/255, ~9 % irregular coverage), pre-processes the masks exactly like
test.py:56-69 (PIL NEAREST resize to the frame size, > 0, 4x cross dilation), and runs the reference's
sliding-window loop test.py:129-179 (numpy restatement oracle/video_ref.py, pinned line by line) around the
REAL reference InpaintGenerator (imported read-only through oracle/ref_import.py, CPU fp32) carrying the
deterministic 'stress' weights of e2fgvi_amd.synth -- the released checkpoints cannot be fetched (no network).

Stored: the uint8 input frames + raw masks (they have to travel: the GPU box has no /root/reference) and, per model,
a strided sub-sample of the uint8 composited result plus per-frame means.  tests/test_tennis.py feeds the stored
inputs to e2fgvi_amd.video.inpaint_video on the MI355X and requires <= 1 grey level.
"""
import os
import sys

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from e2fgvi_amd.synth import synth_state_dict  # noqa: E402
from oracle import ref_import, video_ref  # noqa: E402

N_FRAMES = 25
SUB = 4
SRC = os.path.join(ref_import.REFERENCE_ROOT, "examples")


def read_inputs():
    names = sorted(os.listdir(os.path.join(SRC, "tennis")))[:N_FRAMES]
    frames = [np.array(Image.open(os.path.join(SRC, "tennis", n)).convert("RGB")) for n in names]
    mnames = sorted(os.listdir(os.path.join(SRC, "tennis_mask")))[:N_FRAMES]
    raw = [np.array(Image.open(os.path.join(SRC, "tennis_mask", n)).convert("L")) for n in mnames]
    return np.stack(frames).astype(np.uint8), np.stack(raw).astype(np.uint8)


def reference_masks(raw, size_wh):
    """test.py:56-69 with PIL itself and the cv2.dilate restatement"""
    out = []
    for m in raw:
        m = Image.fromarray(m).resize(size_wh, Image.NEAREST)
        m = (np.array(m.convert("L")) > 0).astype(np.uint8)
        out.append(video_ref.dilate_cross_np(m, 4))
    return out


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    frames, raw = read_inputs()
    L, h, w, _ = frames.shape
    masks = reference_masks(raw, (w, h))
    store = dict(frames=frames, masks_raw=raw, meta=np.array([L, h, w, SUB]))
    for model in ("e2fgvi", "e2fgvi_hq"):
        net = ref_import.build_reference_model(model, synth_state_dict(model, "stress", 0))

        def fn(x, n_local):
            with torch.no_grad():
                return net(x, n_local)[0]

        comp = video_ref.run(fn, [f for f in frames], masks)          # uint8 [L,h,w,3]
        store[model + "_sub"] = comp[:, ::SUB, ::SUB].copy()
        store[model + "_frame_mean"] = comp.reshape(L, -1).astype(np.float64).mean(1)
        hole = np.stack(masks).astype(bool)
        store[model + "_hole_mean"] = np.array([comp[i][hole[i]].astype(np.float64).mean() for i in range(L)])
        print(model, comp.shape, "hole coverage %.3f" % hole.mean(), "mean in holes %.2f" % store[model + "_hole_mean"].mean(),
              flush=True)
    np.savez_compressed(os.path.join(HERE, "tennis25.npz"), **store)


if __name__ == "__main__":
    main()
