"""SURVEY.md 8(f) rank 2 with what is on disk: the reference's demo clip (examples/tennis, 25 frames + irregular masks,
stored by tests/golden/make_tennis_golden.py together with the output of the REAL reference's test.py loop on CPU)
through e2fgvi_amd.video.inpaint_video on the MI355X -- real frames, real masks (PNG -> uint8, ~13 % coverage after
dilation), PIL-NEAREST + cross-dilation mask preprocessing, sliding windows with reference frames, compositing and
0.5/0.5 blending.  Weights are the deterministic 'stress' set (the released checkpoints cannot be fetched)."""
import importlib
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "tennis25.npz")


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["e2fgvi", "e2fgvi_hq"])
@pytest.mark.parametrize("batch_windows", [1, 2])
def test_tennis_clip_matches_reference_loop(dev, model, batch_windows):
    from e2fgvi_amd import video
    from e2fgvi_amd.synth import synth_state_dict
    z = np.load(GOLD)
    L, h, w, sub = [int(v) for v in z["meta"]]
    net = importlib.import_module("model." + model).InpaintGenerator()
    net.load_state_dict(synth_state_dict(model, "stress", 0))
    net = net.to(dev).eval()
    out = video.inpaint_video(net, z["frames"], z["masks_raw"], batch_windows=batch_windows)
    assert out.shape == (L, h, w, 3) and out.dtype == np.uint8
    d = np.abs(out[:, ::sub, ::sub].astype(int) - z[model + "_sub"].astype(int))
    # |pred error| <= 1e-3 in [-1,1] = 0.13 grey levels: at most one level where the float lands next to an integer
    print("tennis %s: max grey-level diff %d, differing samples %.4f" % (model, d.max(), (d > 0).mean()))
    assert d.max() <= 1 and (d > 0).mean() < 0.02
    fm = out.reshape(L, -1).astype(np.float64).mean(1)
    assert np.abs(fm - z[model + "_frame_mean"]).max() < 0.02
    # outside the (dilated) masks the frames are the input, bit-exact
    masks = video.prepare_masks(z["masks_raw"], (h, w), dev).cpu().numpy().astype(bool)
    assert (out[~masks] == z["frames"][~masks]).all()
