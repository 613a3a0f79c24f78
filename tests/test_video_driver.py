"""The sliding-window driver (e2fgvi_amd/video.py) against the numpy restatement of test.py's loop."""
import numpy as np
import pytest
import torch

from e2fgvi_amd import video
from oracle import video_ref


def _toy_video(L, h, w, seed=0):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, (h + 2 * L, w + 2 * L, 3)).astype(np.float32)
    k = np.ones((5, 5)) / 25.0
    frames = []
    for i in range(L):
        f = base[i:i + h, 2 * i:2 * i + w]
        frames.append(np.clip(f, 0, 255).astype(np.uint8))
    masks = []
    for i in range(L):
        m = np.zeros((h, w), np.uint8)
        m[h // 4 + i % 3:h // 2 + i % 3, w // 4 + i:w // 2 + i] = 255
        masks.append(m)
    return frames, masks


def _fake_model(x, n_local):
    # deterministic, batch-free stand-in with the InpaintGenerator output convention
    b, t, c, H, W = x.shape
    y = torch.tanh(x.reshape(b * t, c, H, W) * 0.7 + 0.1 * x.mean(dim=(1, 2, 3, 4)).view(1, 1, 1, 1))
    return y, None


def test_dilation_matches_numpy():
    rng = np.random.RandomState(1)
    m = (rng.rand(3, 20, 31) > 0.93)
    ours = video.dilate_cross(torch.from_numpy(m), 4).numpy()
    ref = np.stack([video_ref.dilate_cross_np(x, 4) for x in m]).astype(bool)
    assert (ours == ref).all()


@pytest.mark.parametrize("L,stride,num_ref", [(23, 5, -1), (12, 5, 2), (7, 3, -1)])
def test_driver_logic_matches_reference_loop(L, stride, num_ref):
    frames, masks = _toy_video(L, 50, 70)
    dil = [video_ref.dilate_cross_np(m > 0, 4) for m in masks]
    ref = video_ref.run(lambda x, n: _fake_model(x, n)[0], frames, dil, stride, 10, num_ref)
    out = video.inpaint_video(_fake_model, np.stack(frames), np.stack(masks), stride, 10, num_ref, device=torch.device("cpu"))
    assert out.shape == ref.shape == (L, 50, 70, 3) and out.dtype == np.uint8
    assert np.abs(out.astype(int) - ref.astype(int)).max() == 0


def test_batched_windows_equal_sequential():
    frames, masks = _toy_video(41, 40, 60, seed=5)
    a = video.inpaint_video(_fake_model_batch, np.stack(frames), np.stack(masks), 5, 10, -1, device=torch.device("cpu"))
    b = video.inpaint_video(_fake_model_batch, np.stack(frames), np.stack(masks), 5, 10, -1, device=torch.device("cpu"),
                            batch_windows=3)
    assert (a == b).all()


def _fake_model_batch(x, n_local):
    # per-clip (batch independent) stand-in
    b, t, c, H, W = x.shape
    y = torch.tanh(x * 0.7 + 0.1 * x.mean(dim=(1, 2, 3, 4), keepdim=True)).reshape(b * t, c, H, W)
    return y, None


def test_ref_index_selection():
    assert video.get_ref_index(10, list(range(5, 16)), 40) == [0, 20, 30]
    assert video.get_ref_index(10, list(range(5, 16)), 40, 10, 2) == video_ref.get_ref_index(10, list(range(5, 16)), 40, 10, 2)


@pytest.mark.gpu
def test_driver_on_gpu_matches_cpu_oracle(dev):
    """whole path: HIP model inside the device driver vs oracle model inside the reference loop (uint8 frames)"""
    import importlib
    from e2fgvi_amd.synth import synth_state_dict
    from oracle import e2fgvi_oracle as O
    L, h, w = 11, 60, 100                      # padded to 60x108 by the driver
    frames, masks = _toy_video(L, h, w, seed=3)
    sd = synth_state_dict("e2fgvi_hq", "stress", 0)
    net = importlib.import_module("model.e2fgvi_hq").InpaintGenerator()
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    out = video.inpaint_video(net, np.stack(frames), np.stack(masks), 5, 10, -1)
    dil = [video_ref.dilate_cross_np(m > 0, 4) for m in masks]
    ref = video_ref.run(lambda x, n: O.forward(sd, x, n, "e2fgvi_hq")[0], frames, dil, 5, 10, -1)
    d = np.abs(out.astype(int) - ref.astype(int))
    # |pred error| <= 1e-3 -> at most one grey level where the float lands next to an integer
    assert d.max() <= 1 and (d > 0).mean() < 0.02, (d.max(), (d > 0).mean())
