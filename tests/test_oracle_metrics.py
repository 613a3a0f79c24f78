"""The metrics oracle (oracle/metrics_ref.py) against a brute-force evaluation of the SSIM definition, and the device
kernels (csrc/metrics.hip) against the oracle."""
import numpy as np
import pytest
import torch

from oracle import metrics_ref


def _pair(h, w, seed, noise=12.0):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 127 + 80 * np.sin(xx / 9.0 + seed) * np.cos(yy / 7.0) + rng.randn(h, w) * 10
    a = np.clip(np.stack([base, base * 0.8 + 20, 255 - base], 2), 0, 255).astype(np.uint8)
    b = np.clip(a.astype(np.float64) + rng.randn(h, w, 3) * noise, 0, 255).astype(np.uint8)
    return a, b


def test_ssim_oracle_equals_bruteforce_definition():
    """uniform 9x9 windows evaluated one by one == the filter-based restatement (interior only, like the crop)"""
    a, b = _pair(31, 40, 0)
    win, pad = 9, 4
    X, Y = a.astype(np.float64), b.astype(np.float64)
    NP = win * win
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    vals = []
    for c in range(3):
        for y in range(pad, 31 - pad):
            for x in range(pad, 40 - pad):
                wx = X[y - pad:y + pad + 1, x - pad:x + pad + 1, c].ravel()
                wy = Y[y - pad:y + pad + 1, x - pad:x + pad + 1, c].ravel()
                ux, uy = wx.mean(), wy.mean()
                vx, vy = wx.var(ddof=1), wy.var(ddof=1)
                vxy = ((wx - ux) * (wy - uy)).sum() / (NP - 1)
                vals.append(((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2)))
    brute = float(np.mean(vals))
    assert abs(metrics_ref.compare_ssim(X, Y, 255, win) - brute) < 1e-10


def test_psnr_oracle():
    a, b = _pair(20, 30, 1)
    assert metrics_ref.calc_psnr_and_ssim(a, a, 9) == (float("inf"), 1.0)
    p, s = metrics_ref.calc_psnr_and_ssim(a, b, 9)
    assert 20 < p < 40 and 0.2 < s < 1.0
    flat = np.full((20, 30, 3), 10, np.uint8)
    assert abs(metrics_ref.calculate_psnr(flat.astype(float), flat.astype(float) + 5) - 20 * np.log10(255 / 5)) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("hw,win", [((240, 432), 65), ((70, 90), 65), ((66, 65), 65), ((31, 40), 9)])
def test_psnr_ssim_kernel_matches_oracle(dev, hw, win):
    from e2fgvi_amd import metrics
    pairs = [_pair(hw[0], hw[1], s) for s in range(3)]
    # the third pair carries .5 values like the blended frames of evaluate.py
    a = np.stack([p[0] for p in pairs]).astype(np.float32)
    b = np.stack([p[1] for p in pairs]).astype(np.float32)
    b[2] = b[2] * 0.5 + a[2] * 0.5
    got = metrics.psnr_ssim(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), win).cpu().numpy()
    for i in range(3):
        p, s = metrics_ref.calc_psnr_and_ssim(a[i], b[i], win)
        assert abs(got[i, 0] - p) <= 1e-9 * abs(p) and abs(got[i, 1] - s) <= 1e-9, (i, got[i], p, s)


@pytest.mark.gpu
def test_calc_psnr_and_ssim_api(dev):
    from e2fgvi_amd import metrics
    a, b = _pair(240, 432, 7)
    p, s = metrics.calc_psnr_and_ssim(a, b)
    pr, sr = metrics_ref.calc_psnr_and_ssim(a, b)
    assert abs(p - pr) < 1e-9 * pr and abs(s - sr) < 1e-9
    assert metrics.calc_psnr_and_ssim(a, a) == (float("inf"), 1.0)
    with pytest.raises(Exception):
        metrics.psnr_ssim(torch.zeros(1, 40, 40, 3, device=dev), torch.zeros(1, 40, 40, 3, device=dev))   # window > image


@pytest.mark.gpu
def test_evaluate_video_matches_reference_loop(dev):
    """evaluate.py's per-video loop: completion (no dilation / padding) + per-frame PSNR / SSIM of the blended float frames,
    device path vs numpy reference loop + metrics oracle, with a stand-in model on the CPU in both"""
    from e2fgvi_amd import metrics
    from oracle import video_ref
    rng = np.random.RandomState(5)
    L, h, w = 13, 70, 90
    frames = [np.clip(rng.randint(0, 256, (h, w, 3)) * 0.3 + 90 + 20 * i, 0, 255).astype(np.uint8) for i in range(L)]
    masks = [np.zeros((h, w), np.uint8) for _ in range(L)]
    for i, m in enumerate(masks):
        m[20 + i:45 + i, 30:60] = 1
    fake = lambda x, n: torch.tanh(x.reshape(-1, 3, h, w) * 0.6)
    ref_comp = video_ref.run(fake, frames, masks, 5, 10, -1, pad=False, as_float=True)
    ref = np.array([metrics_ref.calc_psnr_and_ssim(o, c, 65) for o, c in zip(frames, ref_comp)])
    psnr, ssim = metrics.evaluate_video(lambda x, n: (fake(x.cpu(), n).to(dev), None), np.stack(frames), np.stack(masks))
    assert np.abs(psnr - ref[:, 0]).max() <= 1e-9 * ref[:, 0].max() and np.abs(ssim - ref[:, 1]).max() <= 1e-9
