"""CPU restatement of evaluate.py's per-frame metrics (TEST INFRASTRUCTURE): core/metrics.py:20-56.

PSNR is the reference's own numpy formula.  SSIM: the reference calls ``skimage.measure.compare_ssim(img1, img2,
data_range=255, multichannel=True, win_size=65)`` (scikit-image 0.16, environment.yml); scikit-image is not installed
here and not under /root/reference, so its published algorithm (Wang et al. 2004 as implemented in
skimage/metrics/_structural_similarity.py) is restated on top of ``scipy.ndimage.uniform_filter`` -- the very filter
skimage calls.  PARITY UNPINNED against skimage itself; anchored by a brute-force window evaluation in
tests/test_oracle_metrics.py.
"""
import numpy as np
from scipy.ndimage import uniform_filter


def calculate_psnr(img1, img2):                                   # core/metrics.py:20-36
    mse = np.mean((img1 - img2) ** 2)
    if mse == 0:
        return float("inf")
    return 20.0 * np.log10(255.0 / np.sqrt(mse))


def _ssim_channel(X, Y, win_size, data_range, K1=0.01, K2=0.03):
    NP = win_size ** 2
    cov_norm = NP / (NP - 1)                                      # use_sample_covariance=True
    ux = uniform_filter(X, size=win_size)
    uy = uniform_filter(Y, size=win_size)
    uxx = uniform_filter(X * X, size=win_size)
    uyy = uniform_filter(Y * Y, size=win_size)
    uxy = uniform_filter(X * Y, size=win_size)
    vx = cov_norm * (uxx - ux * ux)
    vy = cov_norm * (uyy - uy * uy)
    vxy = cov_norm * (uxy - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win_size - 1) // 2
    return S[pad:S.shape[0] - pad, pad:S.shape[1] - pad].mean()


def compare_ssim(img1, img2, data_range=255, win_size=65):
    """multichannel=True: mean over the last axis of the per-channel mean SSIM"""
    return float(np.mean([_ssim_channel(img1[..., c], img2[..., c], win_size, data_range) for c in range(img1.shape[-1])]))


def calc_psnr_and_ssim(img1, img2, win_size=65):                  # core/metrics.py:39-56
    img1 = img1.astype(np.float64)
    img2 = img2.astype(np.float64)
    return calculate_psnr(img1, img2), compare_ssim(img1, img2, 255, win_size)
