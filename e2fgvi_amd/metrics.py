"""evaluate.py's image metrics on the device (reference: core/metrics.py:13-56).

``calc_psnr_and_ssim(img1, img2)`` keeps the reference's name and meaning -- PSNR and
``skimage.measure.compare_ssim(data_range=255, multichannel=True, win_size=65)`` of two [H,W,3] frames in [0,255] -- but
takes device tensors (or anything ``torch.as_tensor`` accepts, uploaded first) and also whole batches [N,H,W,3]; the
arithmetic runs in fp64 HIP kernels (csrc/metrics.hip).  VFID (I3D network + checkpoint) is out of scope.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as _L
from .ops import _chk, _ptr, _stream


def psnr_ssim(img1, img2, win_size=65):
    """fp32 device tensors [N,H,W,3] (or [H,W,3]) in [0,255] -> float64 tensor [N,2] of (psnr, ssim) on the device."""
    lib = _L.load()
    a = _chk(img1 if img1.dim() == 4 else img1[None], "img1")
    b = _chk(img2 if img2.dim() == 4 else img2[None], "img2")
    if a.shape != b.shape or a.shape[-1] != 3:
        raise ValueError("image shapes differ or are not [N,H,W,3]: %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    N, H, W, _ = a.shape
    nbytes = lib.e2fgvi_psnr_ssim_workspace(N, H, W)
    if nbytes < 0:
        _L.check(int(nbytes), "psnr_ssim_workspace")
    work = torch.empty(int(nbytes) // 8, dtype=torch.float64, device=a.device)
    out = torch.empty((N, 2), dtype=torch.float64, device=a.device)
    _L.check(lib.e2fgvi_psnr_ssim(_ptr(a), _ptr(b), N, H, W, win_size, _ptr(work), _ptr(out), _stream()), "psnr_ssim")
    return out


def calc_psnr_and_ssim(img1, img2, device="cuda"):
    """core/metrics.py:39-56: img1, img2 ndarray / tensor [H,W,3] in [0,255] -> (psnr, ssim) python floats."""
    a = torch.as_tensor(np.asarray(img1, dtype=np.float32) if not isinstance(img1, torch.Tensor) else img1).float().to(device)
    b = torch.as_tensor(np.asarray(img2, dtype=np.float32) if not isinstance(img2, torch.Tensor) else img2).float().to(device)
    r = psnr_ssim(a.contiguous(), b.contiguous()).cpu()
    return float(r[0, 0]), float(r[0, 1])


def calculate_epe(flow1, flow2):
    """core/metrics.py:13-18 (end point error of two flow fields [N,2,H,W]); a reduction torch already does on device."""
    return torch.sum((flow1 - flow2) ** 2, dim=1).sqrt().view(-1).mean().item()


def evaluate_video(model, frames_u8, masks_u8, neighbor_stride=5, ref_length=10, win_size=65):
    """evaluate.py:75-125 for one video on the device: the sliding-window completion (no mask dilation, no padding --
    the dataset delivers frames at the model's size with binary masks) followed by per-frame PSNR / SSIM of the blended
    frames against the originals.  Returns (psnr[L], ssim[L]) float64 numpy arrays; their means are the video's scores."""
    from . import video
    comp = video.inpaint_video(model, frames_u8, masks_u8, neighbor_stride, ref_length, -1, dilate=False, pad=False,
                               keep_float=True)
    ori = torch.as_tensor(np.ascontiguousarray(frames_u8)).to(comp.device).float()
    r = psnr_ssim(ori, comp, win_size).cpu().numpy()
    return r[:, 0], r[:, 1]
