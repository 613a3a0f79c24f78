"""Pins the CPU restatements of mmcv's modulated_deform_conv2d (oracle/dcn.py, oracle/dcn_ref.c).
No mmcv source or golden vectors exist upstream, so the op is anchored on identities:
zero offsets + unit mask == conv2d; integer offsets == shifted conv; constant sub-pixel offsets ==
grid_sample(zeros, align_corners=True) per tap; and two independent restatements agreeing."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle.dcn import deform_columns, modulated_deform_conv2d


def _close(a, b, rel=3e-6):
    """max abs difference relative to the largest reference magnitude (fp32 summation-order noise)"""
    return (a - b).abs().max().item() <= rel * max(1.0, b.abs().max().item())


def _rand(seed, *shape):
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randn(*shape, generator=g)


def test_zero_offset_unit_mask_is_conv2d():
    x, w, b = _rand(0, 2, 16, 9, 11), _rand(1, 12, 16, 3, 3), _rand(2, 12)
    off = torch.zeros(2, 4 * 18, 9, 11)
    msk = torch.ones(2, 4 * 9, 9, 11)
    out = modulated_deform_conv2d(x, off, msk, w, b, 1, 1, 1, 1, 4)
    assert _close(out, F.conv2d(x, w, b, padding=1))
    out2 = modulated_deform_conv2d(x, torch.zeros(2, 72, 5, 6), torch.ones(2, 36, 5, 6), w, b, 2, 1, 1, 1, 4)
    assert _close(out2, F.conv2d(x, w, b, stride=2, padding=1))


def test_mask_scales_taps():
    x, w = _rand(3, 1, 8, 6, 7), _rand(4, 4, 8, 3, 3)
    off = torch.zeros(1, 18, 6, 7)
    msk = torch.full((1, 9, 6, 7), 0.25)
    out = modulated_deform_conv2d(x, off, msk, w, None, 1, 1, 1, 1, 1)
    assert _close(out, 0.25 * F.conv2d(x, w, None, padding=1))


def test_integer_offset_is_shifted_conv():
    x, w = _rand(5, 1, 8, 10, 12), _rand(6, 4, 8, 3, 3)
    dy, dx = 2, -3
    off = torch.zeros(1, 18, 10, 12)
    off[:, 0::2] = dy
    off[:, 1::2] = dx
    out = modulated_deform_conv2d(x, off, torch.ones(1, 9, 10, 12), w, None, 1, 1, 1, 1, 1)
    # sampling x at (y+dy, x+dx) with zeros outside == conv over the shifted, zero-filled image
    xs = torch.zeros_like(x)
    xs[:, :, :10 - dy, 3:] = x[:, :, dy:, :12 - 3]
    ref = F.conv2d(F.pad(x, (4, 4, 4, 4)), w, None)          # full correlation on a padded canvas
    ref = ref[:, :, 3 + dy:3 + dy + 10, 3 + dx:3 + dx + 12]
    assert _close(out, ref)


def test_constant_subpixel_offset_is_grid_sample():
    x = _rand(7, 1, 4, 9, 10)
    dy, dx = 0.3, -1.6
    off = torch.zeros(1, 18, 9, 10)
    off[:, 0::2] = dy
    off[:, 1::2] = dx
    cols, _ = deform_columns(x, off, torch.ones(1, 9, 9, 10), (3, 3), 1, 1, 1, 1)
    H, W = 9, 10
    for k in range(9):
        i, j = divmod(k, 3)
        ys = torch.arange(H).view(H, 1).float() - 1 + i + dy
        xs = torch.arange(W).view(1, W).float() - 1 + j + dx
        grid = torch.stack((2 * xs.expand(H, W) / (W - 1) - 1, 2 * ys.expand(H, W) / (H - 1) - 1), -1)[None]
        ref = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
        assert (cols[:, :, k].view(1, 4, H, W) - ref).abs().max() < 1e-5


@pytest.mark.parametrize("mag,stride,pad,dil", [(0.0, 1, 1, 1), (2.5, 1, 1, 1), (30.0, 1, 1, 1), (3.0, 2, 1, 1), (3.0, 1, 2, 2)])
def test_torch_and_c_restatements_agree(mag, stride, pad, dil):
    from oracle.dcn_c import modulated_deform_conv2d_c
    N, C, H, W, Co, dg = 2, 32, 11, 13, 10, 2
    Ho = (H + 2 * pad - (dil * 2 + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * 2 + 1)) // stride + 1
    x, w, b = _rand(8, N, C, H, W), _rand(9, Co, C, 3, 3) / 10, _rand(10, Co)
    off = _rand(11, N, dg * 18, Ho, Wo) * mag
    msk = torch.sigmoid(_rand(12, N, dg * 9, Ho, Wo))
    a = modulated_deform_conv2d(x, off, msk, w, b, stride, pad, dil, 1, dg)
    c = modulated_deform_conv2d_c(x, off, msk, w, b, stride, pad, dil, 1, dg)
    assert a.shape == c.shape == (N, Co, Ho, Wo)
    assert _close(a, c, 1e-5)


def test_exactly_on_border_and_outside():
    """py == -1 / py == H are outside (value 0); py in (H-1, H) uses only the last row (mmcv guard semantics)."""
    x = torch.ones(1, 16, 4, 4)
    w = torch.zeros(1, 16, 1, 1)
    w[0, 0] = 1
    for dy, expect in ((-1.0, 0.0), (-0.5, 0.5), (0.0, 1.0), (3.5, 0.5), (4.0, 0.0)):
        off = torch.zeros(1, 2, 4, 4)
        off[:, 0] = dy - torch.arange(4).float().view(4, 1)     # sample row dy for every output row
        out = modulated_deform_conv2d(x, off, torch.ones(1, 1, 4, 4), w, None, 1, 0, 1, 1, 1)
        assert abs(out[0, 0, 0, 0].item() - expect) < 1e-6, (dy, out[0, 0, 0, 0].item())
