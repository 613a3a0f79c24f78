"""The algebraic identities behind the round-2 restructurings, checked on the CPU with plain torch (no HIP involved): each
product-side rewrite replaces a reference expression by an equal one, and these tests pin the equality itself.

  * FusionFeedForward (model/modules/tfocal_transformer.py:75-99): Linear(1960 -> 512) applied to nn.Unfold(7, stride 3,
    padding 3) of the folded tensor == a 7x7 / stride-3 / pad-3 convolution of that tensor with the same weights viewed
    [512, 40, 7, 7] (engine_x.py runs it that way: no unfold kernel, no [rows, 1960] tensor);
  * GELU applied before the unfold == after it (the unfold is a gather with zero padding and GELU(0) = 0);
  * the decoder's Conv2d(64, 3, 3, padding=1) (model/e2fgvi.py:99-103) == one [pixels x 64] x [64 x 27] product followed
    by a shifted nine-term sum (csrc/conv_tail.hip)."""
import torch
import torch.nn.functional as F


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def test_ffn_second_linear_is_a_strided_conv_of_the_folded_tensor():
    g = _gen(0)
    b, H, W, hd = 2, 18, 27, 40                       # 6 x 9 tokens of 7x7 patches, stride 3, padding 3
    y = torch.randn(b, hd, H, W, generator=g, dtype=torch.float64)
    w2 = torch.randn(512, hd * 49, generator=g, dtype=torch.float64) / 44
    b2 = torch.randn(512, generator=g, dtype=torch.float64)
    unf = F.unfold(y, (7, 7), stride=3, padding=3)                       # [b, 1960, n_tokens], channel-major (c, ky, kx)
    ref = F.linear(unf.transpose(1, 2), w2, b2)                          # tfocal_transformer.py:95-97
    conv = F.conv2d(y, w2.view(512, hd, 7, 7), b2, stride=3, padding=3)  # [b, 512, 6, 9]
    assert conv.shape[2:] == (6, 9)
    assert torch.allclose(conv.flatten(2).transpose(1, 2), ref, rtol=0, atol=1e-12)


def test_gelu_commutes_with_the_zero_padded_unfold():
    y = torch.randn(1, 40, 18, 27, generator=_gen(1), dtype=torch.float64)
    a = F.gelu(F.unfold(y, (7, 7), stride=3, padding=3))
    b = F.unfold(F.gelu(y), (7, 7), stride=3, padding=3)
    # (allclose, not equal: torch's vectorised and scalar-tail erf differ in the last bit depending on the element's position)
    assert torch.allclose(a, b, rtol=0, atol=1e-15) and float(F.gelu(torch.zeros(1, dtype=torch.float64))) == 0.0


def test_three_channel_conv_is_one_gemm_plus_a_shifted_sum():
    g = _gen(2)
    N, H, W = 2, 11, 13
    x = torch.randn(N, 64, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(3, 64, 3, 3, generator=g, dtype=torch.float64) / 24
    bias = torch.randn(3, generator=g, dtype=torch.float64)
    ref = F.conv2d(x, w, bias, padding=1)
    # Z[q][(tap, co)] = sum_c x[q][c] w[co][c][tap]: ONE product over the channels, computed on the zero-padded image
    xp = F.pad(x, (1, 1, 1, 1)).permute(0, 2, 3, 1)                       # [N, H+2, W+2, 64]
    B = w.permute(1, 2, 3, 0).reshape(64, 27)                            # columns (ky, kx, co)
    Z = (xp @ B).view(N, H + 2, W + 2, 3, 3, 3)
    out = bias.view(1, 1, 1, 3).expand(N, H, W, 3).clone()
    for ky in range(3):
        for kx in range(3):
            out = out + Z[:, ky:ky + H, kx:kx + W, ky, kx, :]
    assert torch.allclose(out.permute(0, 3, 1, 2), ref, rtol=0, atol=1e-12)


def test_planar16_layout_is_a_permutation_of_nhwc():
    x = torch.arange(2 * 3 * 5 * 32, dtype=torch.float32).view(2, 3, 5, 32)
    p = x.view(2, 3, 5, 2, 16).permute(3, 0, 1, 2, 4).contiguous()       # [C/16, N, H, W, 16] (e2fgvi_nhwc_to_planar16)
    g, n, yy, xx, c = 1, 1, 2, 4, 7
    assert p[g, n, yy, xx, c] == x[n, yy, xx, g * 16 + c]
    assert torch.equal(p.permute(1, 2, 3, 0, 4).reshape(2, 3, 5, 32), x)


def test_softcomp_fold_is_nine_phase_convolutions_of_the_token_grid():
    """Round 3 (ops.SoftCompGather): nn.Fold(7x7, stride 3, padding 3) of Linear(hidden -> 49 C) -- SoftComp,
    model/modules/tfocal_transformer.py:49-72 -- equals nine small convolutions over the token grid, phase (py, px) writing the
    pixels (3 ty + py, 3 tx + px): kernel rows ky read token rows ty - pad + ky with pad = 1 (py = 0: taps ki = 6, 3, 0) or
    pad = 0 (py = 1, 2: taps ki = py + 3, py), zero rows outside the grid; the Linear's bias folds to a per-pixel image."""
    g = _gen(3)
    F_, fh, fw, hid, C = 2, 5, 7, 24, 6
    w = torch.randn(49 * C, hid, generator=g, dtype=torch.float64) / 8            # row c*49 + ki*7 + kj
    b = torch.randn(49 * C, generator=g, dtype=torch.float64)
    tok = torch.randn(F_, fh, fw, hid, generator=g, dtype=torch.float64)
    emb = F.linear(tok.view(F_, fh * fw, hid), w, b)
    ref = F.fold(emb.permute(0, 2, 1), output_size=(3 * fh, 3 * fw), kernel_size=(7, 7), stride=(3, 3), padding=(3, 3))
    bias_img = F.fold(b.view(1, 49 * C, 1).expand(1, 49 * C, fh * fw), output_size=(3 * fh, 3 * fw), kernel_size=(7, 7),
                      stride=(3, 3), padding=(3, 3))
    out = bias_img.expand(F_, C, 3 * fh, 3 * fw).clone()
    w4 = w.view(C, 7, 7, hid)
    x = tok.permute(0, 3, 1, 2)                                                    # [F, hid, fh, fw]
    for py in range(3):
        ky_taps, pad_y = ([6, 3, 0], 1) if py == 0 else ([py + 3, py], 0)
        for px in range(3):
            kx_taps, pad_x = ([6, 3, 0], 1) if px == 0 else ([px + 3, px], 0)
            wp = w4[:, ky_taps][:, :, kx_taps].permute(0, 3, 1, 2)                 # [C, hid, kh, kw]
            kh, kw = len(ky_taps), len(kx_taps)
            # explicit output grid fh x fw: pad_y rows above / pad_x columns left, whatever else the kernel reaches is zero
            xp = F.pad(x, (pad_x, kw - 1 - pad_x, pad_y, kh - 1 - pad_y))
            out[:, :, py::3, px::3] += F.conv2d(xp, wp)
    assert torch.allclose(out, ref, rtol=0, atol=1e-11)
