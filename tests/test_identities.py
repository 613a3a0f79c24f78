"""The algebraic identities behind the round-2 restructurings, checked on the CPU with plain torch (no HIP involved): each
product-side rewrite replaces a reference expression by an equal one, and these tests pin the equality itself.

  * FusionFeedForward (model/modules/tfocal_transformer.py:75-99): Linear(1960 -> 512) applied to nn.Unfold(7, stride 3,
    padding 3) of the folded tensor == a 7x7 / stride-3 / pad-3 convolution of that tensor with the same weights viewed
    [512, 40, 7, 7] (engine_x.py runs it that way: no unfold kernel, no [rows, 1960] tensor);
  * GELU applied before the unfold == after it (the unfold is a gather with zero padding and GELU(0) = 0);
  * the decoder's Conv2d(64, 3, 3, padding=1) (model/e2fgvi.py:99-103) == one [pixels x 64] x [64 x 27] product followed
    by a shifted nine-term sum (csrc/conv_tail.hip)."""
import pytest
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


def _split3(x, rne=False):
    """the kernels' splits, returned as fp64 values of the three bf16 numbers (+ the two fp32 remainders):
    rne=False  what the kernels and the weight packers do (csrc/common.h e2_split2): hi = x with its low 16 bits cleared, mid = the
               same of the exact remainder, lo = the rest;
    rne=True   the round-to-nearest variant measured and rejected in round 5 (hi = RNE_bf16(x), mid = RNE_bf16(x - hi), lo = the
               rest; v_cvt_pk_bf16_f32 + v_dot2c_f32_bf16: fewer but slower instructions, profiles/r05_split_rne_ab.txt) -- kept
               here because its exactness argument is the one tools/probe/split_probe.hip checks on the chip"""
    import numpy as np
    x = np.asarray(x, dtype=np.float32)

    def piece(v):
        u = v.view(np.uint32)
        if not rne:
            return (u & np.uint32(0xFFFF0000)).view(np.float32)
        # round to nearest even on the upper 16 bits (v_cvt_pk_bf16_f32)
        r = (u.astype(np.uint64) + np.uint64(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1)).astype(np.uint64)) & np.uint64(0xFFFF0000)
        return r.astype(np.uint32).view(np.float32)
    hi = piece(x)
    r = (x - hi).astype(np.float32)
    mid = piece(r)
    r2 = (r - mid).astype(np.float32)
    lo = piece(r2)
    return hi.astype(np.float64), mid.astype(np.float64), lo.astype(np.float64), r, r2


@pytest.mark.parametrize("rne", [False, True], ids=["truncating (the kernels)", "round-to-nearest (rejected variant)"])
def test_three_way_bf16_split_is_exact_and_six_products_are_fp32_level(rne):
    """the arithmetic behind the split-operand ("x3") kernels, on the CPU, for the split the kernels use and for the
    round-to-nearest variant of round 5's probe: (1) hi + mid + lo == x bit for bit and every piece is
    a bf16 number (16 low bits zero), for random, tiny, huge, negative and few-bit values -- both remainders are exact fp32
    differences and the last one has at most 8 significant bits; (2) the six partial products the kernels issue differ from
    the exact product by less than 2^-21 |ab| (the three dropped ones: mid*lo, lo*mid, lo*lo), i.e. below one fp32 rounding
    of the product; (3) a K = 512 dot product accumulated in fp32 from those six terms is as close to the fp64 result as the
    plain fp32 dot product is"""
    import numpy as np
    rng = np.random.default_rng(5)
    # (values below 2^-110: the last piece becomes an fp32 denormal and loses bits -- an absolute error < 2^-133, checked apart)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32), (rng.standard_normal(2000) * 1e-25).astype(np.float32),
                        (rng.standard_normal(2000) * 1e30).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 3.0, 1 + 2.0 ** -23, -(1 + 2.0 ** -16), 65504.0, 2.0 ** -126, 2.0 ** -120,
                                  1 - 2.0 ** -24, 255.0, 256.5], dtype=np.float32)])
    hi, mid, lo, r, r2 = _split3(x, rne)
    assert np.array_equal(hi + mid + lo, x.astype(np.float64)), "the three pieces do not sum to the value"
    assert np.array_equal(r.astype(np.float64), x.astype(np.float64) - hi) and np.array_equal(r2.astype(np.float64), r.astype(np.float64) - mid), \
        "a remainder was rounded"
    assert np.array_equal(lo, r2.astype(np.float64)), "the last remainder does not fit a bf16 number"
    for piece in (hi, mid, lo):
        assert not (piece.astype(np.float32).view(np.uint32) & np.uint32(0xFFFF)).any()
    tiny = (rng.standard_normal(5000) * 1e-36).astype(np.float32)
    th, tm, tl, _, _ = _split3(tiny, rne)
    assert np.max(np.abs(th + tm + tl - tiny.astype(np.float64))) < 2.0 ** -133
    a = rng.standard_normal(100000).astype(np.float32)
    b = rng.standard_normal(100000).astype(np.float32)
    ah, am, al, _, _ = _split3(a, rne)
    bh, bm, bl, _, _ = _split3(b)
    six = al * bh + ah * bl + am * bm + am * bh + ah * bm + ah * bh            # exact in fp64: each term has <= 16 significant bits
    exact = a.astype(np.float64) * b.astype(np.float64)
    assert np.max(np.abs(six - exact) / np.abs(exact)) < 2.0 ** -21
    # a dot product: fp32 accumulation of the six-term products against a plain fp32 dot product, both against fp64
    K, n = 512, 2000
    A = rng.standard_normal((n, K)).astype(np.float32)
    B = rng.standard_normal((K,)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    plain = np.zeros(n, np.float32)
    split = np.zeros(n, np.float32)
    Ah, Am, Al, _, _ = _split3(A, rne)
    Bh, Bm, Bl, _, _ = _split3(B)
    for k in range(K):
        plain = (plain + A[:, k] * B[k]).astype(np.float32)
        for t in (Al[:, k] * Bh[k], Ah[:, k] * Bl[k], Am[:, k] * Bm[k], Am[:, k] * Bh[k], Ah[:, k] * Bm[k], Ah[:, k] * Bh[k]):
            split = (split + t.astype(np.float32)).astype(np.float32)         # each term is exact in fp32
    rms = np.sqrt(np.mean(ref ** 2))
    e_plain, e_split = np.max(np.abs(plain - ref)) / rms, np.max(np.abs(split - ref)) / rms
    # (one fp32 rounding per TERM here -- six per product, the worst case; the MFMA rounds once per 16-product instruction and
    #  measures at or below the plain fp32 kernel on the GPU, tests/test_gpu_x3.py)
    assert e_split < 8 * np.sqrt(K) * 2.0 ** -24 and e_split < 4 * e_plain + 1e-7, (e_plain, e_split)


@pytest.mark.parametrize("nparts", [2, 3])
def test_propagation_split_is_the_whole_layer(nparts):
    """engine.split_prop_weights: conv_offset.0 over cat(cond_n1, cur, cond_n2, flows) and backbone.0 over
    cat(cur[, other direction], feat_prop) equal the recurrent-part convolution plus the non-recurrent one (bias once) -- the
    identity the engine's side-stream precompute rests on, with the engine's own slices"""
    from e2fgvi_amd.engine import split_prop_weights
    g = torch.Generator().manual_seed(31 + nparts)
    ch, h, w = 128, 10, 14
    w_off0 = torch.randn(128, 3 * ch + 4, 3, 3, generator=g, dtype=torch.float64) * 0.05
    w_bb0 = torch.randn(128, nparts * ch, 3, 3, generator=g, dtype=torch.float64) * 0.05
    bias = torch.randn(128, generator=g, dtype=torch.float64)
    ws = {k: v.double() for k, v in split_prop_weights(w_off0, w_bb0).items()}
    c1, cur, c2 = (torch.randn(1, ch, h, w, generator=g, dtype=torch.float64) for _ in range(3))
    fl = torch.randn(1, 4, h, w, generator=g, dtype=torch.float64)
    whole = F.conv2d(torch.cat([c1, cur, c2, fl], 1), w_off0, bias, padding=1)
    parts = F.conv2d(torch.cat([c1, c2, fl], 1), ws["off_rec"], bias, padding=1) + F.conv2d(cur, ws["off_cur"], None, padding=1)
    assert (whole - parts).abs().max() < 1e-12
    other = [torch.randn(1, ch, h, w, generator=g, dtype=torch.float64) for _ in range(nparts - 1)]   # cur[, the backward feature]
    prop = torch.randn(1, ch, h, w, generator=g, dtype=torch.float64)
    whole = F.conv2d(torch.cat(other + [prop], 1), w_bb0, bias, padding=1)
    parts = F.conv2d(prop, ws["bb_rec"], bias, padding=1) + F.conv2d(torch.cat(other, 1), ws["bb_pre"], None, padding=1)
    assert (whole - parts).abs().max() < 1e-12
