"""Kernel-level parity: every C-ABI kernel against plain torch fp32 CPU ops / the oracle pieces."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.util import assert_close, fp32_tol, gen as _gen, nchw, nhwc

pytestmark = pytest.mark.gpu

REL = 2e-5      # fp32 MFMA accumulation vs CPU fp32, relative to the rms of the reference output
# fused attention vs the oracle's fp32 chain (two fp32 implementations with different softmax / summation orders over
# 840 ... 4200 keys; measured 3.2-3.5e-5 x rms over 4 data sets): twice the measured maximum
ATT_TOL = 8e-5


def conv64(x, w, b=None, **kw):
    """the reference convolution accumulated in fp64 (so that the comparison sees only the kernel's own fp32 rounding)"""
    return F.conv2d(x.double(), w.double(), None if b is None else b.double(), **kw)


def _act_ref(x, act, slope):
    from e2fgvi_amd import ops
    if act == ops.ACT_RELU:
        return F.relu(x)
    if act == ops.ACT_LRELU:
        return F.leaky_relu(x, slope)
    if act == ops.ACT_TANH:
        return torch.tanh(x)
    return x


CONV_CASES = [
    # N, H, W, cpg list, groups, Cout, k, stride, pad, act, residual, tile, bk
    (2, 20, 28, [32], 1, 64, 3, 1, 1, 2, False, 0, None),
    (1, 33, 47, [64], 1, 128, 3, 1, 1, 0, True, 1, None),
    (1, 33, 47, [64], 1, 128, 3, 1, 1, 0, True, 1, 16),
    (1, 30, 54, [128], 1, 128, 3, 1, 1, 2, True, 3, None),
    (1, 30, 54, [128], 1, 128, 3, 1, 1, 2, False, 6, None),
    (2, 24, 40, [64], 1, 64, 3, 2, 1, 2, False, 2, None),
    (1, 30, 54, [128, 128, 128, 4], 1, 128, 3, 1, 1, 2, False, 0, None),     # conv_offset.0
    (1, 30, 54, [128, 128], 1, 432, 3, 1, 1, 0, False, 0, None),
    (3, 16, 24, [32, 48], 8, 256, 3, 1, 1, 2, False, 0, None),               # encoder g8
    (2, 16, 24, [64, 128], 4, 384, 3, 1, 1, 2, False, 0, None),              # encoder g4
    (2, 16, 24, [128, 192], 2, 512, 3, 1, 1, 2, False, 0, None),             # encoder g2
    (2, 24, 32, [4], 1, 64, 3, 2, 1, 2, False, 0, None),                     # encoder first conv (3->4 padded)
    (3, 16, 32, [8], 1, 32, 7, 1, 3, 1, False, 0, None),                     # spynet conv 1
    (3, 16, 32, [32], 1, 64, 7, 1, 3, 1, False, 0, None),
    (3, 16, 32, [64], 1, 32, 7, 1, 3, 1, False, 4, None),
    (3, 16, 32, [64], 1, 32, 7, 1, 3, 1, False, 5, None),
    (3, 16, 32, [32], 1, 16, 7, 1, 3, 1, False, 0, None),
    (3, 4, 8, [16], 1, 2, 7, 1, 3, 0, True, 0, None),                        # spynet last conv + residual
    (2, 30, 54, [128], 1, 512, 7, 3, 3, 0, False, 0, None),                  # soft split
    (2, 9, 13, [256], 1, 128, 1, 1, 0, 0, True, 0, None),                    # fusion 1x1
    (1, 40, 72, [64], 1, 3, 3, 1, 1, 3, False, 0, None),                     # decoder last conv (tanh)
    # halo-staged kernel (tile codes 10000 + id), every instantiated configuration
    (1, 33, 47, [64], 1, 128, 3, 1, 1, 0, True, 10001, None),
    (1, 33, 47, [64], 1, 128, 3, 1, 1, 2, True, 10002, None),
    (1, 30, 54, [128], 1, 128, 3, 1, 1, 2, False, 10003, None),
    (2, 30, 54, [128], 1, 200, 3, 1, 1, 2, False, 10004, None),
    (2, 24, 40, [64], 1, 64, 3, 1, 1, 2, False, 10005, None),
    (2, 24, 40, [64], 1, 64, 3, 1, 1, 2, True, 10011, None),
    (2, 17, 40, [32], 1, 32, 3, 1, 1, 1, False, 10006, None),
    (2, 17, 40, [32], 1, 16, 3, 1, 1, 1, False, 10012, None),
    (3, 16, 32, [64], 1, 32, 7, 1, 3, 1, False, 10007, None),
    (3, 16, 32, [32], 1, 16, 7, 1, 3, 1, False, 10008, None),
    (3, 16, 32, [32], 1, 64, 7, 1, 3, 1, False, 10009, None),
    (3, 20, 25, [64], 1, 64, 7, 1, 3, 1, True, 10010, None),
    (3, 16, 32, [8], 1, 32, 7, 1, 3, 1, False, 10007, 16),                   # spynet conv 1 (8 channels, padded block)
    (2, 16, 24, [128, 192], 2, 512, 3, 1, 1, 2, False, 10001, None),         # grouped, two sources
    (1, 30, 54, [128, 128, 128, 4], 1, 128, 3, 1, 1, 2, False, 10001, None),  # four sources incl. a 4-channel one
    (1, 40, 72, [64], 1, 3, 3, 1, 1, 3, False, 10006, None),                 # tanh + 3 output channels
    # row-staged weight slabs (one barrier per kernel row)
    (1, 33, 47, [64], 1, 128, 3, 1, 1, 2, True, 10022, None),
    (2, 30, 54, [128], 1, 200, 3, 1, 1, 2, False, 10024, None),
    (2, 24, 40, [64], 1, 64, 3, 1, 1, 2, True, 10031, None),
    (2, 24, 40, [64, 32], 1, 64, 3, 1, 1, 2, False, 10025, None),
    (2, 17, 40, [32], 1, 16, 3, 1, 1, 1, False, 10032, None),
    (1, 40, 72, [64], 1, 3, 3, 1, 1, 3, False, 10026, None),
    (3, 16, 32, [64], 1, 32, 7, 1, 3, 1, False, 10027, None),
    (3, 16, 32, [8], 1, 32, 7, 1, 3, 1, False, 10027, 16),
    (3, 20, 25, [64], 1, 64, 7, 1, 3, 1, True, 10030, None),
    (3, 16, 32, [32], 1, 16, 7, 1, 3, 1, False, 10028, None),
    # 16-output-channel tiles on v_mfma_f32_16x16x4_f32
    (3, 16, 32, [32], 1, 16, 7, 1, 3, 1, False, 10041, None),
    (3, 20, 33, [16], 1, 2, 7, 1, 3, 0, True, 10041, None),
    (2, 17, 40, [64], 1, 3, 3, 1, 1, 3, False, 10042, None),
    (2, 17, 40, [32], 1, 16, 3, 1, 1, 2, True, 10042, None),
    (3, 16, 32, [64], 1, 12, 7, 1, 3, 1, False, 10043, None),
    (2, 17, 40, [64], 1, 16, 3, 1, 1, 1, False, 10044, None),
    (2, 17, 40, [32], 1, 5, 3, 1, 1, 1, False, 10045, None),
    (3, 16, 32, [16], 1, 16, 7, 1, 3, 1, False, 10046, None),
    # 8-channel K-chunks (pack granule 8): implicit GEMM and halo
    (3, 16, 32, [8], 1, 32, 7, 1, 3, 1, False, 0, 8),
    (2, 24, 32, [4], 1, 64, 3, 2, 1, 2, False, 0, 8),
    (3, 4, 8, [8], 1, 32, 7, 1, 3, 1, False, 0, 8),
    (3, 16, 32, [8], 1, 32, 7, 1, 3, 1, False, 10051, 8),
    (2, 24, 32, [8], 1, 24, 3, 1, 1, 2, True, 10052, 8),
    (2, 24, 32, [4], 1, 64, 3, 1, 1, 2, False, 10053, 8),
    (2, 24, 32, [8], 1, 64, 7, 1, 3, 2, False, 10054, 8),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(str(v) for v in c[:9]))
def test_conv(dev, case):
    from e2fgvi_amd import ops
    N, H, W, cpg, groups, Cout, k, stride, pad, act, use_res, tile, bk = case
    g = _gen(1)
    srcs = [torch.randn(N, groups * c, H, W, generator=g) for c in cpg]
    cin_g = sum(cpg)
    w = torch.randn(Cout, cin_g, k, k, generator=g) / math.sqrt(cin_g * k * k)
    b = torch.randn(Cout, generator=g)
    # reference: per-group interleaved concat (reference e2fgvi.py:101-107 for the encoder)
    xcat = torch.cat([s.view(N, groups, c, H, W) for s, c in zip(srcs, cpg)], 2).view(N, groups * cin_g, H, W)
    ref = conv64(xcat, w, b, stride=stride, padding=pad, groups=groups)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if res is not None:
        ref = ref + res
    ref = _act_ref(ref, act, 0.2)

    layer = ops.PackedConv(w.to(dev), b.to(dev), cpg, groups=groups, stride=stride, pad=pad, bk=bk)
    out = layer([nhwc(s).to(dev) for s in srcs], residual=None if res is None else nhwc(res).to(dev), act=act,
                slope=0.2, tile=tile)
    assert_close(nchw(out.cpu()), ref, fp32_tol(cin_g * k * k), "conv %s" % (case,))


def test_conv_dcn_postprocess_epilogue(dev):
    """ACT_DCNPOST: conv_offset's last layer emits finished offsets / masks (feat_prop.py:38-53)"""
    from e2fgvi_amd import ops
    g = _gen(30)
    N, H, W = 2, 14, 22
    x = torch.randn(N, 128, H, W, generator=g)
    w = torch.randn(432, 128, 3, 3, generator=g) / 40
    b = torch.randn(432, generator=g) * 0.1
    f1 = torch.randn(N, 2, H, W, generator=g) * 2
    f2 = torch.randn(N, 2, H, W, generator=g) * 2
    raw = conv64(x, w, b, padding=1)
    o1, o2, m = torch.chunk(raw, 3, 1)
    off = 10 * torch.tanh(torch.cat((o1, o2), 1))
    q1, q2 = torch.chunk(off, 2, 1)
    ref = torch.cat([q1 + f1.flip(1).repeat(1, 72, 1, 1), q2 + f2.flip(1).repeat(1, 72, 1, 1), torch.sigmoid(m)], 1)
    layer = ops.PackedConv(w.to(dev), b.to(dev), [128], pad=1)
    out = layer([nhwc(x).to(dev)], residual=nhwc(torch.cat([f1, f2], 1)).to(dev), act=ops.ACT_DCNPOST, slope=10.0)
    assert_close(nchw(out.cpu()), ref, 3e-5, "dcn post-process epilogue")


def test_conv_slices_and_nchw_out(dev):
    """channel-offset sources, output into a slice of a wider buffer, NCHW store."""
    from e2fgvi_amd import ops
    g = _gen(2)
    N, H, W = 2, 12, 20
    wide = torch.randn(N, H, W, 96, generator=g)
    w = torch.randn(40, 32, 3, 3, generator=g) / 17
    b = torch.randn(40, generator=g)
    ref = conv64(nchw(wide)[:, 32:64], w, b, padding=1)
    layer = ops.PackedConv(w.to(dev), b.to(dev), [32], pad=1)
    dst = torch.zeros(N, H, W, 64, device=dev)
    layer([(wide.to(dev), 32)], out=dst, out_coff=16)
    assert_close(nchw(dst.cpu())[:, 16:56], ref, REL, "slice out")
    assert dst.cpu()[..., :16].abs().max() == 0 and dst.cpu()[..., 56:].abs().max() == 0
    o2 = layer([(wide.to(dev), 32)], out_nchw=True)
    assert_close(o2.cpu(), ref, REL, "nchw out")


@pytest.mark.parametrize("rows,cin,cout", [(450, 512, 1536), (77, 512, 512), (300, 1960, 512), (130, 512, 1960),
                                           (64, 512, 6272)])
def test_linear(dev, rows, cin, cout):
    from e2fgvi_amd import ops
    g = _gen(3)
    x = torch.randn(rows, cin, generator=g)
    w = torch.randn(cout, cin, generator=g) / math.sqrt(cin)
    b = torch.randn(cout, generator=g)
    r = torch.randn(rows, cout, generator=g)
    lin = ops.PackedLinear(w.to(dev), b.to(dev))
    out = lin(x.to(dev), residual=r.to(dev))
    assert_close(out.cpu(), F.linear(x.double(), w.double(), b.double()) + r, fp32_tol(cin), "linear %dx%d->%d" % (rows, cin, cout))


@pytest.mark.parametrize("tile", [1, 2, 3, 101, 102, 105])
@pytest.mark.parametrize("mag", [0.0, 3.0, 40.0])
def test_mdcn_generic(dev, tile, mag):
    """mmcv semantics with explicit offset/mask tensors (oracle/dcn.py), incl. far out-of-bounds offsets."""
    from e2fgvi_amd import ops
    from oracle.dcn import modulated_deform_conv2d
    g = _gen(4)
    N, C, H, W, Co, dg = 2, 64, 13, 19, 48, 4
    x = torch.randn(N, C, H, W, generator=g)
    off = torch.randn(N, dg * 18, H, W, generator=g) * mag
    msk = torch.rand(N, dg * 9, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / 24
    b = torch.randn(Co, generator=g)
    ref = modulated_deform_conv2d(x, off, msk, w, b, 1, 1, 1, 1, dg)
    layer = ops.PackedDcn(w.to(dev), b.to(dev), dg, pad=1)
    out = layer([nhwc(x).to(dev)], nhwc(off).to(dev), mask=nhwc(msk).to(dev), tile=tile)
    assert_close(nchw(out.cpu()), ref, 5e-5, "mdcn generic")


def test_mdcn_e2fgvi_fused(dev):
    """the fused form used by the propagation: two sources, raw conv_offset output + flows (feat_prop.py:38-58)."""
    from e2fgvi_amd import ops
    from oracle.dcn import modulated_deform_conv2d
    g = _gen(5)
    N, H, W, dg = 1, 30, 54, 16
    a = torch.randn(N, 128, H, W, generator=g)
    c = torch.randn(N, 128, H, W, generator=g)
    raw = torch.randn(N, 432, H, W, generator=g) * 0.5
    f1 = torch.randn(N, 2, H, W, generator=g) * 2
    f2 = torch.randn(N, 2, H, W, generator=g) * 2
    w = torch.randn(128, 256, 3, 3, generator=g) / 48
    b = torch.randn(128, generator=g)
    o1, o2, m = torch.chunk(raw, 3, 1)
    offset = 10 * torch.tanh(torch.cat((o1, o2), 1))
    q1, q2 = torch.chunk(offset, 2, 1)
    q1 = q1 + f1.flip(1).repeat(1, 72, 1, 1)
    q2 = q2 + f2.flip(1).repeat(1, 72, 1, 1)
    ref = modulated_deform_conv2d(torch.cat([a, c], 1), torch.cat([q1, q2], 1), torch.sigmoid(m), w, b, 1, 1, 1, 1, dg)
    layer = ops.PackedDcn(w.to(dev), b.to(dev), dg, pad=1)
    flows = nhwc(torch.cat([f1, f2], 1)).to(dev)
    for tile in (0, 1, 2, 4, 5, 6):          # incl. the K-group (split-K) tiles the 60x108 propagation uses
        out = layer([nhwc(a).to(dev), nhwc(c).to(dev)], nhwc(raw).to(dev), flows=flows, max_residue=10.0, tile=tile)
        assert_close(nchw(out.cpu()), ref, 5e-5, "mdcn fused tile %d" % tile)
    # the engine's form: offsets / masks already post-processed (ACT_DCNPOST epilogue of conv_offset.6), no flows
    final = torch.cat([q1, q2, torch.sigmoid(m)], 1)
    for tile in (0, 5):
        out = layer([nhwc(a).to(dev), nhwc(c).to(dev)], nhwc(final).to(dev), tile=tile)
        assert_close(nchw(out.cpu()), ref, 5e-5, "mdcn finished offsets tile %d" % tile)


def test_mdcn_argument_errors(dev):
    from e2fgvi_amd import ops
    from e2fgvi_amd.lib import HipError
    layer = ops.PackedDcn(torch.randn(32, 32, 3, 3, device=dev), None, 2, pad=1)
    x = torch.randn(1, 8, 8, 32, device=dev)
    with pytest.raises(HipError):                                      # odd offset row stride (dy, dx are one 8-byte load)
        layer([x], torch.randn(1, 8, 8, 55, device=dev))
    with pytest.raises(ValueError):
        layer([x], torch.randn(1, 8, 8, 20, device=dev))               # fused tensor narrower than dg * 3 * K
    with pytest.raises(ValueError):
        layer([torch.randn(1, 8, 8, 16, device=dev)], torch.randn(1, 8, 8, 54, device=dev))


@pytest.mark.parametrize("B,T,fh,fw", [(1, 3, 10, 18), (2, 2, 20, 36), (1, 5, 20, 36), (1, 2, 15, 45), (1, 40, 5, 9)])
def test_focal_attention(dev, B, T, fh, fw):
    """fused attention vs the oracle's roll/partition/cat/softmax chain (pre-projection output)."""
    from e2fgvi_amd import ops
    from e2fgvi_amd.engine import build_key_table
    from e2fgvi_amd.synth import rolled_valid_index
    from oracle import e2fgvi_oracle as O
    g = _gen(6)
    Cc = 512
    xn = torch.randn(B, T, fh, fw, Cc, generator=g)
    sd = {"a.qkv.weight": torch.randn(1536, Cc, generator=g) / math.sqrt(Cc) * 2.0,
          "a.qkv.bias": torch.randn(1536, generator=g) * 0.1,
          "pool_layers.0.weight": torch.full((1, 45), 1 / 45.) + 0.02 * torch.randn(1, 45, generator=g),
          "pool_layers.0.bias": torch.zeros(1)}
    xp = O.pool_windows(sd, "", xn)                                         # [B,nWh,nWw,T,C]
    pre = O.window_attention(sd, "a.", xn, xp, preproj=True)
    ref = O.window_reverse(pre, B, T, fh, fw).reshape(-1, Cc)
    qkv = F.linear(xn.reshape(-1, Cc), sd["a.qkv.weight"], sd["a.qkv.bias"])
    kvp = F.linear(xp.permute(0, 3, 1, 2, 4).reshape(-1, Cc), sd["a.qkv.weight"], sd["a.qkv.bias"])
    tab, nk = build_key_table(fh, fw, rolled_valid_index().tolist())
    outs = {}
    both = torch.cat([qkv, kvp], 0).to(dev)          # back to back: one buffer resource covers both (the LDS-DMA kernel needs it)
    q_d, p_d = both[:qkv.shape[0]], both[qkv.shape[0]:]
    # 0 = what the engine runs; 2 / 4 (+10: two key groups) = round 2's register-staged kernel; +20 / +30 = the LDS-DMA kernel
    for waves in (0, 2, 4, 12, 14, 22, 24, 32, 34):
        out = ops.focal_attention(q_d, p_d, torch.from_numpy(tab).to(dev), torch.from_numpy(nk).to(dev),
                                  B, T, fh, fw, waves=waves)
        assert_close(out.cpu(), ref, ATT_TOL, "attention waves=%d" % waves)
        outs[waves] = out
    # same arithmetic in the same key order (34 / 32 vs 14 / 12: two key groups each; on a small grid the launcher gives
    # 4 / 2 two key groups as well, 24 / 22 always one: another summation order): agreement to fp32 summation noise
    # summation noise grows with the square root of the key count (7 data sets: <= 0.85 of a 2e-5 base, hence 3e-5)
    agree = 3e-5 * max(1.0, (T * 210 / 840.0) ** 0.5)
    for a, b_ in ((24, 4), (22, 2), (34, 14), (32, 12)):
        assert_close(outs[a], outs[b_], agree, "LDS-DMA kernel (waves=%d) vs the register-staged one (waves=%d)" % (a, b_))


@pytest.mark.parametrize("B,T,fh,fw,far", [(1, 4, 60, 108, False), (1, 4, 90, 162, False), (1, 4, 60, 108, True)])
def test_focal_attention_large_grids(dev, B, T, fh, fw, far):
    """the BASELINE HQ token grids (720x1296 -> 60x108 = 12x12 windows, 1080x1944 -> 90x162 = 18x18): interior windows
    see a fully valid pooled neighbourhood (210 keys per frame, no analytic -100 mass), border windows wrap around the
    whole grid.  ``far``: qkv and kv_pool more than 4 GiB apart, so one buffer resource cannot cover both (the
    kernel's two-resource path, attention.hip ONE_RSRC = false).  Reference: tfocal_transformer_hq.py:231-425."""
    from e2fgvi_amd import ops
    from e2fgvi_amd.engine import build_key_table
    from e2fgvi_amd.synth import rolled_valid_index
    from oracle import e2fgvi_oracle as O
    g = _gen(60 + fh)
    Cc = 512
    xn = torch.randn(B, T, fh, fw, Cc, generator=g)
    sd = {"a.qkv.weight": torch.randn(1536, Cc, generator=g) / math.sqrt(Cc) * 2.0,
          "a.qkv.bias": torch.randn(1536, generator=g) * 0.1,
          "pool_layers.0.weight": torch.full((1, 45), 1 / 45.) + 0.02 * torch.randn(1, 45, generator=g),
          "pool_layers.0.bias": torch.zeros(1)}
    xp = O.pool_windows(sd, "", xn)
    pre = O.window_attention(sd, "a.", xn, xp, preproj=True)
    ref = O.window_reverse(pre, B, T, fh, fw).reshape(-1, Cc)
    qkv = F.linear(xn.reshape(-1, Cc), sd["a.qkv.weight"], sd["a.qkv.bias"])
    kvp = F.linear(xp.permute(0, 3, 1, 2, 4).reshape(-1, Cc), sd["a.qkv.weight"], sd["a.qkv.bias"])
    tab, nk = build_key_table(fh, fw, rolled_valid_index().tolist())
    assert nk.max() == 210 and nk.min() < 210
    if far:
        span = 5 * (1 << 30)
        arena = torch.empty(span + kvp.numel() * 4 + 256, dtype=torch.uint8, device=dev)
        q_d = arena[:qkv.numel() * 4].view(torch.float32).view(qkv.shape)
        p_d = arena[span:span + kvp.numel() * 4].view(torch.float32).view(kvp.shape)
        q_d.copy_(qkv); p_d.copy_(kvp)
        assert abs(q_d.data_ptr() - p_d.data_ptr()) >= (1 << 32)
    else:
        q_d, p_d = qkv.to(dev), kvp.to(dev)
    for waves in (0, 14) + (() if far else (24, 34)):
        out = ops.focal_attention(q_d, p_d, torch.from_numpy(tab).to(dev), torch.from_numpy(nk).to(dev), B, T, fh, fw,
                                  waves=waves)
        assert_close(out.cpu(), ref, ATT_TOL, "attention %dx%d far=%s waves=%d" % (fh, fw, far, waves))


def test_conv_batch_chunking_across_4gib(dev):
    """sources of a batch spanning >= 4 GiB are processed in image chunks (32-bit buffer resources inside the kernels,
    PackedConv.__call__): images on both sides of every chunk boundary must equal the single-image result."""
    from e2fgvi_amd import ops
    g = _gen(77)
    cin, cout, H, W, N = 512, 8, 512, 512, 9                   # 512 MiB per image, 4.5 GiB per batch
    w = torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)
    bias = torch.randn(cout, generator=g)
    layer = ops.PackedConv(w.to(dev), bias.to(dev), [cin])
    x = torch.empty(N, H, W, cin, device=dev)
    for n in range(N):
        x[n].normal_(generator=None)
    assert x.numel() * 4 >= (1 << 32)
    res = torch.randn(N, H, W, cout, device=dev)
    out = layer([x], residual=res, act=ops.ACT_LRELU, slope=0.1)
    step = ((1 << 32) - 2) // (H * W * cin * 4)
    for n in sorted({0, step - 1, step, N - 1}):
        one = layer([x[n:n + 1].contiguous()], residual=res[n:n + 1].contiguous(), act=ops.ACT_LRELU, slope=0.1)
        assert torch.equal(out[n:n + 1], one), "image %d differs from its single-image run" % n
    ref = F.leaky_relu(conv64(nchw(x[step].cpu()[None]), w, bias) + nchw(res[step].cpu()[None]), 0.1)
    assert_close(nchw(out[step:step + 1].cpu()), ref, REL, "chunked conv vs torch")
    del x, out


def test_mmcv_convmodule_standin(dev):
    """e2fgvi_amd.mmcv_ops.ConvModule (flow_comp.py:181-215's building block) == conv2d + ReLU, NCHW in / out"""
    from e2fgvi_amd import mmcv_ops
    torch.manual_seed(5)                      # module init + inputs below: a literal seed, the same tensors in every process
    m = mmcv_ops.ConvModule(8, 32, 7, 1, 3, norm_cfg=None, act_cfg=dict(type="ReLU")).to(dev)
    x = torch.randn(2, 8, 24, 40)
    ref = F.relu(conv64(x, m.conv.weight.detach().cpu(), m.conv.bias.detach().cpu(), padding=3))
    assert_close(m(x.to(dev)).cpu(), ref, fp32_tol(8 * 49), "ConvModule")
    m2 = mmcv_ops.ConvModule(16, 2, 7, 1, 3, norm_cfg=None, act_cfg=None).to(dev)
    x2 = torch.randn(1, 16, 16, 32)
    assert_close(m2(x2.to(dev)).cpu(), conv64(x2, m2.conv.weight.detach().cpu(), m2.conv.bias.detach().cpu(), padding=3), fp32_tol(16 * 49), "ConvModule no act")
    lin = torch.nn.Conv2d(4, 4, 3)
    mmcv_ops.constant_init(lin, 0.5, bias=0.25)
    assert float(lin.weight.min()) == 0.5 and float(lin.bias.max()) == 0.25
    assert mmcv_ops.load_checkpoint(lin, "https://example.invalid/spynet.pth") is None


def test_layout_roundtrip(dev):
    from e2fgvi_amd import ops
    g = _gen(7)
    x = torch.randn(3, 37, 21, 45, generator=g)
    y = ops.nchw_to_nhwc(x.to(dev), ld=40, scale=0.5, shift=0.25)
    ref = F.pad(nhwc(x * 0.5 + 0.25), (0, 3))
    assert_close(y.cpu(), ref, 1e-7, "nchw_to_nhwc")
    z = ops.nhwc_to_nchw(y, channels=37)
    assert_close(z.cpu(), x * 0.5 + 0.25, 1e-7, "nhwc_to_nchw")
    # frames: C <= 4 -> pixels of 4 channels (the fp32 path's input), C <= 8 -> pixels of 8 (the bf16 path's); one thread per pixel
    for ld, dt in ((4, torch.float32), (4, torch.bfloat16), (8, torch.float32)):
        for c in (3, 4, 1):
            fr = torch.randn(3, c, 23, 31, generator=g)
            y = ops.nchw_to_nhwc(fr.to(dev), ld=ld, scale=0.5, shift=0.5, out_dtype=dt)
            ref = F.pad(nhwc(fr * 0.5 + 0.5), (0, ld - c)).to(dt)
            assert torch.equal(y.cpu(), ref), "nchw_to_nhwc C=%d ld=%d %s" % (c, ld, dt)


@pytest.mark.parametrize("align,size_in,size_out", [(True, (240, 432), (60, 108)), (False, (60, 108), (64, 128)),
                                                    (False, (64, 128), (60, 108)), (True, (30, 54), (60, 108)),
                                                    (False, (30, 54), (32, 64))])
def test_resize(dev, align, size_in, size_out):
    from e2fgvi_amd import ops
    g = _gen(8)
    x = torch.rand(2, 3, *size_in, generator=g)
    sc = torch.tensor([2.0, 0.5, 1.5])
    sh = torch.tensor([0.1, -0.2, 0.3])
    ref = F.interpolate(x, size=size_out, mode="bilinear", align_corners=align) * sc.view(1, 3, 1, 1) + sh.view(1, 3, 1, 1)
    out = ops.resize_bilinear(x.to(dev), size_out, align, src_nchw=True, out_ld=4, scale=sc.to(dev), shift=sh.to(dev))
    assert_close(out.cpu()[..., :3], nhwc(ref), 2e-6, "resize nchw src")
    assert out.cpu()[..., 3].abs().max() == 0
    out2 = ops.resize_bilinear(nhwc(x).to(dev), size_out, align, scale=sc.to(dev), shift=sh.to(dev))
    assert_close(out2.cpu(), nhwc(ref), 2e-6, "resize nhwc src")


def test_resize_vec4_path(dev):
    """C % 4 == 0 NHWC sources take the 16-byte vectorised kernel (decoder x2 upsample, e2fgvi.py:126-129)"""
    from e2fgvi_amd import ops
    x = torch.randn(2, 64, 30, 54, generator=_gen(21))
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    out = ops.resize_bilinear(nhwc(x).to(dev), (60, 108), True)
    assert_close(out.cpu(), nhwc(ref), 2e-6, "resize vec4")


def test_avgpool(dev):
    from e2fgvi_amd import ops
    x = torch.randn(3, 4, 16, 24, generator=_gen(9))
    assert_close(ops.avgpool2(nhwc(x).to(dev)).cpu(), nhwc(F.avg_pool2d(x, 2, 2)), 1e-6, "avgpool")


def test_spynet_level_input(dev):
    from e2fgvi_amd import ops
    from oracle import e2fgvi_oracle as O
    g = _gen(10)
    Fr, h, w = 4, 16, 32
    pyr = torch.randn(Fr, 3, h, w, generator=g)
    ref_idx, supp_idx = [0, 1, 2, 1, 2, 3], [1, 2, 3, 0, 1, 2]
    flow_prev = torch.randn(6, 2, h // 2, w // 2, generator=g) * 3
    flow_up = F.interpolate(flow_prev, scale_factor=2, mode="bilinear", align_corners=True) * 2.0
    warped = O.flow_warp(pyr[supp_idx], flow_up.permute(0, 2, 3, 1), padding_mode="border")
    ref = torch.cat([pyr[ref_idx], warped, flow_up], 1)
    p4 = F.pad(nhwc(pyr), (0, 1)).to(dev)
    out = ops.spynet_level_input(p4, torch.tensor(ref_idx, dtype=torch.int32, device=dev),
                                 torch.tensor(supp_idx, dtype=torch.int32, device=dev), nhwc(flow_prev).to(dev))
    # grid_sample on the CPU side normalises / un-normalises the coordinates (~1e-6 px of jitter on |flow| ~ 6 px): measured 1.1e-5
    assert_close(out.cpu(), nhwc(ref), 4e-5, "spynet level input")
    out0 = ops.spynet_level_input(p4, torch.tensor(ref_idx, dtype=torch.int32, device=dev),
                                  torch.tensor(supp_idx, dtype=torch.int32, device=dev), None)
    ref0 = torch.cat([pyr[ref_idx], pyr[supp_idx], torch.zeros(6, 2, h, w)], 1)
    assert_close(out0.cpu(), nhwc(ref0), 1e-6, "spynet level 0 input")


def test_prop_cond(dev):
    from e2fgvi_amd import ops
    from oracle import e2fgvi_oracle as O
    g = _gen(11)
    b, lt, h, w, Cc = 2, 4, 15, 27, 128
    fp = torch.randn(b, Cc, h, w, generator=g)
    f2 = torch.randn(b, Cc, h, w, generator=g)
    flows = torch.randn(b, lt - 1, 2, h, w, generator=g) * 4
    i = 2
    f_n1 = flows[:, i - 1]
    c1 = O.flow_warp(fp, f_n1.permute(0, 2, 3, 1))
    f_n2 = f_n1 + O.flow_warp(flows[:, i - 2], f_n1.permute(0, 2, 3, 1))
    c2 = O.flow_warp(f2, f_n2.permute(0, 2, 3, 1))
    fl_nhwc = flows.permute(0, 1, 3, 4, 2).contiguous().to(dev)
    cond, fl = ops.prop_cond(nhwc(fp).to(dev), nhwc(f2).to(dev), fl_nhwc[0, i - 1], fl_nhwc[0, i - 2], (lt - 1) * h * w * 2)
    # grid_sample on the CPU side goes through normalise/unnormalise of the coordinates: ~1e-6 px of jitter
    assert_close(cond.cpu(), nhwc(torch.cat([c1, c2], 1)), 3e-4, "cond")
    assert_close(fl.cpu(), nhwc(torch.cat([f_n1, f_n2], 1)), 3e-4, "flows")
    cond1, fl1 = ops.prop_cond(nhwc(fp).to(dev), None, fl_nhwc[0, 0], None, (lt - 1) * h * w * 2)
    c1b = O.flow_warp(fp, flows[:, 0].permute(0, 2, 3, 1))
    assert_close(cond1.cpu(), nhwc(torch.cat([c1b, torch.zeros_like(c1b)], 1)), 3e-4, "cond i=1")
    assert_close(fl1.cpu(), nhwc(torch.cat([flows[:, 0], torch.zeros_like(flows[:, 0])], 1)), 1e-6, "flows i=1")


def test_layernorm_and_pool(dev):
    from e2fgvi_amd import ops
    from oracle import e2fgvi_oracle as O
    g = _gen(12)
    B, T, fh, fw, Cc = 2, 3, 10, 18, 512
    x = torch.randn(B, T, fh, fw, Cc, generator=g) * 3 + 1
    gm, bt = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    ref = F.layer_norm(x, (Cc,), gm, bt, 1e-5)
    out = ops.layernorm(x.view(-1, Cc).to(dev), gm.to(dev), bt.to(dev))
    assert_close(out.cpu(), ref.view(-1, Cc), 5e-6, "layernorm")
    sd = {"pool_layers.0.weight": torch.randn(1, 45, generator=g) * 0.1, "pool_layers.0.bias": torch.randn(1, generator=g)}
    refp = O.pool_windows(sd, "", ref).permute(0, 3, 1, 2, 4).reshape(-1, Cc)
    outp = ops.window_pool(out, sd["pool_layers.0.weight"].view(45).to(dev), sd["pool_layers.0.bias"].to(dev), B * T, fh, fw)
    assert_close(outp.cpu(), refp, 1e-5, "window_pool")


@pytest.mark.parametrize("H,W", [(60, 108), (30, 54)])
def test_fold_unfold(dev, H, W):
    """FFN fold/normalise/unfold/GELU and SoftComp fold vs F.fold / F.unfold (tfocal_transformer.py:92-97,70-71)."""
    from e2fgvi_amd import ops
    from e2fgvi_amd.engine import token_grid
    g = _gen(13)
    T2T = dict(kernel_size=(7, 7), stride=(3, 3), padding=(3, 3))
    fh, fw = token_grid(H, W)
    Fr, Cc = 2, 40
    hid = torch.randn(Fr, fh * fw, Cc * 49, generator=g)                      # reference channel order c*49+tap
    norm = F.fold(torch.ones(Fr, 49, fh * fw), output_size=(H, W), **T2T)
    folded_ref = F.fold(hid.permute(0, 2, 1), output_size=(H, W), **T2T) / norm
    unf_ref = F.gelu(F.unfold(folded_ref, **T2T).permute(0, 2, 1))              # [Fr, n, c*49+tap]
    hid_p = hid.view(Fr, fh * fw, Cc, 49).permute(0, 1, 3, 2).reshape(Fr * fh * fw, 49 * Cc).contiguous()
    folded = ops.ffn_fold(hid_p.to(dev), Fr, fh, fw, H, W, Cc)
    assert_close(folded.cpu(), nhwc(folded_ref), 1e-5, "ffn_fold")
    unf = ops.ffn_unfold_gelu(folded, fh, fw)
    unf_ref_p = unf_ref.reshape(Fr, fh * fw, Cc, 49).permute(0, 1, 3, 2).reshape(Fr * fh * fw, 49 * Cc)
    assert_close(unf.cpu(), unf_ref_p, 1e-5, "ffn_unfold_gelu")
    # the engine's form: GELU in front of the (pure-gather) unfold -- bit-identical in fp32
    unf2 = ops.ffn_unfold(ops.ffn_fold_gelu(hid_p.to(dev), Fr, fh, fw, H, W, Cc), fh, fw)
    assert torch.equal(unf2, unf), "GELU(fold) + unfold != fold + GELU(unfold)"
    C2 = 128
    emb = torch.randn(Fr, fh * fw, C2 * 49, generator=g)
    bias = torch.randn(C2, H, W, generator=g)
    res = torch.randn(Fr, C2, H, W, generator=g)
    sc_ref = F.fold(emb.permute(0, 2, 1), output_size=(H, W), **T2T) + bias[None] + res
    emb_p = emb.view(Fr, fh * fw, C2, 49).permute(0, 1, 3, 2).reshape(Fr * fh * fw, 49 * C2).contiguous()
    out = ops.softcomp_fold(emb_p.to(dev), Fr, fh, fw, H, W, C2, bias_hwc=bias.permute(1, 2, 0).contiguous().to(dev),
                            residual=nhwc(res).to(dev))
    assert_close(out.cpu(), nhwc(sc_ref), 1e-5, "softcomp_fold")


def test_bad_arguments_raise(dev):
    """error behaviour: wrong device / dtype / geometry raise instead of silently falling back."""
    from e2fgvi_amd import lib, ops
    w = torch.randn(16, 8, 3, 3, device=dev)
    layer = ops.PackedConv(w, None, [8], pad=1)
    with pytest.raises(TypeError):
        layer([torch.zeros(1, 4, 4, 8)])                      # CPU tensor
    with pytest.raises(TypeError):
        layer([torch.zeros(1, 4, 4, 8, device=dev, dtype=torch.float16)])
    with pytest.raises(ValueError):
        ops.PackedConv(w, None, [6], pad=1)                    # channel count mismatch
    with pytest.raises(lib.HipError):
        ops.PackedConv(torch.randn(16, 6, 3, 3, device=dev), None, [6], pad=1)   # cpg not a multiple of 4
    with pytest.raises(lib.HipError):
        ops.PackedDcn(torch.randn(8, 24, 3, 3, device=dev), None, 3)             # 8 channels per deform group


def test_mmcv_compatible_op(dev):
    """NCHW-in / NCHW-out op with mmcv's signature"""
    from e2fgvi_amd.mmcv_ops import ModulatedDeformConv2d
    from oracle.dcn import modulated_deform_conv2d as ref_op
    g = _gen(20)
    m = ModulatedDeformConv2d(32, 24, 3, padding=1, deform_groups=2).to(dev)
    x = torch.randn(2, 32, 10, 14, generator=g)
    off = torch.randn(2, 36, 10, 14, generator=g) * 2
    msk = torch.rand(2, 18, 10, 14, generator=g)
    out = m(x.to(dev), off.to(dev), msk.to(dev))
    ref = ref_op(x, off, msk, m.weight.detach().cpu(), m.bias.detach().cpu(), 1, 1, 1, 1, 2)
    assert_close(out.cpu(), ref, 5e-5, "mmcv-compatible op")


WINO_CASES = [
    # N, H, W, cpg, groups, Cout, act, tile, dst_ld, dst_coff
    (1, 16, 16, [8], 1, 32, 0, 0, None, 0),              # one block, one chunk
    (2, 30, 54, [64], 1, 64, 2, 0, None, 0),             # partial blocks in both directions
    (2, 60, 108, [256], 1, 384, 2, 0, None, 0),          # encoder.layers.8 shape (2 frames)
    (2, 30, 54, [128, 192], 2, 512, 2, 0, None, 0),      # encoder.layers.10: grouped virtual concat
    (2, 30, 54, [64, 128], 4, 384, 2, 0, None, 0),       # layers.12: Cout_g = 96 -> 32-wide tiles
    (2, 30, 54, [32, 48], 8, 256, 2, 0, None, 0),        # layers.14: Cout_g = 32, cpg 48 (6 chunks)
    (3, 22, 38, [256, 256], 1, 128, 2, 32, None, 0),     # layers.16 with forced 32-wide tiles
    (1, 34, 50, [128], 1, 128, 1, 64, 160, 16),          # write into a channel slice of a wider tensor
    (1, 18, 66, [40], 1, 24, 3, 0, None, 0),             # Cout not a multiple of 32, tanh
    (5, 2, 2, [16], 1, 8, 0, 0, None, 0),                # image smaller than a block
    (1, 60, 108, [128], 1, 128, 2, 132, None, 0),        # propagation conv, 8x16-pixel blocks
    (1, 60, 108, [128, 128, 128], 1, 128, 2, 164, None, 0),   # backbone.0 (forward), 8x16 blocks x 64 couts
    (2, 30, 54, [128, 128, 128, 4], 1, 128, 2, 0, None, 0),   # conv_offset.0: a 4-channel source (flows) ends a chunk
    (1, 20, 36, [12, 4, 8], 1, 40, 0, 132, None, 0),          # sources of 12 / 4 / 8 channels
]


@pytest.mark.parametrize("case", WINO_CASES, ids=lambda c: "x".join(str(v) for v in c[:8]))
def test_conv3x3_winograd(dev, case):
    """Winograd F(2x2,3x3) fp32 conv vs torch fp32 conv2d: same operator contract as conv2d_nhwc (virtual concat,
    groups, bias, activation, strided destination); tolerance 2e-5 x rms (fp32 Winograd rounding, ~4x plain fp32)"""
    from e2fgvi_amd import ops
    N, H, W, cpg, groups, Cout, act, tile, dst_ld, dst_coff = case
    g = _gen(41)
    srcs = [torch.randn(N, groups * c, H, W, generator=g) for c in cpg]
    cin_g = sum(cpg)
    w = torch.randn(Cout, cin_g, 3, 3, generator=g) / math.sqrt(cin_g * 9)
    b = torch.randn(Cout, generator=g)
    xcat = torch.cat([s.view(N, groups, c, H, W) for s, c in zip(srcs, cpg)], 2).view(N, groups * cin_g, H, W)
    ref = _act_ref(conv64(xcat, w, b, stride=1, padding=1, groups=groups), act, 0.2)
    layer = ops.PackedConv(w.to(dev), b.to(dev), cpg, groups=groups, stride=1, pad=1, algo="winograd")
    if dst_ld is None:
        out = layer([nhwc(s).to(dev) for s in srcs], act=act, slope=0.2, tile=tile)
    else:
        full = torch.full((N, H, W, dst_ld), 7.0, device=dev)
        layer([nhwc(s).to(dev) for s in srcs], out=full, out_coff=dst_coff, act=act, slope=0.2, tile=tile)
        out = full[..., dst_coff:dst_coff + Cout]
        rest = torch.cat([full[..., :dst_coff], full[..., dst_coff + Cout:]], 3)
        assert (rest == 7.0).all(), "winograd conv wrote outside its channel slice"
    tol = fp32_tol(cin_g * 9, floor=3e-5)
    assert_close(nchw(out.cpu()), ref, tol, "winograd conv %s" % (case,))
    # and against the implicit-GEMM path of the library (both are fp32)
    direct = ops.PackedConv(w.to(dev), b.to(dev), cpg, groups=groups, stride=1, pad=1)([nhwc(s).to(dev) for s in srcs],
                                                                                       act=act, slope=0.2)
    assert_close(out.cpu(), direct.cpu(), 1.5 * tol, "winograd vs implicit GEMM %s" % (case,))
    # every block shape of the fp32 kernel agrees with the tile the case names (round 6 removed the LDS-DMA staged copies, + 1000)
    for base in (32, 64, 132, 164):
        other = layer([nhwc(s).to(dev) for s in srcs], act=act, slope=0.2, tile=base)
        assert_close(nchw(other.cpu()), ref, tol, "winograd conv, block shape %d %s" % (base, case))


@pytest.mark.parametrize("tile", [0, 64, 132])
def test_conv3x3_winograd_residual(dev, tile):
    """residual add in the Winograd epilogue (backbone.2 of the propagation: feat_prop + conv(...)), aligned and not"""
    from e2fgvi_amd import ops
    g = _gen(42)
    x = torch.randn(2, 128, 30, 54, generator=g)
    w = torch.randn(128, 128, 3, 3, generator=g) / math.sqrt(128 * 9)
    b = torch.randn(128, generator=g)
    layer = ops.PackedConv(w.to(dev), b.to(dev), [128], pad=1, algo="winograd")
    for res_ld, res_coff in ((128, 0), (136, 8), (131, 3)):
        resfull = torch.randn(2, 30, 54, res_ld, generator=g)
        res = resfull[..., res_coff:res_coff + 128]
        ref = F.leaky_relu(conv64(x, w, b, padding=1) + nchw(res), 0.1)
        out = layer([nhwc(x).to(dev)], residual=resfull.to(dev), res_coff=res_coff, act=2, slope=0.1, tile=tile)
        assert_close(nchw(out.cpu()), ref, 3e-5, "winograd conv + residual (ld %d coff %d) tile %d" % (res_ld, res_coff, tile))


@pytest.mark.parametrize("tile", [0, 32, 164])
def test_conv3x3_winograd_dcnpost(dev, tile):
    """ACT_DCNPOST epilogue (10*tanh + flow.flip on the offsets, sigmoid on the masks; feat_prop.py:38-53) on the
    Winograd path == the implicit-GEMM path's, and == the torch formula"""
    from e2fgvi_amd import ops
    g = _gen(43)
    N, H, W = 2, 30, 54
    x = torch.randn(N, 128, H, W, generator=g)
    w = torch.randn(432, 128, 3, 3, generator=g) / math.sqrt(128 * 9)
    b = torch.randn(432, generator=g) * 0.1
    fl = torch.randn(N, H, W, 4, generator=g) * 3
    raw = conv64(x, w, b, padding=1)
    o1, o2, m = torch.chunk(raw, 3, 1)
    off = 10 * torch.tanh(torch.cat([o1, o2], 1))
    f1 = fl[..., 0:2].permute(0, 3, 1, 2)
    f2 = fl[..., 2:4].permute(0, 3, 1, 2)
    off1, off2 = torch.chunk(off, 2, 1)
    ref = torch.cat([off1 + f1.flip(1).repeat(1, 72, 1, 1), off2 + f2.flip(1).repeat(1, 72, 1, 1), torch.sigmoid(m)], 1)
    wl = ops.PackedConv(w.to(dev), b.to(dev), [128], pad=1, algo="winograd")
    out = wl([nhwc(x).to(dev)], residual=fl.to(dev), act=ops.ACT_DCNPOST, slope=10.0, tile=tile)
    assert_close(nchw(out.cpu()), ref, 5e-5, "winograd DCNPOST vs torch tile %d" % tile)
    dl = ops.PackedConv(w.to(dev), b.to(dev), [128], pad=1)
    direct = dl([nhwc(x).to(dev)], residual=fl.to(dev), act=ops.ACT_DCNPOST, slope=10.0)
    assert_close(out.cpu(), direct.cpu(), 6e-5, "winograd DCNPOST vs implicit GEMM tile %d" % tile)


def test_conv3x3_winograd_argument_errors(dev):
    from e2fgvi_amd import ops
    from e2fgvi_amd.lib import HipError
    w = torch.randn(32, 16, 3, 3, device=dev)
    with pytest.raises(ValueError):
        ops.PackedConv(torch.randn(32, 10, 3, 3, device=dev), None, [10], pad=1, algo="winograd")     # cpg % 4
    with pytest.raises(ValueError):
        ops.PackedConv(w, None, [16], stride=2, pad=1, algo="winograd")
    layer = ops.PackedConv(w, None, [16], pad=1, algo="winograd")
    with pytest.raises(HipError):
        layer([torch.randn(1, 15, 16, 16, device=dev)])                                               # odd H
    with pytest.raises(HipError):
        layer([torch.randn(1, 16, 16, 16, device=dev)], tile=48)                                      # unknown tile code


def test_fusion_layer_in_place_over_its_residual(dev):
    """engine.propagate runs the 1x1 fusion layer with out == residual (feat_prop.py:139-146: stack(fusion(cat(bwd, fwd))) + x, the
    sum written over x).  That is only sound if every kernel the layer can be handed to reads a residual element in the thread that
    writes it, before it writes it (advisor, round 5): every candidate -- the implicit GEMM's tiles, the LDS-DMA fp32 kernel's,
    the split-operand GEMM's -- is run aliased and compared bit for bit with its own out-of-place result."""
    from e2fgvi_amd import ops
    g = _gen(4242)
    w = torch.randn(128, 256, 1, 1, generator=g) / 16
    b = torch.randn(128, generator=g) * 0.1
    s0 = torch.randn(10, 60, 108, 128, generator=g).to(dev)
    s1 = torch.randn(10, 60, 108, 128, generator=g).to(dev)
    res = torch.randn(10, 60, 108, 128, generator=g).to(dev)
    layers = [("igemm", ops.PackedConv(w.to(dev), b.to(dev), [128, 128]), ops.TUNE_CANDIDATES),
              ("f32x", ops.PackedConvX(w.to(dev), b.to(dev), [128, 128], dtype=torch.float32), ops.XTUNE_CANDIDATES),
              ("f32x3", ops.PackedConvX(w.to(dev), b.to(dev), [128, 128], dtype=torch.float32, x3=True), ops.XTUNE_CANDIDATES)]
    ran = 0
    for name, layer, tiles in layers:
        for tile in tiles:
            try:
                ref = layer([s0, s1], residual=res, tile=tile)
            except Exception:                      # a tile this shape does not admit
                continue
            io = res.clone()
            layer([s0, s1], residual=io, out=io, tile=tile)
            torch.cuda.synchronize()
            assert torch.equal(io, ref), (name, tile, float((io - ref).abs().max()))
            ran += 1
    assert ran >= 12, ran
    # ... and what the layer of the engine itself picks (table decision) at one clip
    auto = ops.PackedConv(w.to(dev), b.to(dev), [128, 128])
    auto.try_x3 = True
    auto.tune = True
    ref = auto([s0, s1], residual=res)
    io = res.clone()
    auto([s0, s1], residual=io, out=io)
    assert torch.equal(io, ref)
