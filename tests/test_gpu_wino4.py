"""Wide-tile Winograd kernel F(2x4,3x3) x 64 couts (csrc/conv_wino4.hip; the F(4x4) and 32-cout shapes were pruned in round 6) against torch fp32 conv2d -- the operator the
reference runs at model/e2fgvi.py:77-93,112-150 and model/modules/feat_prop.py:20-28,73-79 -- through the C ABI
(e2fgvi_conv3x3_winograd4).  Same contract as the F(2x2,3x3) tests of test_gpu_ops.py: virtual concat, groups, bias,
activation, residual, strided destination, the DCN offset post-processing.  Tolerances: F(2x4) 2e-5 x rms (as F(2x2)),
F(4x4) 1e-4 x rms (measured fp32 rounding of the 6x6 transforms: 1.9e-5 x rms at 512 input channels)."""
import math
import pytest
import torch
import torch.nn.functional as F

from tests.util import assert_close, gen as _gen, nchw, nhwc
from tests.test_gpu_ops import _act_ref, conv64

pytestmark = pytest.mark.gpu

# floors; above them the bound grows with sqrt(input channels): the transforms' own fp32 roundings are amplified by the
# transform coefficients (F(2x4): up to 4, F(4x4): up to 8) before the channel sum.  Measured over 4 data sets
# (tools/gpu_suite_soak.sh): F(2x4) 1.1e-6 x sqrt(cin), F(4x4) 4.9e-6 x sqrt(cin) (x rms of the output); allowed: twice that.
TOL = {2464: 3e-5}
SLOPE = {2464: 2.4e-6}


def wtol(code, cin, floor_scale=1.0):
    return max(floor_scale * TOL[code], SLOPE[code] * math.sqrt(cin))

W4_CASES = [
    # N, H, W, cpg, groups, Cout, act, dst_ld, dst_coff
    (1, 16, 16, [8], 1, 32, 0, None, 0),                 # one block, one chunk
    (2, 28, 52, [64], 1, 64, 2, None, 0),                # partial blocks in both directions
    (2, 60, 108, [256], 1, 384, 2, None, 0),             # encoder.layers.8 shape (2 frames)
    (2, 32, 56, [128, 192], 2, 512, 2, None, 0),         # encoder.layers.10: grouped virtual concat
    (2, 32, 56, [64, 128], 4, 384, 2, None, 0),          # layers.12: Cout_g = 96
    (2, 32, 56, [32, 48], 8, 256, 2, None, 0),           # layers.14: Cout_g = 32, cpg 48 (6 chunks)
    (3, 24, 40, [256, 256], 1, 128, 2, None, 0),         # layers.16
    (1, 36, 52, [128], 1, 128, 1, 160, 16),              # write into a channel slice of a wider tensor
    (1, 20, 68, [40], 1, 24, 3, None, 0),                # Cout not a multiple of 32, tanh
    (5, 4, 4, [16], 1, 8, 0, None, 0),                   # image smaller than a block
    (1, 60, 108, [128, 128, 128], 1, 128, 2, None, 0),   # backbone.0 (forward) on one frame
    (2, 32, 56, [128, 128, 128, 4], 1, 128, 2, None, 0), # conv_offset.0: a 4-channel source (flows) ends a chunk
    (1, 20, 36, [12, 4, 8], 1, 40, 0, None, 0),          # sources of 12 / 4 / 8 channels
    (1, 240, 432, [64], 1, 64, 2, None, 0),              # decoder.4 at full resolution (one frame)
]


@pytest.mark.parametrize("code", [2464])
@pytest.mark.parametrize("case", W4_CASES, ids=lambda c: "x".join(str(v) for v in c[:7]))
def test_conv3x3_winograd4(dev, case, code):
    from e2fgvi_amd import ops
    N, H, W, cpg, groups, Cout, act, dst_ld, dst_coff = case
    g = _gen(51)
    srcs = [torch.randn(N, groups * c, H, W, generator=g) for c in cpg]
    cin_g = sum(cpg)
    w = torch.randn(Cout, cin_g, 3, 3, generator=g) / math.sqrt(cin_g * 9)
    b = torch.randn(Cout, generator=g)
    xcat = torch.cat([s.view(N, groups, c, H, W) for s, c in zip(srcs, cpg)], 2).view(N, groups * cin_g, H, W)
    ref = _act_ref(conv64(xcat, w, b, stride=1, padding=1, groups=groups), act, 0.2)
    layer = ops.PackedConv(w.to(dev), b.to(dev), cpg, groups=groups, stride=1, pad=1, algo="winograd")
    if dst_ld is None:
        out = layer([nhwc(s).to(dev) for s in srcs], act=act, slope=0.2, tile=code)
    else:
        full = torch.full((N, H, W, dst_ld), 7.0, device=dev)
        layer([nhwc(s).to(dev) for s in srcs], out=full, out_coff=dst_coff, act=act, slope=0.2, tile=code)
        out = full[..., dst_coff:dst_coff + Cout]
        rest = torch.cat([full[..., :dst_coff], full[..., dst_coff + Cout:]], 3)
        assert (rest == 7.0).all(), "winograd4 conv wrote outside its channel slice"
    assert_close(nchw(out.cpu()), ref, wtol(code, cin_g), "winograd4 conv %d %s" % (code, case))


@pytest.mark.parametrize("code", [2464])
def test_conv3x3_winograd4_residual(dev, code):
    """residual add in the epilogue (backbone.2 of the propagation: feat_prop + conv(...)), aligned and not"""
    from e2fgvi_amd import ops
    g = _gen(52)
    x = torch.randn(2, 128, 32, 56, generator=g)
    w = torch.randn(128, 128, 3, 3, generator=g) / math.sqrt(128 * 9)
    b = torch.randn(128, generator=g)
    layer = ops.PackedConv(w.to(dev), b.to(dev), [128], pad=1, algo="winograd")
    for res_ld, res_coff in ((128, 0), (136, 8), (131, 3)):
        resfull = torch.randn(2, 32, 56, res_ld, generator=g)
        res = resfull[..., res_coff:res_coff + 128]
        ref = F.leaky_relu(conv64(x, w, b, padding=1) + nchw(res), 0.1)
        out = layer([nhwc(x).to(dev)], residual=resfull.to(dev), res_coff=res_coff, act=2, slope=0.1, tile=code)
        assert_close(nchw(out.cpu()), ref, wtol(code, 128), "winograd4 %d conv + residual (ld %d coff %d)" % (code, res_ld, res_coff))


@pytest.mark.parametrize("code", [2464])
def test_conv3x3_winograd4_dcnpost(dev, code):
    """ACT_DCNPOST epilogue (10*tanh + flow.flip on the offsets, sigmoid on the masks; feat_prop.py:38-53) == the torch
    formula (the epilogue uses hardware exp2 / rcp: absolute error ~2e-6 on offsets of magnitude ~10)"""
    from e2fgvi_amd import ops
    g = _gen(53)
    N, H, W = 2, 32, 56
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
    out = wl([nhwc(x).to(dev)], residual=fl.to(dev), act=ops.ACT_DCNPOST, slope=10.0, tile=code)
    assert_close(nchw(out.cpu()), ref, max(5e-5, 2 * wtol(code, 128)), "winograd4 %d DCNPOST vs torch" % code)


def test_conv3x3_winograd4_argument_errors(dev):
    from e2fgvi_amd import ops
    from e2fgvi_amd.lib import HipError
    w = torch.randn(32, 16, 3, 3, device=dev)
    layer = ops.PackedConv(w, None, [16], pad=1, algo="winograd")
    with pytest.raises(HipError):
        layer([torch.randn(1, 16, 18, 16, device=dev)], tile=2464)       # W % 4
    out = layer([torch.randn(1, 18, 16, 16, device=dev)], tile=2464)     # H % 2 is enough for F(2x4)
    assert out.shape == (1, 18, 16, 32)
    # the shapes pruned in round 6 are refused by the library, not silently replaced
    from e2fgvi_amd import lib
    import ctypes as C
    d = lib.ConvDesc()
    d.tile = 32
    assert lib.load().e2fgvi_conv3x3_winograd4(C.byref(d), 4, None) != 0
