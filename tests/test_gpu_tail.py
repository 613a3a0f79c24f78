"""The decoder's last layer on csrc/conv_tail.hip (taps on the N side of one GEMM + shifted sum) against torch conv2d --
the operator the reference runs at model/e2fgvi.py:99-103,261 (nn.Conv2d(64, 3, 3, padding=1) + torch.tanh) -- through the
C ABI (e2fgvi_conv3x3_tail).  fp32 sources: exact fp32 MFMA, 2e-5 x rms (the bound of the other fp32 conv kernels).  bf16
sources: the reference is computed on the SAME bf16-rounded inputs and weights in fp64, so only the fp32 accumulation
order differs: 2e-5 x rms as well."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import assert_close, gen, nhwc

pytestmark = pytest.mark.gpu

CASES = [
    # N, H, W, src_ld, act, bias
    (1, 16, 32, 64, 0, True),        # exactly one tile
    (2, 17, 45, 64, 3, True),        # ragged in both directions, tanh
    (1, 3, 5, 64, 3, True),          # image smaller than a tile
    (3, 40, 72, 96, 3, True),        # source rows wider than the 64 channels read
    (1, 33, 64, 64, 2, False),       # LeakyReLU, no bias
    (1, 240, 432, 64, 3, True),      # decoder.6 at the 432x240 benchmark size (one frame)
    (1, 720, 1296, 64, 3, True),     # ... and at 720p
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv3x3_tail(dev, case, dtype):
    from e2fgvi_amd import ops
    N, H, W, ld, act, use_bias = case
    g = gen(H * 1000 + W)
    x = torch.randn(N, 64, H, W, generator=g)
    w = torch.randn(3, 64, 3, 3, generator=g) * (1.0 / 24.0)
    b = torch.randn(3, generator=g) * 0.1 if use_bias else None
    wide = torch.randn(N, H, W, ld, generator=g).to(dev)            # the channels past 64 must not be read
    wide[..., :64] = nhwc(x).to(dev)
    src = wide.to(dtype)
    conv = ops.PackedTailConv(w.to(dev), None if b is None else b.to(dev), dtype=dtype)
    got = conv([src], act=act, slope=0.2)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
        w = w.bfloat16().float()
    ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=1)
    if act == 3:
        ref = torch.tanh(ref)
    elif act == 2:
        ref = F.leaky_relu(ref, 0.2)
    assert got.dtype == torch.float32 and tuple(got.shape) == (N, 3, H, W)
    assert_close(got, ref, 2e-5, "conv3x3_tail %s %s" % (case, dtype))


def test_conv3x3_tail_matches_the_implicit_gemm(dev):
    """same layer through the general fp32 kernel (conv.hip) and the tail kernel"""
    from e2fgvi_amd import ops
    g = gen(5)
    x = nhwc(torch.randn(2, 64, 60, 108, generator=g)).to(dev)
    w = (torch.randn(3, 64, 3, 3, generator=g) / 24.0).to(dev)
    b = (torch.randn(3, generator=g) * 0.1).to(dev)
    a = ops.PackedConv(w, b, [64], pad=1)([x], act=ops.ACT_TANH, out_nchw=True)
    t = ops.PackedTailConv(w, b)([x], act=ops.ACT_TANH)
    assert_close(t, a, 3e-5, "tail vs implicit GEMM")


def test_conv3x3_tail_rejects_other_shapes(dev):
    from e2fgvi_amd import ops
    with pytest.raises(Exception):
        ops.PackedTailConv(torch.zeros(4, 64, 3, 3, device=dev), None)
    with pytest.raises(Exception):
        ops.PackedTailConv(torch.zeros(3, 32, 3, 3, device=dev), None)
