"""fp32 layers on the bf16 matrix pipe ("x3", csrc/conv_bf16x.hip MODE 2): fp32 operands split exactly into three bf16
pieces, six bf16 MFMA terms per product, fp32 accumulation.  The claim under test is that this is an fp32 kernel: against
fp64 references it must meet the SAME bounds as the exact-fp32 MFMA kernels (tests/util.fp32_tol: 8 sqrt(K) 2^-24 of the
output rms) on the same cases, its weight planes must sum to the fp32 weights bit for bit, and reruns must be bit-identical.
Replaces the reference's nn.Conv2d / nn.Linear calls of model/e2fgvi_hq.py and model/modules/tfocal_transformer_hq.py."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_bf16x import CASES, F32_TAP_CASES, conv64
from tests.util import assert_close, fp32_tol, gen as _gen, name_seed, nchw, nhwc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_x3(dev, case):
    """every geometry of the LDS-DMA kernel's own case list (groups, virtual concat, strides, channel tails, narrow N, 1-pixel
    images), every tile shape: fp32 tolerance against fp64"""
    from e2fgvi_amd import ops
    name, N, H, W, cpg, groups, Cout, k, stride, pad, tiles = case
    g = _gen(name_seed(name, 23))
    w = torch.randn(Cout, sum(cpg), k, k, generator=g) / math.sqrt(sum(cpg) * k * k)
    bias = torch.randn(Cout, generator=g) * 0.1
    srcs, parts = [], []
    for c in cpg:
        t = torch.randn(N, H, W, c * groups + 8, generator=g)
        srcs.append(t)
        parts.append(t[..., 4:4 + c * groups])
    x = torch.cat([torch.cat([p_[..., gi * c:(gi + 1) * c] for p_, c in zip(parts, cpg)], -1) for gi in range(groups)], -1)
    ref0 = conv64(nchw(x), w, bias, stride=stride, padding=pad, groups=groups)
    layer = ops.PackedConvX(w.to(dev), bias.to(dev), cpg, groups=groups, stride=stride, pad=pad, dtype=torch.float32, x3=True)
    src_d = [(s.to(dev), 4) for s in srcs]
    res = torch.randn(N, ref0.shape[2], ref0.shape[3], Cout, generator=g)
    ref = F.leaky_relu(ref0 + nchw(res), 0.1)
    outs = {}
    for tile in sorted(set(t for t in tiles if t < 10) | {1, 5, 7, 8, 107, 108}):
        out = layer(src_d, residual=res.to(dev), act=ops.ACT_LRELU, slope=0.1, tile=tile)
        assert out.dtype == torch.float32
        assert_close(nchw(out.cpu()), ref, fp32_tol(sum(cpg) * k * k), "%s x3 tile %d" % (name, tile))
        outs[tile] = out.clone()
    # the ping-pong forms (round 6: the two waves of a SIMD one interval apart) issue the products of tiles 7 / 8 in the same order
    assert torch.equal(outs[107], outs[7]) and torch.equal(outs[108], outs[8]), "%s: tiles 107 / 108 are not bit-identical to 7 / 8" % name
    with pytest.raises(Exception):
        layer(src_d, tile=11)               # the row-shift tiles are bf16-only


@pytest.mark.parametrize("case", F32_TAP_CASES, ids=[c[0] for c in F32_TAP_CASES])
def test_conv_x3_tap_packed(dev, case):
    from e2fgvi_amd import ops
    name, N, H, W, cin, Cout, k, stride, pad, tiles = case
    g = _gen(name_seed(name, 29))
    w = torch.randn(Cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    bias = torch.randn(Cout, generator=g) * 0.1
    wide = torch.randn(N, H, W, cin + 8, generator=g)
    ref0 = conv64(nchw(wide[..., 4:4 + cin]), w, bias, stride=stride, padding=pad)
    layer = ops.PackedConvX(w.to(dev), bias.to(dev), [cin], stride=stride, pad=pad, dtype=torch.float32, taps=True, x3=True)
    plain = ops.PackedConvX(w.to(dev), bias.to(dev), [cin], stride=stride, pad=pad, dtype=torch.float32, x3=True)
    assert layer.taps and not plain.taps and layer.wpacked.numel() < plain.wpacked.numel()
    res = torch.randn(N, ref0.shape[2], ref0.shape[3], Cout, generator=g)
    src = [(wide.to(dev), 4)]
    for tile in tiles:
        out = layer(src, residual=res.to(dev), tile=tile)
        assert_close(nchw(out.cpu()), ref0 + nchw(res), fp32_tol(cin * k * k), "%s x3 taps tile %d" % (name, tile))


def test_x3_weight_planes_sum_to_the_weight(dev):
    """the packing splits without loss: hi + mid + lo == w in fp32, for ordinary, tiny, huge and exactly-bf16 weights"""
    from e2fgvi_amd import ops
    g = _gen(41)
    Cout, cin = 96, 64
    w = torch.randn(Cout, cin, 1, 1, generator=g)
    w[0] *= 1e-25            # (below 2^-110 the last piece would be an fp32 denormal: tests/test_identities.py)
    w[1] *= 1e30
    w[2] = w[2].bfloat16().float()
    w[3] = 0.0
    w[4, :8, 0, 0] = torch.tensor([1.0, -1.0, 2.0 ** -126, -(2.0 ** -120), 3.0, 1.0 + 2.0 ** -23, -(1.0 + 2.0 ** -16), 65504.0])
    layer = ops.PackedConvX(w.to(dev), None, [cin], dtype=torch.float32, x3=True)
    torch.cuda.synchronize()
    Npad = 96
    p = layer.wpacked.view(cin // 32, 3, 4, Npad, 8).float().cpu()     # [K-step][plane][k-octet][n][8]
    rebuilt = (p[:, 0].double() + p[:, 1].double() + p[:, 2].double())     # exact in fp64
    rebuilt = rebuilt.permute(2, 0, 1, 3).reshape(Npad, cin)           # [n][step, octet, 8] -> [n][channel]
    assert torch.equal(rebuilt.float(), w.view(Cout, cin)), "the three planes do not sum to the weight"
    assert torch.equal(rebuilt, w.view(Cout, cin).double())


def test_x3_matches_the_exact_fp32_kernel_on_exactly_representable_data(dev):
    """operands with <= 8 significant bits have mid = lo = 0: the kernel must then reproduce an exact product sum (every term
    fits fp32 for K = 256): bit-equal to the fp64 result rounded once"""
    from e2fgvi_amd import ops
    g = _gen(43)
    x = torch.randint(-8, 9, (3000, 256), generator=g).float()
    w = torch.randint(-8, 9, (128, 256), generator=g).float()
    lin = ops.PackedLinearX(w.to(dev), None, dtype=torch.float32)
    lin3 = ops.PackedConvX(w.to(dev), None, [256], dtype=torch.float32, x3=True)
    ref = (x.double() @ w.double().t()).float()
    out3 = torch.empty(3000, 1, 1, 128, device=dev)
    lin3([x.to(dev).view(3000, 1, 1, 256)], out=out3)
    assert torch.equal(out3.view(3000, 128).cpu(), ref)
    assert torch.equal(lin(x.to(dev)).cpu(), ref)


def test_x3_error_is_not_worse_than_the_fp32_mfma_kernel(dev):
    """same call on the exact-fp32 MFMA kernel and on x3, both against fp64: the split kernel's error stays within 1.5x of the
    fp32 kernel's (measured: equal or smaller -- the six terms are exact products, only the accumulation rounds)"""
    from e2fgvi_amd import ops
    from tests.util import err
    g = _gen(47)
    for rows, cin, cout in ((7200, 512, 1960), (7200, 1960, 512), (4000, 2880, 256)):
        x = torch.randn(rows, cin, generator=g)
        w = torch.randn(cout, cin, generator=g) / math.sqrt(cin)
        b = torch.randn(cout, generator=g) * 0.1
        ref = F.linear(x.double(), w.double(), b.double())
        a = ops.PackedLinearX(w.to(dev), b.to(dev), dtype=torch.float32)(x.to(dev)).cpu()
        l3 = ops.PackedConvX(w.to(dev), b.to(dev), [cin], dtype=torch.float32, x3=True)
        o3 = torch.empty(rows, 1, 1, cout, device=dev)
        l3([x.to(dev).view(rows, 1, 1, cin)], out=o3)
        e32, e3 = err(a, ref)[1], err(o3.view(rows, cout).cpu(), ref)[1]
        print("linear %dx%d->%d: fp32 MFMA kernel %.2e, x3 %.2e of the output rms" % (rows, cin, cout, e32, e3))
        assert e3 <= fp32_tol(cin) and e3 <= 1.5 * e32 + 1e-6


def test_x3_dcn_postprocess_and_nchw(dev):
    """the epilogue variants the fp32 engine uses: ACT_DCNPOST (conv_offset.6) and the fp32 NCHW store"""
    from e2fgvi_amd import ops
    g = _gen(53)
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
    layer = ops.PackedConvX(w.to(dev), b.to(dev), [128], pad=1, dtype=torch.float32, x3=True)
    out = layer([nhwc(x).to(dev)], residual=nhwc(torch.cat([f1, f2], 1)).to(dev), act=ops.ACT_DCNPOST, slope=10.0)
    assert_close(nchw(out.cpu()), ref, 3e-5, "x3 dcn post-process epilogue")
    w3 = torch.randn(24, 128, 3, 3, generator=g) / 34
    l3 = ops.PackedConvX(w3.to(dev), None, [128], pad=1, dtype=torch.float32, x3=True)
    out = l3([nhwc(x).to(dev)], out_nchw=True)
    assert_close(out.cpu(), conv64(x, w3, padding=1), fp32_tol(1152), "x3 NCHW store")


def test_x3_reruns_are_bit_identical(dev):
    from e2fgvi_amd import ops
    g = _gen(59)
    x = torch.randn(7360, 512, generator=g).to(dev).view(7360, 1, 1, 512)
    w = (torch.randn(1536, 512, generator=g) / 22).to(dev)
    l3 = ops.PackedConvX(w, None, [512], dtype=torch.float32, x3=True)
    first = l3([x], tile=7).clone()
    bad = 0
    for _ in range(100):
        bad += int(not torch.equal(l3([x], tile=7), first))
    assert bad == 0, "%d of 100 launches differ" % bad
    # the ping-pong forms (asm LDS-DMA, raw barriers, their own waits): 200 launches each, the bits of tile 7 / tile 8's first launch
    first8 = l3([x], tile=8).clone()
    for tile, want in ((107, first), (108, first8)):
        bad = sum(int(not torch.equal(l3([x], tile=tile), want)) for _ in range(200))
        assert bad == 0, "tile %d: %d of 200 launches differ from the one-barrier tile" % (tile, bad)


def test_fp32_layers_may_pick_the_split_kernel(dev):
    """PackedConv / PackedLinear with try_x3: the decision (code ops.X3_BASE + tile when x3 wins) is recorded and whatever is chosen
    meets the fp32 bound; E2FGVI_X3=0 (ops.X3_ENABLED False) never builds the alternative"""
    from e2fgvi_amd import ops
    g = _gen(61)
    w = torch.randn(1960, 512, generator=g) / math.sqrt(512)
    b = torch.randn(1960, generator=g) * 0.1
    x = torch.randn(7200, 512, generator=g)
    ref = F.linear(x.double(), w.double(), b.double())
    lin = ops.PackedLinear(w.to(dev), b.to(dev))
    lin.tune = lin.try_x3 = True
    from tests.util import timed_tuning
    with timed_tuning():
        for _ in range(2):                        # first call tunes, second replays the decision
            assert_close(lin(x.to(dev)).cpu(), ref, fp32_tol(512), "tuned linear with the x3 alternative")
    key = [k for k in ops._TUNED if k[0] == 1960 and k[1] == (512,) and k[-1] == "x3"]
    assert key, "no tuning decision recorded"
    print("fc1-shaped fp32 linear: tile code", ops._TUNED[key[0]])
    # a Winograd layer: the static rule against x3
    wc = torch.randn(256, 128, 3, 3, generator=g) / 34
    xc = torch.randn(10, 60, 108, 128, generator=g)
    conv = ops.PackedConv(wc.to(dev), None, [128], pad=1, algo="auto")
    conv.try_x3 = True
    refc = conv64(nchw(xc), wc, padding=1)
    with timed_tuning():
        for _ in range(2):
            assert_close(nchw(conv([xc.to(dev)]).cpu()), refc, 3e-5, "winograd layer with the x3 alternative")
    saved = ops.X3_ENABLED
    try:
        ops.X3_ENABLED = False
        lin2 = ops.PackedLinear(w.to(dev), b.to(dev))
        lin2.tune = lin2.try_x3 = True
        with timed_tuning():
            lin2(x.to(dev))
        assert lin2.alt3 is None
    finally:
        ops.X3_ENABLED = saved


# ---------------------------------------------------------------------------------------------- Winograd with split operands
from tests.test_gpu_ops import WINO_CASES, _act_ref     # noqa: E402


def _pattern_on_mismatch(out, fp32, shape, case):
    """diagnostics for an intermittent wrong block: where (image, 16x16-pixel block, 32-cout tile, row / column parity) a split-
    operand Winograd output leaves the fp32 Winograd kernel's -- printed before the assertion fails"""
    d = (out - fp32).abs()
    if d.max().item() <= 1e-2 * fp32.abs().max().item():
        return
    bad = d > 1e-2 * fp32.abs().max().item()
    N, H, W, C = out.shape
    print("MISMATCH shape %d case %s: max err %.3f, bad fraction %.4f" % (shape, case[:6], d.max().item(), bad.float().mean().item()))
    print("  by image:", [round(bad[n].float().mean().item(), 4) for n in range(N)])
    print("  by 32-cout tile:", [round(bad[..., c:c + 32].float().mean().item(), 4) for c in range(0, C, 32)])
    print("  row parity:", [round(bad[:, p::2].float().mean().item(), 4) for p in (0, 1)], "col parity:",
          [round(bad[:, :, p::2].float().mean().item(), 4) for p in (0, 1)])
    for n in range(N):
        print("  img %d by 16x16 block:" % n, [[round(bad[n, by * 16:(by + 1) * 16, bx * 16:(bx + 1) * 16].float().mean().item(), 3)
                                                  for bx in range((W + 15) // 16)] for by in range((H + 15) // 16)])


@pytest.mark.parametrize("case", WINO_CASES, ids=lambda c: "x".join(str(v) for v in c[:8]))
def test_conv3x3_winograd_x3(dev, case):
    """the fp32 Winograd kernel's own case list on the split-bf16 build (csrc/conv_wino.hip X3; tile codes ops.W3_BASE + block
    shape): same bound against fp64 as the fp32 kernel (fp32 Winograd rounding), every block shape, and within the fp32
    kernel's own error of the fp32 kernel"""
    from e2fgvi_amd import ops
    N, H, W, cpg, groups, Cout, act, tile, dst_ld, dst_coff = case
    g = _gen(name_seed("wino x3 %s" % (case,), 3))
    srcs = [torch.randn(N, groups * c, H, W, generator=g) for c in cpg]
    cin_g = sum(cpg)
    w = torch.randn(Cout, cin_g, 3, 3, generator=g) / math.sqrt(cin_g * 9)
    b = torch.randn(Cout, generator=g)
    xcat = torch.cat([s_.view(N, groups, c, H, W) for s_, c in zip(srcs, cpg)], 2).view(N, groups * cin_g, H, W)
    ref = _act_ref(conv64(xcat, w, b, stride=1, padding=1, groups=groups), act, 0.2)
    layer = ops.PackedConv(w.to(dev), b.to(dev), cpg, groups=groups, stride=1, pad=1, algo="winograd")
    tol = fp32_tol(cin_g * 9, floor=3e-5)
    src_d = [nhwc(s_).to(dev) for s_ in srcs]
    fp32 = layer(src_d, act=act, slope=0.2, tile=132)
    for shape in (132, 164, 32, 5132, 6064):
        if dst_ld is None:
            out = layer(src_d, act=act, slope=0.2, tile=ops.W3_BASE + shape)
        else:
            full = torch.full((N, H, W, dst_ld), 7.0, device=dev)
            layer(src_d, out=full, out_coff=dst_coff, act=act, slope=0.2, tile=ops.W3_BASE + shape)
            out = full[..., dst_coff:dst_coff + Cout]
            rest = torch.cat([full[..., :dst_coff], full[..., dst_coff + Cout:]], 3)
            assert (rest == 7.0).all(), "winograd x3 wrote outside its channel slice"
        _pattern_on_mismatch(out, fp32, shape, case)
        assert_close(nchw(out.cpu()), ref, tol, "winograd x3 shape %d %s" % (shape, case))
        assert_close(out.cpu(), fp32.cpu(), 1.5 * tol, "winograd x3 vs fp32 winograd, shape %d %s" % (shape, case))


@pytest.mark.parametrize("shape", [132, 164, 32, 5132, 6064])
def test_conv3x3_winograd_x3_epilogues(dev, shape):
    """residual (aligned and not) and ACT_DCNPOST on the split-bf16 Winograd kernel; reruns bit-identical"""
    from e2fgvi_amd import ops
    g = _gen(67 + shape)
    x = torch.randn(2, 128, 30, 54, generator=g)
    w = torch.randn(128, 128, 3, 3, generator=g) / math.sqrt(128 * 9)
    b = torch.randn(128, generator=g)
    layer = ops.PackedConv(w.to(dev), b.to(dev), [128], pad=1, algo="winograd")
    for res_ld, res_coff in ((128, 0), (131, 3)):
        resfull = torch.randn(2, 30, 54, res_ld, generator=g)
        res = resfull[..., res_coff:res_coff + 128]
        ref = F.leaky_relu(conv64(x, w, b, padding=1) + nchw(res), 0.1)
        out = layer([nhwc(x).to(dev)], residual=resfull.to(dev), res_coff=res_coff, act=2, slope=0.1, tile=ops.W3_BASE + shape)
        assert_close(nchw(out.cpu()), ref, 3e-5, "winograd x3 + residual (ld %d coff %d) shape %d" % (res_ld, res_coff, shape))
    w2 = torch.randn(432, 128, 3, 3, generator=g) / math.sqrt(128 * 9)
    b2 = torch.randn(432, generator=g) * 0.1
    fl = torch.randn(2, 30, 54, 4, generator=g) * 3
    raw = conv64(x, w2, b2, padding=1)
    o1, o2, m = torch.chunk(raw, 3, 1)
    off = 10 * torch.tanh(torch.cat([o1, o2], 1))
    f1, f2 = fl[..., 0:2].permute(0, 3, 1, 2), fl[..., 2:4].permute(0, 3, 1, 2)
    off1, off2 = torch.chunk(off, 2, 1)
    ref = torch.cat([off1 + f1.flip(1).repeat(1, 72, 1, 1), off2 + f2.flip(1).repeat(1, 72, 1, 1), torch.sigmoid(m)], 1)
    wl = ops.PackedConv(w2.to(dev), b2.to(dev), [128], pad=1, algo="winograd")
    first = wl([nhwc(x).to(dev)], residual=fl.to(dev), act=ops.ACT_DCNPOST, slope=10.0, tile=ops.W3_BASE + shape)
    assert_close(nchw(first.cpu()), ref, 5e-5, "winograd x3 DCNPOST shape %d" % shape)
    first = first.clone()
    bad = sum(int(not torch.equal(wl([nhwc(x).to(dev)], residual=fl.to(dev), act=ops.ACT_DCNPOST, slope=10.0,
                                     tile=ops.W3_BASE + shape), first)) for _ in range(50))
    assert bad == 0, "%d of 50 launches differ" % bad


# ---------------------------------------------------------------------------------------------- attention with split operands
def _attention_case(B, T, fh, fw, seed):
    from e2fgvi_amd.engine import build_key_table
    from e2fgvi_amd.synth import rolled_valid_index
    from oracle import e2fgvi_oracle as O
    g = _gen(seed)
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
    return ref, qkv, kvp, torch.from_numpy(tab), torch.from_numpy(nk)


@pytest.mark.parametrize("B,T,fh,fw", [(1, 3, 10, 18), (2, 2, 20, 36), (1, 5, 20, 36), (1, 2, 15, 45), (1, 40, 5, 9), (1, 4, 60, 108)])
def test_focal_attention_x3(dev, B, T, fh, fw):
    """the fp32 attention's own cases (+ a 12x12-window HQ grid with 210-key windows) on the split-operand kernel, every
    workgroup shape, against the oracle's roll / partition / cat / softmax chain (tfocal_transformer.py:226-396) at the fp32
    kernel's tolerance; the planes of split3_kv sum to the k / v columns bit for bit; reruns are bit-identical"""
    from e2fgvi_amd import ops
    from tests.test_gpu_ops import ATT_TOL
    ref, qkv, kvp, tab, nk = _attention_case(B, T, fh, fw, 6 + fh)
    both = torch.cat([qkv, kvp], 0).to(dev)
    q_d, p_d = both[:qkv.shape[0]], both[qkv.shape[0]:]
    planes = ops.split3_kv(both)
    torch.cuda.synchronize()
    assert torch.equal(planes.double().sum(0).float(), both[:, 512:]) and torch.equal(planes.double().sum(0), both[:, 512:].double())
    fp32 = ops.focal_attention(q_d, p_d, tab.to(dev), nk.to(dev), B, T, fh, fw)
    for waves in (0, 2, 4, 8, 14):                # 14: four waves x two key groups (needs the window's key table + two 72 KB rings in LDS)
        if waves == 14 and T > 20:
            with pytest.raises(Exception):
                ops.focal_attention_x3(q_d, planes, tab.to(dev), nk.to(dev), B, T, fh, fw, waves=waves)
            continue
        out = ops.focal_attention_x3(q_d, planes, tab.to(dev), nk.to(dev), B, T, fh, fw, waves=waves)
        assert_close(out.cpu(), ref, ATT_TOL, "attention x3 %dx%d T=%d waves=%d" % (fh, fw, T, waves))
        # two fp32-level results, each held to ATT_TOL against the oracle above: their difference to the same figure, growing
        # with sqrt(keys) (measured over three data sets: <= 3.3e-5, profiles/r03_gpu_suite_soak_x3_summary.txt)
        assert_close(out, fp32, ATT_TOL * max(1.0, (T * 210 / 840.0) ** 0.5), "attention x3 vs the fp32 kernel, waves=%d" % waves)
    first = ops.focal_attention_x3(q_d, planes, tab.to(dev), nk.to(dev), B, T, fh, fw).clone()
    bad = sum(int(not torch.equal(ops.focal_attention_x3(q_d, planes, tab.to(dev), nk.to(dev), B, T, fh, fw), first)) for _ in range(30))
    assert bad == 0, "%d of 30 launches differ" % bad


# ---------------------------------------------------------------------------------------------- deformable conv with split operands
def test_mdcn_x3(dev):
    """the propagation's fused deformable conv (two sources, raw conv_offset output + flows, feat_prop.py:38-58) and the
    finished-offsets form on the split-operand MFMA (mfma="x3"): the fp32 kernel's bound against the oracle's
    modulated_deform_conv2d, every tile, the weight planes sum to the weights, reruns bit-identical"""
    from e2fgvi_amd import ops
    from oracle.dcn import modulated_deform_conv2d
    g = _gen(71)
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
    layer = ops.PackedDcn(w.to(dev), b.to(dev), dg, pad=1, mfma="x3")
    fp32 = ops.PackedDcn(w.to(dev), b.to(dev), dg, pad=1)
    torch.cuda.synchronize()
    # the planes against the fp32 packing (layouts: [chunk][plane][4 octets][N][8] vs [chunk][8 quads][N][4])
    w3 = layer.wpacked.view(-1, 3, 4, 128, 8).double().sum(1)                      # [chunk][octet][n][8]  -> k = octet * 8 + e
    w1 = fp32.wpacked.view(-1, 8, 128, 4).double()                                  # [chunk][quad][n][4]   -> k = quad * 4 + e
    assert torch.equal(w3.permute(0, 2, 1, 3).reshape(-1, 128, 32), w1.permute(0, 2, 1, 3).reshape(-1, 128, 32)), \
        "the three weight planes do not sum to the fp32 weights"
    flows = nhwc(torch.cat([f1, f2], 1)).to(dev)
    srcs = [nhwc(a).to(dev), nhwc(c).to(dev)]
    # (round 5: the weights no longer pass through the LDS -- every tile fits with three planes per operand, incl. the three- and
    #  four-K-group ones, 5 and 7; each tile 20 times: bit-identical reruns)
    for tile in (0, 1, 2, 3, 4, 5, 6, 7):
        out = layer(srcs, nhwc(raw).to(dev), flows=flows, max_residue=10.0, tile=tile)
        assert_close(nchw(out.cpu()), ref, 5e-5, "mdcn x3 fused tile %d" % tile)
        bad = sum(int(not torch.equal(layer(srcs, nhwc(raw).to(dev), flows=flows, max_residue=10.0, tile=tile), out)) for _ in range(20))
        assert bad == 0, "tile %d: %d of 20 launches differ" % (tile, bad)
    final = torch.cat([q1, q2, torch.sigmoid(m)], 1)
    out = layer(srcs, nhwc(final).to(dev))
    assert_close(nchw(out.cpu()), ref, 5e-5, "mdcn x3 finished offsets")
    first = out.clone()
    bad = sum(int(not torch.equal(layer(srcs, nhwc(final).to(dev)), first)) for _ in range(100))
    assert bad == 0, "%d of 100 launches differ" % bad


@pytest.mark.parametrize("rows", [7360, 256, 1000])
def test_qkv_epilogue_writes_the_attention_planes(dev, rows):
    """ABI 8 (round 5): the split-operand GEMM's epilogue writes the K / V columns of a qkv Linear as the three exact bf16 planes
    the split-operand attention reads -- bit for bit what e2fgvi_split3_kv makes of the fp32 rows of the same kernel -- and leaves
    the Q columns in the fp32 rows; every tile shape, row counts with and without a ragged last tile; then through
    PackedLinear's kv_planes (the engine's call), whichever kernel the table hands it."""
    from e2fgvi_amd import lib as L, ops
    g = _gen(4100 + rows)
    w = (torch.randn(1536, 512, generator=g) * 0.05).to(dev)
    b = torch.randn(1536, generator=g).to(dev)
    x = torch.randn(rows, 512, generator=g).to(dev)
    layer = ops.PackedConvX(w, b, [512], dtype=torch.float32, x3=True)
    x4 = x.view(rows, 1, 1, 512)
    ran = 0
    for tile in (1, 2, 4, 5, 6, 7, 8, 107, 108):
        full = torch.empty(rows, 1, 1, 1536, device=dev)
        try:
            layer([x4], out=full, tile=tile)
        except L.HipError:
            continue
        want = ops.split3_kv(full.view(rows, 1536))
        out = torch.full((rows, 1, 1, 1536), float("nan"), device=dev)
        planes = torch.zeros(3, rows, 1024, dtype=torch.bfloat16, device=dev)
        layer([x4], out=out, tile=tile, planes=planes, split_from=512)
        assert torch.equal(planes, want), "tile %d: planes differ from split3_kv of the fp32 rows" % tile
        assert torch.equal(out[..., :512], full[..., :512]), "tile %d: Q columns" % tile
        assert torch.isnan(out[..., 512:]).all(), "tile %d: the fp32 K / V columns must not be stored" % tile
        # the planes sum to the fp32 value
        assert torch.equal(planes.double().sum(0), full.view(rows, 1536)[:, 512:].double())
        ran += 1
    assert ran >= 4
    lin = ops.PackedLinear(w, b)
    lin.tune = lin.try_x3 = True
    planes = torch.zeros(3, rows, 1024, dtype=torch.bfloat16, device=dev)
    y = lin(x, kv_planes=planes)
    ref = lin(x)
    assert torch.equal(y[:, :512], ref[:, :512]) and torch.equal(planes, ops.split3_kv(ref))
    with pytest.raises(Exception):
        layer([x4], out=out, planes=planes[:, :, :1000].contiguous(), split_from=512)
