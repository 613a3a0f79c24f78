"""Stage-level and end-to-end parity of the HIP forward against the CPU oracle (oracle/e2fgvi_oracle.py),
on identical synthetic weights and clips.  Tolerances: the north star's fp32 max-abs 1e-3 on the output
frames, plus scale-free per-stage checks (max err relative to the rms of the oracle tensor) that still
bite at the reference's tiny default init."""
import pytest
import torch

from tests.util import assert_bound, assert_close, err, nchw, nhwc

pytestmark = pytest.mark.gpu

_CACHE = {}


def _setup(model, kind, hw, t, lt, b=1, seed=1):
    key = (model, kind, hw, t, lt, b, seed)
    if key not in _CACHE:
        from e2fgvi_amd.synth import synth_clip, synth_state_dict
        from oracle import e2fgvi_oracle as O
        torch.set_num_threads(max(1, torch.get_num_threads()))
        sd = synth_state_dict(model, kind, 0)
        x, _ = synth_clip(b, t, hw[0], hw[1], seed=seed, moving=True)
        tr = {}
        out, flows = O.forward(sd, x, lt, model, tr)
        _CACHE[key] = (sd, x, tr, out, flows)
    return _CACHE[key]


def _engine(model, kind, dev):
    key = ("eng", model, kind)
    if key not in _CACHE:
        from e2fgvi_amd.engine import Engine
        from e2fgvi_amd.synth import synth_state_dict
        _CACHE[key] = Engine(synth_state_dict(model, kind, 0), model, dev)
    return _CACHE[key]


CFG = [("e2fgvi", "stress", (240, 432), 3, 3), ("e2fgvi_hq", "stress", (120, 216), 4, 3),
       # round 6: the trained-regime stand-in (synth.py "peaked": sharp attention, saturated DCN offsets / masks, flows of several px)
       ("e2fgvi", "peaked", (240, 432), 3, 3), ("e2fgvi_hq", "peaked", (120, 216), 4, 3)]


@pytest.mark.parametrize("model,kind,hw,t,lt", CFG)
def test_stage_flows(dev, model, kind, hw, t, lt):
    sd, x, tr, out, flows = _setup(model, kind, hw, t, lt)
    eng = _engine(model, kind, dev)
    fwd, bwd = eng.flows(x.to(dev), lt)
    b = x.shape[0]
    h, w = hw[0] // 4, hw[1] // 4
    assert_close(fwd.cpu().permute(0, 1, 4, 2, 3), flows[0], 1e-3, "flow fwd")
    assert_close(bwd.cpu().permute(0, 1, 4, 2, 3), flows[1], 1e-3, "flow bwd")


@pytest.mark.parametrize("model,kind,hw,t,lt", CFG)
def test_stage_encoder(dev, model, kind, hw, t, lt):
    sd, x, tr, out, flows = _setup(model, kind, hw, t, lt)
    eng = _engine(model, kind, dev)
    enc = eng.encode(x.to(dev))
    assert_close(nchw(enc.cpu()), tr["enc"], 1e-4, "encoder")


@pytest.mark.parametrize("model,kind,hw,t,lt", CFG)
def test_stage_propagation(dev, model, kind, hw, t, lt):
    """fed with the oracle's encoder features and flows"""
    sd, x, tr, out, flows = _setup(model, kind, hw, t, lt)
    eng = _engine(model, kind, dev)
    b = x.shape[0]
    enc = tr["enc"]
    _, c, h, w = enc.shape
    loc = enc.view(b, t, c, h, w)[:, :lt].permute(1, 0, 3, 4, 2).contiguous().to(dev)     # [l_t,b,h,w,C]
    fa = flows[0].permute(0, 1, 3, 4, 2).contiguous().to(dev)
    fb = flows[1].permute(0, 1, 3, 4, 2).contiguous().to(dev)
    prop = eng.propagate(loc, fa, fb)
    ref = tr["prop"].view(b, t, c, h, w)[:, :lt].permute(1, 0, 3, 4, 2)
    assert_close(prop.cpu(), ref, 2e-4, "propagation")


@pytest.mark.parametrize("model,hw,lt", [("e2fgvi", (240, 432), 6), ("e2fgvi_hq", (120, 216), 4), ("e2fgvi", (240, 432), 3)])
def test_propagation_split_matches_the_whole_layers(dev, model, hw, lt):
    """engine.PROP_SPLIT (round 5): the non-recurrent input channels of conv_offset.0 / backbone.0 batched on the side stream and
    added as the per-step layers' residual == the whole layers, up to the order of the fp32 sums; the two-stream run returns the
    bits of the serial one; several clips per forward keep the whole layers (stress weights: real deformable offsets)"""
    eng = _engine(model, "stress", dev)
    assert eng.prop_split, "the split layers were not built"
    h, w = hw[0] // 4, hw[1] // 4
    g = torch.Generator().manual_seed(97 + lt)
    loc = torch.randn(lt, 1, h, w, 128, generator=g).to(dev)
    fa = (2.0 * torch.randn(1, lt - 1, h, w, 2, generator=g)).to(dev)
    fb = (2.0 * torch.randn(1, lt - 1, h, w, 2, generator=g)).to(dev)
    split = eng.propagate(loc, fa, fb).clone()
    keep, flows_overlap = eng.prop_split, eng.overlap_flows
    try:
        eng.overlap_flows = False
        serial = eng.propagate(loc, fa, fb).clone()
        eng.prop_split = {}
        whole = eng.propagate(loc, fa, fb).clone()
    finally:
        eng.prop_split, eng.overlap_flows = keep, flows_overlap
    torch.cuda.synchronize()
    assert torch.equal(split, serial), "side-stream and serial runs of the split layers differ"
    assert_close(split.cpu(), whole.cpu(), 1e-4, "propagation, split against whole layers (l_t = %d)" % lt)
    # two clips: the whole layers (identical to a run without the split layers)
    loc2 = torch.cat([loc, loc.flip(0)], 1).contiguous()
    fa2, fb2 = torch.cat([fa, fb], 0).contiguous(), torch.cat([fb, fa], 0).contiguous()
    two = eng.propagate(loc2, fa2, fb2).clone()
    try:
        eng.prop_split = {}
        two_whole = eng.propagate(loc2, fa2, fb2).clone()
    finally:
        eng.prop_split = keep
    assert torch.equal(two, two_whole)
    assert_close(two[:, 0].cpu(), whole[:, 0].cpu(), 1e-4, "clip 0 of two against the one-clip run")


@pytest.mark.parametrize("model,kind,hw,t,lt", CFG)
def test_stage_transformer(dev, model, kind, hw, t, lt):
    sd, x, tr, out, flows = _setup(model, kind, hw, t, lt)
    eng = _engine(model, kind, dev)
    from e2fgvi_amd.engine import token_grid
    b = x.shape[0]
    prop = tr["prop"]                                                     # [b,t,C,h,w]
    _, _, c, h, w = prop.shape
    fh, fw = token_grid(h, w)
    feat = nhwc(prop.reshape(b * t, c, h, w)).to(dev)
    tok = eng.soft_split(feat).view(-1, 512)
    assert_close(tok.cpu(), tr["tokens0"].reshape(-1, 512), 1e-4, "soft split")
    for i in range(8):
        tin = tr["tokens%d" % i].reshape(-1, 512).contiguous().to(dev)
        tout, x1 = eng.block(i, tin, b, t, fh, fw, (h, w))
        assert_close(x1.cpu(), tr["block%d_attn_out" % i].reshape(-1, 512), 1e-4, "block %d attention" % i)
        assert_close(tout.cpu(), tr["tokens%d" % (i + 1)].reshape(-1, 512), 1e-4, "block %d" % i)
    dec_in = eng.compose(tr["tokens8"].reshape(-1, 512).contiguous().to(dev), feat, b, t, fh, fw)
    assert_close(nchw(dec_in.cpu()), tr["dec_in"].reshape(b * t, c, h, w), 1e-4, "soft composite")
    dec = eng.decode(nhwc(tr["dec_in"].reshape(b * t, c, h, w)).to(dev))
    assert_close(dec.cpu(), out, 1e-4, "decoder")


E2E = [("e2fgvi", "stress", (240, 432), 3, 3, 1), ("e2fgvi", "default", (240, 432), 3, 3, 1),
       ("e2fgvi", "stress", (240, 432), 5, 3, 1), ("e2fgvi_hq", "stress", (120, 216), 4, 3, 1),
       ("e2fgvi_hq", "default", (120, 216), 4, 4, 1), ("e2fgvi_hq", "stress", (120, 216), 3, 2, 2),
       # minimum local window (l_t = 2) with many reference frames; two clips with reference frames
       ("e2fgvi", "stress", (240, 432), 6, 2, 1), ("e2fgvi", "stress", (240, 432), 3, 2, 2),
       # non-square window grids: 3x2 windows (token grid 15x18), 1x2 windows, a single window
       ("e2fgvi_hq", "stress", (180, 216), 3, 3, 1), ("e2fgvi_hq", "stress", (60, 216), 4, 2, 1),
       ("e2fgvi_hq", "default", (60, 108), 2, 2, 3),
       # SURVEY.md 8(d) C2, second split: T = 10 with 5 local + 5 reference frames (configs/train_e2fgvi.json:9-10) -- 10-frame
       # batches in the encoder / transformer / decoder (the kernels the decision table selects only at full size), 5-frame
       # propagation; and two clips of it
       ("e2fgvi", "stress", (240, 432), 10, 5, 1), ("e2fgvi", "default", (240, 432), 10, 5, 1), ("e2fgvi", "stress", (240, 432), 10, 5, 2),
       # a single local frame (test.py on a 1-frame video): empty flow tensors, propagation without neighbours
       ("e2fgvi_hq", "stress", (60, 108), 3, 1, 1), ("e2fgvi", "stress", (240, 432), 1, 1, 1),
       # round 6: the "peaked" weights (synth.py): sharp attention (mean largest probability 0.5-0.7), residual DCN offsets at the
       # +-10 px tanh limit, saturated masks, non-uniform pool_layers -- the stand-in for trained weights
       ("e2fgvi", "peaked", (240, 432), 5, 3, 1), ("e2fgvi_hq", "peaked", (120, 216), 4, 3, 1), ("e2fgvi", "peaked", (240, 432), 10, 5, 1)]


@pytest.mark.parametrize("model,kind,hw,t,lt,b", E2E)
def test_end_to_end(dev, model, kind, hw, t, lt, b):
    """the drop-in module, loaded through load_state_dict, against the oracle: max|d| <= 1e-3 (north star)"""
    import importlib
    sd, x, tr, out, flows = _setup(model, kind, hw, t, lt, b)
    net = importlib.import_module("model." + model).InpaintGenerator()
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    with torch.no_grad():
        got, (ff, fb) = net(x.to(dev), lt)
    assert tuple(got.shape) == (b * t, 3, hw[0], hw[1])
    d, r = err(got, out)
    assert tuple(ff.shape) == tuple(flows[0].shape) == (b, lt - 1, 2, hw[0] // 4, hw[1] // 4) and tuple(fb.shape) == tuple(ff.shape)
    if lt > 1:
        df, rf = err(ff, flows[0])
        db, rb = err(fb, flows[1])
        assert_bound(df, 1e-3 * max(1.0, flows[0].abs().max().item()), "e2e flow fwd %s %s %s" % (model, kind, hw))
        assert_bound(db, 1e-3 * max(1.0, flows[1].abs().max().item()), "e2e flow bwd %s %s %s" % (model, kind, hw))
    print("e2e %s %s: out max abs %.3e (%.2e x rms)" % (model, kind, d, r))
    assert torch.isfinite(got).all()
    tag = "e2e %s %s %s t=%d lt=%d b=%d" % (model, kind, hw, t, lt, b)
    assert_bound(d, 1e-3, tag + " output max abs (north star 1e-3)")
    # measured on MI355X over four data sets: <= 2.1e-5 x rms (gpurun_out/soak, profiles/r03_gpu_suite_soak_summary.txt); the bound
    # is 10 x that -- a 1e-3 absolute bound alone is 12 % of the output rms at default init and would let a 1 % bug in a kernel
    # that only runs at full size pass
    assert_bound(r, 2e-4, tag + " output max abs / rms")


# --------------------------------------------------------------------------- golden fixtures (real reference)
import glob
import os

import numpy as np

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "g[0-9]*_*.npz")),
              key=lambda q: int(os.path.basename(q).split("_")[0][1:]))


@pytest.mark.parametrize("path", GOLD, ids=os.path.basename)
def test_hip_matches_reference_golden(dev, path):
    """HIP forward vs sub-sampled outputs of the REAL reference (tests/golden/make_golden.py), <= 1e-3."""
    import importlib
    from e2fgvi_amd.synth import synth_state_dict
    from tests.util import golden_case
    z, model, kind, x, lt, so, sf = golden_case(path)
    net = importlib.import_module("model." + model).InpaintGenerator()
    net.load_state_dict(synth_state_dict(model, kind, 0))
    net = net.to(dev).eval()
    out, (ff, fb) = net(x.to(dev), lt)
    out, ff, fb = out.cpu(), ff.cpu(), fb.cpu()
    d = np.abs(out[:, :, ::so, ::so].numpy() - z["out_sub"]).max()
    rms = float(z["out_stats"][2])
    print("golden %s: max abs %.3e (rms of reference %.3e)" % (os.path.basename(path), d, rms))
    tag = "golden " + os.path.basename(path)
    assert_bound(d, 1e-3, tag + " output max abs (north star 1e-3)")
    assert_bound(d, 2e-4 * rms, tag + " output max abs vs 2e-4 x rms (10 x the measured worst)")
    fmax = max(1.0, float(z["flow_fwd_stats"][3]))
    assert_bound(np.abs(ff[..., ::sf, ::sf].numpy() - z["flow_fwd_sub"]).max(), 1e-3 * fmax, tag + " flow fwd")
    assert_bound(np.abs(fb[..., ::sf, ::sf].numpy() - z["flow_bwd_sub"]).max(), 1e-3 * fmax, tag + " flow bwd")
    assert_bound(np.abs(out.double().mean(dim=(1, 2, 3)).numpy() - z["out_frame_mean"]).max(), 1e-4, tag + " frame means")


# --------------------------------------------------------------------------- full-size properties (T=10)
def test_full_size_properties(dev):
    """BASELINE config (432x240, T=10): determinism, clip independence (the sharding premise: a batch of two
    clips == the two clips run alone) and l_t split consistency of the non-local frames' encoder path."""
    import importlib
    from e2fgvi_amd.synth import synth_clip, synth_state_dict
    net = importlib.import_module("model.e2fgvi").InpaintGenerator()
    net.load_state_dict(synth_state_dict("e2fgvi", "stress", 0))
    net = net.to(dev).eval()
    x, _ = synth_clip(2, 10, 240, 432, seed=31, moving=True)
    x = x.to(dev)
    o2, (f2, _) = net(x, 10)
    oa, (fa, _) = net(x[:1], 10)
    ob, _ = net(x[1:], 10)
    oa2, _ = net(x[:1], 10)
    assert tuple(o2.shape) == (20, 3, 240, 432) and torch.isfinite(o2).all()
    assert torch.equal(oa, oa2), "forward is not deterministic"
    assert_bound((o2 - torch.cat([oa, ob])).abs().max().item(), 1e-5, "clip independence (b=2 vs 2 x b=1)")
    assert_bound((f2[:1] - fa).abs().max().item(), 1e-5, "clip independence, flows")
    assert o2.abs().max().item() <= 1.0            # tanh range


def test_full_size_against_oracle(dev):
    """the BASELINE configuration itself (432x240, T = l_t = 10, default random-init weights and stress weights)
    against the CPU oracle: max |d| <= 1e-3 on the output frames (north star)"""
    import importlib
    from e2fgvi_amd.synth import synth_clip, synth_state_dict
    from oracle import e2fgvi_oracle as O
    x, _ = synth_clip(1, 10, 240, 432, seed=100)
    for kind in ("default", "stress"):
        sd = synth_state_dict("e2fgvi", kind, 0)
        net = importlib.import_module("model.e2fgvi").InpaintGenerator()
        net.load_state_dict(sd)
        net = net.to(dev).eval()
        got, (ff, fb) = net(x.to(dev), 10)
        ref, (rf, rb) = O.forward(sd, x, 10, "e2fgvi")
        d, r = err(got, ref)
        print("full size %s: max abs %.3e (%.2e x rms)" % (kind, d, r))
        assert_bound(d, 1e-3, "full size 432x240 T=10 %s output max abs (north star 1e-3)" % kind)
        assert_bound(r, 2e-4, "full size 432x240 T=10 %s output max abs / rms (10 x the measured worst)" % kind)
        assert_bound(err(ff, rf)[0], 1e-3 * max(1.0, rf.abs().max().item()), "full size %s flows" % kind)


def test_argument_errors(dev):
    """the drop-in raises where the reference would fail or silently mis-compute"""
    import importlib
    from e2fgvi_amd.synth import synth_state_dict
    base = importlib.import_module("model.e2fgvi").InpaintGenerator().to(dev).eval()
    hq = importlib.import_module("model.e2fgvi_hq").InpaintGenerator().to(dev).eval()
    x = torch.zeros(1, 3, 3, 240, 432, device=dev)
    with pytest.raises(ValueError):
        base(x, 0)                                   # l_t must be >= 1
    with pytest.raises(ValueError):
        base(x, 4)                                   # l_t > t
    with pytest.raises(ValueError):
        base(torch.zeros(1, 3, 3, 120, 216, device=dev), 2)      # base model is fixed to 432x240
    with pytest.raises(ValueError):
        hq(torch.zeros(1, 3, 3, 100, 216, device=dev), 2)        # H not a multiple of 60 (caller must pad)
    with pytest.raises(RuntimeError):
        hq(torch.zeros(1, 3, 3, 60, 108), 2)                     # CPU tensor


def test_sharded_step_over_rccl(dev):
    """runner.ShardedStep with backend "nccl" (= RCCL), world_size 1, force_gather: the pipelined all-gather (fp32 and the
    uint8 form) around the HIP-graph replay returns exactly the plain forward's frames, step after step."""
    import importlib
    import socket
    import torch.distributed as dist
    from e2fgvi_amd import ops, runner
    from e2fgvi_amd.synth import synth_clip, synth_state_dict
    net = importlib.import_module("model.e2fgvi_hq").InpaintGenerator()
    net.load_state_dict(synth_state_dict("e2fgvi_hq", "stress", 0))
    net = net.to(dev).eval()
    x = synth_clip(2, 3, 60, 108, seed=43, moving=True)[0].to(dev)
    eager, _ = net(x, 2)
    torch.cuda.synchronize()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        for pack in (False, True):
            step = runner.ShardedStep(net, x, 2, group_world=1, use_graph=True, force_gather=True, pack_u8=pack)
            want = ops.pred_to_u8(eager) if pack else eager
            got = [step.run() for _ in range(4)]
            assert got[0] is None and step.graphed
            for g in got[1:]:
                assert torch.equal(g, want)
            assert torch.equal(step.finish(), want)
    finally:
        dist.destroy_process_group()


def test_graph_replay_equals_eager(dev):
    """the HIP-graph replay used by bench.py returns exactly the eager result, run after run"""
    import importlib
    from e2fgvi_amd import runner
    from e2fgvi_amd.synth import synth_clip, synth_state_dict
    net = importlib.import_module("model.e2fgvi_hq").InpaintGenerator()
    net.load_state_dict(synth_state_dict("e2fgvi_hq", "stress", 0))
    net = net.to(dev).eval()
    x = synth_clip(1, 4, 120, 216, seed=41, moving=True)[0].to(dev)
    eager, _ = net(x, 3)
    step = runner.ShardedStep(net, x, 3, use_graph=True)
    a = step.run().clone()
    b = step.run().clone()
    c = step.run().clone()
    assert step.graphed
    assert torch.equal(a, eager) and torch.equal(b, eager) and torch.equal(c, eager)


def test_stream_overlap_is_bit_identical_to_serial(dev):
    """SPyNet runs on a side stream next to the encoder (Engine.overlap_flows), in the fp32 and in the bf16 mode: the
    two-stream schedule must give exactly the single-stream result, every time.  Tripwire for the cross-stream corruption
    of round 1 (packed-fp32 VALU of side-stream kernels beside bf16 MFMA tiles, tools/probe/overlap_probe.hip): in the
    round-1 build 60-70 % of the bf16 forwards differed."""
    from e2fgvi_amd.engine import Engine
    from e2fgvi_amd.synth import synth_clip, synth_state_dict
    sd = synth_state_dict("e2fgvi", "stress", 0)
    # t = 4: many trials; t = 10, l_t = 10: the headline shapes, i.e. the kernels the decision table selects only for 10-frame
    # batches (round 4: conv_wino_x3w beside SPyNet -- builds of it that were clean alone returned wrong encoder blocks in one
    # overlapped forward out of three: registers of loads in flight behind the K loop, reused by the compiler -- DESIGN.md C4)
    for precision, t, lt, trials in (("fp32", 4, 3, 200), ("bf16", 4, 3, 200), ("fp32", 10, 10, 40)):
        x = synth_clip(1, t, 240, 432, seed=3, moving=True)[0].to(dev)
        eng = Engine(sd, "e2fgvi", dev, precision=precision)
        assert eng.overlap_flows
        eng.overlap_flows = False
        base, (bf, bb) = eng.forward(x, lt)
        torch.cuda.synchronize()
        eng.overlap_flows = True
        for _ in range(trials):
            got, (ff, fb) = eng.forward(x, lt)
            torch.cuda.synchronize()
            assert torch.equal(ff, bf) and torch.equal(fb, bb) and torch.equal(got, base), (precision, t)
