"""The sliding-window driver (e2fgvi_amd/video.py) against the numpy restatement of test.py's loop."""
import numpy as np
import pytest
import torch

from e2fgvi_amd import video
from oracle import video_ref


def _toy_video(L, h, w, seed=0):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, (h + 2 * L, w + 2 * L, 3)).astype(np.float32)
    k = np.ones((5, 5)) / 25.0
    frames = []
    for i in range(L):
        f = base[i:i + h, 2 * i:2 * i + w]
        frames.append(np.clip(f, 0, 255).astype(np.uint8))
    masks = []
    for i in range(L):
        m = np.zeros((h, w), np.uint8)
        m[h // 4 + i % 3:h // 2 + i % 3, w // 4 + i:w // 2 + i] = 255
        masks.append(m)
    return frames, masks


def _fake_model(x, n_local):
    # deterministic, batch-free stand-in with the InpaintGenerator output convention (runs on the CPU in both the
    # reference loop and the device driver, so that the comparison isolates the byte kernels)
    b, t, c, H, W = x.shape
    y = torch.tanh(x.reshape(b * t, c, H, W) * 0.7 + 0.1 * x.mean(dim=(1, 2, 3, 4)).view(1, 1, 1, 1))
    return y, None


def test_window_plan_matches_reference_loop():
    """host logic: the (neighbour, reference) ids of every window == what the reference loop selects"""
    for L, stride, ref_length, num_ref in ((23, 5, 10, -1), (12, 5, 10, 2), (7, 3, 10, -1), (100, 5, 10, -1), (41, 5, 7, 4)):
        got = video.plan_windows(L, stride, ref_length, num_ref)
        f_list = list(range(0, L, stride))
        assert len(got) == len(f_list)
        for f, (nb, rf) in zip(f_list, got):
            exp_nb = [i for i in range(max(0, f - stride), min(L, f + stride + 1))]
            assert nb == exp_nb and rf == video_ref.get_ref_index(f, exp_nb, L, ref_length, num_ref)


def test_ref_index_selection():
    assert video.get_ref_index(10, list(range(5, 16)), 40) == [0, 20, 30]
    assert video.get_ref_index(10, list(range(5, 16)), 40, 10, 2) == video_ref.get_ref_index(10, list(range(5, 16)), 40, 10, 2)


def test_padded_size():
    assert video.padded_size(240, 432) == (240, 432) and video.padded_size(720, 1280) == (720, 1296)
    assert video.padded_size(50, 70) == (60, 108) and video.padded_size(1080, 1920) == (1080, 1944)


def test_nearest_table_is_pillows():
    """the source-index tables of the mask resize == what PIL's Image.resize(size, NEAREST) actually samples (test.py:62)"""
    from PIL import Image
    rng = np.random.RandomState(0)
    sizes = [(432, 864), (240, 480), (240, 720), (432, 1280), (100, 37), (37, 100), (240, 241), (432, 431), (7, 1000), (1000, 7)]
    sizes += [(int(a), int(b)) for a, b in rng.randint(1, 700, (40, 2))]
    for n_out, n_in in sizes:
        src = np.arange(n_in, dtype=np.int32).reshape(1, n_in)
        ref = np.array(Image.fromarray(src, mode="I").resize((n_out, 1), Image.NEAREST))[0]
        assert (video.nearest_table(n_in, n_out) == ref).all(), (n_in, n_out)


def _mask_reference(masks, size_hw):
    """read_mask (test.py:56-69) with PIL for the resize and the numpy cross dilation of the oracle"""
    from PIL import Image
    out = []
    for m in masks:
        r = np.array(Image.fromarray(m).resize((size_hw[1], size_hw[0]), Image.NEAREST).convert("L"))
        out.append(video_ref.dilate_cross_np(r > 0, 4))
    return np.stack(out).astype(np.uint8)


@pytest.mark.gpu
@pytest.mark.parametrize("hw_in,hw_out", [((50, 70), (50, 70)), ((100, 141), (50, 70)), ((33, 47), (60, 108)), ((240, 432), (240, 432))])
def test_mask_prepare_kernel(dev, hw_in, hw_out):
    """resize NEAREST + binarise + 4x cross dilation on the device == PIL + numpy reference, bit-exact"""
    rng = np.random.RandomState(2)
    masks = ((rng.rand(3, *hw_in) > 0.97) * rng.randint(1, 256, (3,) + hw_in)).astype(np.uint8)
    got = video.prepare_masks(masks, hw_out, dev).cpu().numpy()
    assert (got == _mask_reference(masks, hw_out)).all()
    nod = video.prepare_masks(masks, hw_out, dev, dilate=False).cpu().numpy()
    assert nod.max() <= 1 and nod.sum() <= got.sum()


@pytest.mark.gpu
@pytest.mark.parametrize("L,stride,num_ref", [(23, 5, -1), (12, 5, 2), (7, 3, -1)])
def test_driver_kernels_match_reference_loop(dev, L, stride, num_ref):
    """masked-clip / composite / blend / uint8 kernels around a stand-in model == the numpy reference loop, bit-exact"""
    frames, masks = _toy_video(L, 50, 70)
    dil = [video_ref.dilate_cross_np(m > 0, 4) for m in masks]
    ref = video_ref.run(lambda x, n: _fake_model(x, n)[0], frames, dil, stride, 10, num_ref)
    out = video.inpaint_video(lambda x, n: (_fake_model(x.cpu(), n)[0].to(dev), None), np.stack(frames), np.stack(masks), stride, 10,
                              num_ref, device=dev)
    assert out.shape == ref.shape == (L, 50, 70, 3) and out.dtype == np.uint8
    assert np.abs(out.astype(int) - ref.astype(int)).max() == 0


@pytest.mark.gpu
def test_batched_windows_equal_sequential(dev):
    frames, masks = _toy_video(41, 40, 60, seed=5)
    fm = lambda x, n: (_fake_model_batch(x.cpu(), n)[0].to(dev), None)
    a = video.inpaint_video(fm, np.stack(frames), np.stack(masks), 5, 10, -1, device=dev)
    b = video.inpaint_video(fm, np.stack(frames), np.stack(masks), 5, 10, -1, device=dev, batch_windows=3)
    assert (a == b).all()


def _fake_model_batch(x, n_local):
    # per-clip (batch independent) stand-in
    b, t, c, H, W = x.shape
    y = torch.tanh(x * 0.7 + 0.1 * x.mean(dim=(1, 2, 3, 4), keepdim=True)).reshape(b * t, c, H, W)
    return y, None


@pytest.mark.gpu
def test_pred_to_u8_kernel(dev):
    """the uint8 packing used before the RCCL gather == the reference's uint8((pred+1)/2*255) (test.py:168-171)"""
    from e2fgvi_amd import ops
    g = torch.Generator().manual_seed(3)
    pred = torch.tanh(torch.randn(5, 3, 24, 40, generator=g) * 2)
    ref = (((pred + 1) / 2).permute(0, 2, 3, 1).numpy() * 255).astype(np.uint8)
    got = ops.pred_to_u8(pred.to(dev)).cpu().numpy()
    assert (got == ref).all()
    crop = ops.pred_to_u8(pred.to(dev), 20, 33).cpu().numpy()
    assert (crop == ref[:, :20, :33]).all()


def test_driver_refuses_cpu():
    with pytest.raises(RuntimeError):
        video.inpaint_video(lambda x, n: (x, None), np.zeros((3, 8, 8, 3), np.uint8), np.zeros((3, 8, 8), np.uint8),
                            device=torch.device("cpu"))


@pytest.mark.gpu
def test_driver_on_gpu_matches_cpu_oracle(dev):
    """whole path: HIP model inside the device driver vs oracle model inside the reference loop (uint8 frames)"""
    import importlib
    from e2fgvi_amd.synth import synth_state_dict
    from oracle import e2fgvi_oracle as O
    L, h, w = 11, 60, 100                      # padded to 60x108 by the driver
    frames, masks = _toy_video(L, h, w, seed=3)
    sd = synth_state_dict("e2fgvi_hq", "stress", 0)
    net = importlib.import_module("model.e2fgvi_hq").InpaintGenerator()
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    out = video.inpaint_video(net, np.stack(frames), np.stack(masks), 5, 10, -1)
    dil = [video_ref.dilate_cross_np(m > 0, 4) for m in masks]
    ref = video_ref.run(lambda x, n: O.forward(sd, x, n, "e2fgvi_hq")[0], frames, dil, 5, 10, -1)
    d = np.abs(out.astype(int) - ref.astype(int))
    # |pred error| <= 1e-3 -> at most one grey level where the float lands next to an integer
    assert d.max() <= 1 and (d > 0).mean() < 0.02, (d.max(), (d > 0).mean())


@pytest.mark.gpu
@pytest.mark.parametrize("model,hw,L,precision", [("e2fgvi", (240, 432), 36, "fp32"), ("e2fgvi_hq", (120, 200), 23, "bf16")])
def test_windows_in_flight_return_the_bytes_of_the_sequential_driver(dev, model, hw, L, precision):
    """video.inpaint_video(in_flight=K), round 6: the forwards of K consecutive windows on K streams (eager launches, the engine's
    SPyNet / propagation-split side work on a side stream PER main stream), compositing in window order on the caller's stream --
    the same bytes as one window at a time, five times over.  (With one side stream shared by both main streams the caching allocator
    handed a flow tensor of window i to window i + 1's side work while window i was still reading it: different bytes.)"""
    import importlib
    from e2fgvi_amd.synth import synth_state_dict
    frames, masks = _toy_video(L, hw[0], hw[1], seed=9)
    net = importlib.import_module("model." + model).InpaintGenerator()
    net.load_state_dict(synth_state_dict(model, "stress", 0))
    net = net.to(dev).eval()
    net.precision = precision
    ref = video.inpaint_video(net, np.stack(frames), np.stack(masks), 5, 10, -1)
    for k in (2, 3, 2, 2, 3):
        out = video.inpaint_video(net, np.stack(frames), np.stack(masks), 5, 10, -1, in_flight=k)
        assert np.array_equal(out, ref), "in_flight=%d: %d bytes differ" % (k, int((out != ref).sum()))


@pytest.mark.gpu
def test_sharded_runner_uint8_gather(dev):
    """inpaint_sharded(pack_u8=True): the frames every rank contributes to the gather are the uint8 NHWC form"""
    from e2fgvi_amd.runner import inpaint_sharded
    clips = torch.randn(2, 3, 3, 16, 24, device=dev)
    net = lambda x, n: (torch.tanh(x.reshape(-1, 3, 16, 24)), None)
    out = inpaint_sharded(net, clips, 2, 0, 1, pack_u8=True)
    ref = (((torch.tanh(clips.reshape(-1, 3, 16, 24)).cpu() + 1) / 2).permute(0, 2, 3, 1).numpy() * 255).astype(np.uint8)
    assert out.dtype == torch.uint8 and tuple(out.shape) == (6, 16, 24, 3) and (out.cpu().numpy() == ref).all()
