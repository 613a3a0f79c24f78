"""Sliding-window video inpainting driver on device.

Mirrors the reference's demo loop -- ``test.py:39-53`` (reference-frame selection), ``:56-69`` (mask NEAREST resize,
binarise, 4x cross dilation), ``:146-179`` (neighbour window of +-stride frames every ``stride`` frames, mirror padding
to multiples of (60,108), compositing with the mask, 0.5/0.5 blending of overlapping predictions) -- without cv2 /
torchvision / PIL: frames and masks come in as uint8 arrays, everything after the upload runs in HIP kernels
(csrc/video.hip) and only the finished uint8 frames are copied back.  The window planning below is host logic; the byte
arithmetic has no CPU path.
"""
import numpy as np
import torch

from . import ops


def get_ref_index(f, neighbor_ids, length, ref_length=10, num_ref=-1):
    """test.py:39-53"""
    ref_index = []
    if num_ref == -1:
        for i in range(0, length, ref_length):
            if i not in neighbor_ids:
                ref_index.append(i)
    else:
        start_idx = max(0, f - ref_length * (num_ref // 2))
        end_idx = min(length, f + ref_length * (num_ref // 2))
        for i in range(start_idx, end_idx + 1, ref_length):
            if i not in neighbor_ids:
                if len(ref_index) > num_ref:
                    break
                ref_index.append(i)
    return ref_index


def plan_windows(length, neighbor_stride=5, ref_length=10, num_ref=-1):
    """The (neighbour ids, reference ids) of every window of test.py:146-153, in the reference's order."""
    windows = []
    for f in range(0, length, neighbor_stride):
        neighbor_ids = list(range(max(0, f - neighbor_stride), min(length, f + neighbor_stride + 1)))
        windows.append((neighbor_ids, get_ref_index(f, neighbor_ids, length, ref_length, num_ref)))
    return windows


def padded_size(h, w, mod_h=60, mod_w=108):
    """test.py:156-159: H, W rounded up to multiples of (60,108)."""
    return h + (mod_h - h % mod_h) % mod_h, w + (mod_w - w % mod_w) % mod_w


def nearest_table(n_in, n_out):
    """Source index of every output index of PIL's ``Image.resize(size, Image.NEAREST)`` (test.py:62).  Pillow's
    ImagingScaleAffine starts at 0.5 * scale and ADDS the scale once per output pixel in double precision, then
    truncates; the running sum is reproduced literally (np.cumsum accumulates sequentially)."""
    scale = float(n_in) / float(n_out)
    steps = np.full(n_out, scale, dtype=np.float64)
    steps[0] = 0.0 + scale * 0.5
    pos = np.cumsum(steps)
    return np.minimum(pos.astype(np.int64), n_in - 1).astype(np.int32)


def prepare_masks(masks_u8, size_hw, device, dilate=True):
    """uint8 masks [L,Hin,Win] (any size, any non-zero = hole) -> device uint8 [L,H,W] of 0/1 like test.py:56-69."""
    m = torch.as_tensor(np.ascontiguousarray(masks_u8)).to(device)
    H, W = size_hw
    ytab = torch.from_numpy(nearest_table(m.shape[1], H)).to(device)
    xtab = torch.from_numpy(nearest_table(m.shape[2], W)).to(device)
    return ops.mask_prepare(m, ytab, xtab, H, W, 4 if dilate else 0)


@torch.no_grad()
def inpaint_video(model, frames_u8, masks_u8, neighbor_stride=5, ref_length=10, num_ref=-1, dilate=True,
                  device=None, pad=True, batch_windows=1, keep_float=False, in_flight=1):
    """frames_u8: uint8 [L,H,W,3]; masks_u8: [L,Hm,Wm] (non-zero = hole; resized to the frames with NEAREST like
    read_mask).  Returns uint8 [L,H,W,3] composited frames, computed like test.py:129-179.
    ``model(masked[b,t,3,H',W'], n_local) -> (pred[b*t,3,H',W'], _)`` on the device.

    ``batch_windows`` > 1 runs windows of equal shape (same number of local and reference frames) as one forward of
    b clips -- clips are independent, so the predictions are the same; the compositing / blending is still applied in
    the reference's window order (the 0.5/0.5 blend is order dependent).

    keep_float=True returns the blended frames as the fp32 device tensor [L,H,W,3] they are before the final
    ``astype(uint8)`` -- what evaluate.py:113-114 feeds to calc_psnr_and_ssim.

    ``in_flight`` = K > 1 (round 6, with batch_windows = 1): the forwards of K consecutive windows run on K streams -- window
    i + 1's encoder fills the CUs window i's one-frame propagation chain leaves idle -- while the compositing
    stays on the caller's stream in the reference's window order: the same kernels on the same data, the same bytes."""
    if device is None:
        device = next(model.parameters()).device if hasattr(model, "parameters") else torch.device("cuda")
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("inpaint_video runs on the MI355X (cuda) device only; there is no CPU path")
    frames_d = torch.as_tensor(np.ascontiguousarray(frames_u8)).to(device)
    L, h, w, _ = frames_d.shape
    masks01 = prepare_masks(masks_u8, (h, w), device, dilate)
    Hp, Wp = padded_size(h, w) if pad else (h, w)
    windows = plan_windows(L, neighbor_stride, ref_length, num_ref)
    # every host->device upload happens here, before the first forward: frame ids of each window and the
    # "first prediction of this frame" flags of the reference's comp_frames[idx] is None test (test.py:172-176)
    ids_dev = [torch.tensor(nb + rf, dtype=torch.int32, device=device) for nb, rf in windows]
    seen = [False] * L
    first_dev = []
    for nb, _ in windows:
        first_dev.append(torch.tensor([0 if seen[j] else 1 for j in nb], dtype=torch.uint8, device=device))
        for j in nb:
            seen[j] = True
    comp = torch.empty((L, h, w, 3), dtype=torch.float32, device=device)

    def predict(group):
        x = torch.cat([ops.masked_clip(frames_d, masks01, ids_dev[i], Hp, Wp) for i in group], 0) if len(group) > 1 \
            else ops.masked_clip(frames_d, masks01, ids_dev[group[0]], Hp, Wp)
        n_local = len(windows[group[0]][0])
        pred, _ = model(x, n_local)
        t = x.shape[1]
        return [pred[k * t:k * t + n_local] for k in range(len(group))]

    def composite(i, pred):
        n = len(windows[i][0])
        ops.composite(pred.contiguous(), ids_dev[i][:n], first_dev[i], frames_d, masks01, comp)

    if batch_windows <= 1 and in_flight > 1:
        cur = torch.cuda.current_stream(device)
        streams = [torch.cuda.Stream(device=device) for _ in range(in_flight)]
        for st in streams:
            st.wait_stream(cur)                         # the uploads, the mask preparation
        queue = []                                      # (window, its local predictions, event) in window order
        for i in range(len(windows)):
            st = streams[i % in_flight]
            with torch.cuda.stream(st):
                p = predict([i])[0]
                ev = torch.cuda.Event()
                ev.record(st)
            p.record_stream(cur)                        # allocated on st, read by the compositing kernel on cur
            queue.append((i, p, ev))
            if len(queue) == in_flight:
                j, pj, evj = queue.pop(0)
                cur.wait_event(evj)
                composite(j, pj)
        for j, pj, evj in queue:
            cur.wait_event(evj)
            composite(j, pj)
    elif batch_windows <= 1:
        # like the reference: every window is composited right after its forward, nothing is kept
        for i in range(len(windows)):
            composite(i, predict([i])[0])
    else:
        by_shape = {}
        for i, (nb, rf) in enumerate(windows):
            by_shape.setdefault((len(nb), len(rf)), []).append(i)
        groups = sorted((idx[k:k + batch_windows] for idx in by_shape.values() for k in range(0, len(idx), batch_windows)),
                        key=lambda g: g[0])
        # the 0.5/0.5 blend is order dependent: predictions are composited in the reference's window order as soon as
        # all earlier windows are done; a window that has to wait keeps only its local frames (a copy, so the batch
        # output with the reference frames' predictions is released)
        pending, nxt = {}, 0
        for grp in groups:
            for i, p in zip(grp, predict(grp)):
                if i == nxt:
                    composite(i, p)
                    nxt += 1
                else:
                    pending[i] = p.clone()
            while nxt in pending:
                composite(nxt, pending.pop(nxt))
                nxt += 1
        assert not pending and nxt == len(windows)
    if keep_float:
        return comp
    return ops.float_to_u8(comp).cpu().numpy()
