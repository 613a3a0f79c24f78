"""Sliding-window video inpainting driver on device.

Mirrors the reference's demo loop -- ``test.py:39-53`` (reference-frame selection), ``:57-70`` (mask
binarise + 4x cross dilation), ``:146-179`` (neighbour window of +-stride frames every ``stride`` frames,
mirror padding to multiples of (60,108), compositing with the mask, 0.5/0.5 blending of overlapping
predictions) -- without cv2 / torchvision: frames and masks come in as uint8 arrays, everything after the
upload happens on the GPU, and only the finished uint8 frames are copied back.
"""
import numpy as np
import torch


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


def dilate_cross(mask, iterations=4):
    """cv2.dilate(m, getStructuringElement(MORPH_CROSS, (3, 3)), iterations=4) for a binary [L,H,W] tensor
    (test.py:64-68): each iteration ORs the 4-neighbourhood; pixels outside the image do not contribute."""
    m = mask.bool()
    for _ in range(iterations):
        n = m.clone()
        n[:, 1:] |= m[:, :-1]
        n[:, :-1] |= m[:, 1:]
        n[:, :, 1:] |= m[:, :, :-1]
        n[:, :, :-1] |= m[:, :, 1:]
        m = n
    return m


def mirror_pad(x, mod_h=60, mod_w=108):
    """test.py:156-165: pad H, W up to multiples of (60,108) by appending the flipped clip and cropping."""
    h, w = x.shape[-2:]
    h_pad = (mod_h - h % mod_h) % mod_h
    w_pad = (mod_w - w % mod_w) % mod_w
    x = torch.cat([x, torch.flip(x, [3])], 3)[:, :, :, :h + h_pad, :]
    x = torch.cat([x, torch.flip(x, [4])], 4)[:, :, :, :, :w + w_pad]
    return x


@torch.no_grad()
def inpaint_video(model, frames_u8, masks_u8, neighbor_stride=5, ref_length=10, num_ref=-1, dilate=True,
                  device=None, pad=True, batch_windows=1):
    """frames_u8: uint8 [L,H,W,3]; masks_u8: [L,H,W] (non-zero = hole).  Returns uint8 [L,H,W,3] composited
    frames, computed like test.py:129-179.  ``model(masked[b,t,3,H',W'], n_local) -> (pred[b*t,3,H',W'], _)``.

    ``batch_windows`` > 1 runs windows of equal shape (same number of local and reference frames) as one forward of
    b clips -- clips are independent, so the predictions are the same; the compositing / blending below is still
    applied in the reference's window order (the 0.5/0.5 blend is order dependent)."""
    frames_u8 = torch.as_tensor(np.asarray(frames_u8))
    masks_u8 = torch.as_tensor(np.asarray(masks_u8))
    if device is None:
        device = next(model.parameters()).device if hasattr(model, "parameters") else torch.device("cpu")
    L, h, w, _ = frames_u8.shape
    frames_d = frames_u8.to(device)
    binary = (masks_u8.to(device) > 0)
    if dilate:
        binary = dilate_cross(binary, 4)
    imgs = (frames_d.permute(0, 3, 1, 2).float() / 255.0).unsqueeze(0) * 2 - 1            # to_tensors()*2-1
    masks = binary.float().view(1, L, 1, h, w)
    bmask = binary.view(L, h, w, 1)

    windows = []
    for f in range(0, L, neighbor_stride):
        neighbor_ids = list(range(max(0, f - neighbor_stride), min(L, f + neighbor_stride + 1)))
        windows.append((neighbor_ids, get_ref_index(f, neighbor_ids, L, ref_length, num_ref)))

    def predict(group):
        clips = []
        for neighbor_ids, ref_ids in group:
            ids = neighbor_ids + ref_ids
            masked = imgs[:, ids] * (1 - masks[:, ids])
            clips.append(mirror_pad(masked) if pad else masked)
        x = torch.cat(clips, 0).contiguous()
        n_local = len(group[0][0])
        pred, _ = model(x, n_local)
        t = x.shape[1]
        pred = (pred[:, :, :h, :w] + 1) / 2
        pred = (pred.permute(0, 2, 3, 1) * 255).view(len(group), t, h, w, 3)             # float, [b,t,h,w,3]
        return [pred[i, :n_local] for i in range(len(group))]

    preds = [None] * len(windows)
    if batch_windows <= 1:
        order = [[i] for i in range(len(windows))]
    else:
        by_shape = {}
        for i, (nb, rf) in enumerate(windows):
            by_shape.setdefault((len(nb), len(rf)), []).append(i)
        order = [idx[k:k + batch_windows] for idx in by_shape.values() for k in range(0, len(idx), batch_windows)]
    for grp in order:
        for i, p in zip(grp, predict([windows[i] for i in grp])):
            preds[i] = p

    comp = [None] * L
    for (neighbor_ids, _), pred in zip(windows, preds):
        for i, idx in enumerate(neighbor_ids):
            # np.array(pred).astype(uint8) * mask + frame * (1 - mask)   (test.py:171-174)
            img = torch.where(bmask[idx], pred[i].to(torch.uint8), frames_d[idx])
            if comp[idx] is None:
                comp[idx] = img
            else:
                comp[idx] = comp[idx].float() * 0.5 + img.float() * 0.5
    out = torch.stack([c.to(torch.uint8) for c in comp], 0)
    return out.cpu().numpy()
