"""numpy restatement of the reference demo loop test.py:129-179 (TEST INFRASTRUCTURE): same arithmetic and
dtype conversions as the reference, around a pluggable CPU model function."""
import numpy as np
import torch


def get_ref_index(f, neighbor_ids, length, ref_length, num_ref):     # test.py:39-53
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


def dilate_cross_np(m, iterations=4):
    """cv2.dilate with a 3x3 cross, default (constant, ignored) border -- test.py:64-68"""
    m = m.astype(bool)
    for _ in range(iterations):
        p = np.pad(m, 1)
        m = p[1:-1, 1:-1] | p[:-2, 1:-1] | p[2:, 1:-1] | p[1:-1, :-2] | p[1:-1, 2:]
    return m.astype(np.uint8)


def run(model_fn, frames, masks, neighbor_stride=5, ref_length=10, num_ref=-1, pad=True, as_float=False):
    """frames: list of uint8 [h,w,3]; masks: list of uint8 [h,w] (already binary 0/1, dilated)."""
    video_length = len(frames)
    h, w = frames[0].shape[:2]
    imgs = torch.from_numpy(np.stack(frames)).permute(0, 3, 1, 2).float().div(255).unsqueeze(0) * 2 - 1
    binary_masks = [np.expand_dims((m != 0).astype(np.uint8), 2) for m in masks]
    mt = torch.from_numpy(np.stack(masks).astype(np.float32)).view(1, video_length, 1, h, w)
    comp_frames = [None] * video_length
    for f in range(0, video_length, neighbor_stride):
        neighbor_ids = [i for i in range(max(0, f - neighbor_stride), min(video_length, f + neighbor_stride + 1))]
        ref_ids = get_ref_index(f, neighbor_ids, video_length, ref_length, num_ref)
        selected_imgs = imgs[:1, neighbor_ids + ref_ids]
        selected_masks = mt[:1, neighbor_ids + ref_ids]
        masked_imgs = selected_imgs * (1 - selected_masks)
        mod_size_h, mod_size_w = 60, 108
        h_pad = (mod_size_h - h % mod_size_h) % mod_size_h if pad else 0      # evaluate.py:86-92 does not pad
        w_pad = (mod_size_w - w % mod_size_w) % mod_size_w if pad else 0
        masked_imgs = torch.cat([masked_imgs, torch.flip(masked_imgs, [3])], 3)[:, :, :, :h + h_pad, :]
        masked_imgs = torch.cat([masked_imgs, torch.flip(masked_imgs, [4])], 4)[:, :, :, :, :w + w_pad]
        pred_imgs = model_fn(masked_imgs, len(neighbor_ids))
        pred_imgs = pred_imgs[:, :, :h, :w]
        pred_imgs = (pred_imgs + 1) / 2
        pred_imgs = pred_imgs.cpu().permute(0, 2, 3, 1).numpy() * 255
        for i in range(len(neighbor_ids)):
            idx = neighbor_ids[i]
            img = np.array(pred_imgs[i]).astype(np.uint8) * binary_masks[idx] + frames[idx] * (1 - binary_masks[idx])
            if comp_frames[idx] is None:
                comp_frames[idx] = img
            else:
                comp_frames[idx] = comp_frames[idx].astype(np.float32) * 0.5 + img.astype(np.float32) * 0.5
    if as_float:                                                           # evaluate.py:113 passes these on as they are
        return [c for c in comp_frames]
    return np.stack([c.astype(np.uint8) for c in comp_frames])
