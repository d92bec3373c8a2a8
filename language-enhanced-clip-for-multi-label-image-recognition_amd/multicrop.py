"""Sliding-window multi-crop test input: the reference's ``DatasetWrapperWithBlock``
(dassl/data/data_manager.py:311-492) rebuilt for the device.

The reference cuts ~570 windows (scales 2..5 of a 480x640 image) out of every test image on the HOST, pushes each one
through PIL (``tfm(F.to_pil_image(block))``: bicubic Resize of the smaller edge to 224, CenterCrop, ToTensor, Normalize)
and ships ``img_blocks`` = one ``[n_s, 3, 224, 224]`` tensor per scale to the GPU.  Here the raw uint8 image goes to HBM
once; the window list is integer arithmetic on (height, width) alone (``enumerate_windows``); and one kernel
(``leclip_crop_resize_fwd``) produces every window's transformed tensor directly in the layout the image tower reads,
bit-compatible with Pillow's resampler.  The windows then simply are more batch through the same hot path.

Window = (y0, x0, rows, cols, pad_top): rows are counted on the image after ``pad_top`` reflected rows were put on top
(reflection continues below the last row as far as needed); columns are never padded and windows are cut at the right
edge.  That asymmetry is the reference's: it calls torchvision's ``F.pad(img, (0, padding_w, 0, padding_h), 'reflect')``,
whose 4-tuple means (left, top, right, bottom) - the amount computed for the width lands on TOP - and then slices past the
right edge (data_manager.py:390-396).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

CLIP_PIXEL_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_PIXEL_STD = (0.26862954, 0.26130258, 0.27577711)
MAX_SCALE = 15.5   # taps of the device resampler: 2 * ceil(2 * scale) + 1 <= 64


def _grid(n_i: int, n_j: int, stride_h: int, stride_w: int):
    i, j = np.meshgrid(np.arange(n_i, dtype=np.int64), np.arange(n_j, dtype=np.int64), indexing="ij")
    return (i * stride_h).ravel(), (j * stride_w).ravel()


def _strides(bs: int, bh: int, bw: int, nh: int, nw: int) -> Tuple[int, int]:
    return ((bs - 1) * bh) // (nh - 1) + 1, ((bs - 1) * bw) // (nw - 1) + 1


def enumerate_windows(height: int, width: int, multi_scale: Sequence[int] = (2, 3, 4, 5)) -> List[np.ndarray]:
    """One int32 array [n_s, 5] per scale, rows (y0, x0, rows, cols, pad_top), in the reference's order."""
    h, w = int(height), int(width)
    per_scale = []
    for bs in multi_scale:
        parts = []
        # (1) square windows on a 2bs x 2bs grid over the reflect-padded image (data_manager.py:385-399)
        n = 2 * bs
        bh, bw = h // bs, w // bs
        sh, sw = _strides(bs, bh, bw, n, n)
        top = sw * (n - 1) - (bs - 1) * bw - w % bs            # the reference's padding_w, applied to the top
        bottom = sh * (n - 1) - (bs - 1) * bh - h % bs         # its padding_h, applied to the bottom
        y0, x0 = _grid(n, n, sh, sw)
        rows = np.minimum(bh, h + top + bottom - y0)
        cols = np.minimum(bw, w - x0)
        if (rows <= 0).any() or (cols <= 0).any():
            raise ValueError(f"image {h}x{w}: scale {bs} produces an empty square window (the reference fails on it too)")
        parts.append(np.stack([y0, x0, rows, cols, np.full_like(y0, top)], axis=1))
        # (2) 1x2 / 2x1, (3) 2x3 / 3x2 of a cell, (4, bs >= 3) 2x3 / 3x2 cells: unpadded, cut at the borders, empty ones skipped
        shapes = [((h // bs, w * 2 // bs), (bs * 2, bs)), ((h * 2 // bs, w // bs), (bs, bs * 2)),
                  ((h // bs, w * 3 // (2 * bs)), (bs * 2, bs * 4 // 3)), ((h * 3 // (2 * bs), w // bs), (bs * 4 // 3, bs * 2))]
        if bs >= 3:
            shapes += [((h * 2 // bs, w * 3 // bs), (bs, bs * 2 // 3)), ((h * 3 // bs, w * 2 // bs), (bs * 2 // 3, bs))]
        for (bh, bw), (nh, nw) in shapes:
            sh, sw = _strides(bs, bh, bw, nh, nw)
            y0, x0 = _grid(nh, nw, sh, sw)
            rows, cols = np.minimum(bh, h - y0), np.minimum(bw, w - x0)
            keep = (rows > 0) & (cols > 0)
            parts.append(np.stack([y0, x0, rows, cols, np.zeros_like(y0)], axis=1)[keep])
        per_scale.append(np.concatenate(parts).astype(np.int32))
    return per_scale


def full_image_window(height: int, width: int) -> np.ndarray:
    return np.array([[0, 0, height, width, 0]], dtype=np.int32)


def check_windows(windows: np.ndarray, size: int):
    short = np.minimum(windows[:, 2], windows[:, 3]).astype(np.float64)
    if (short / size > MAX_SCALE).any():
        raise ValueError(f"window with smaller edge > {MAX_SCALE} x {size}: outside the device resampler's tap budget")


class MultiCropper:
    """uint8 images [B,3,H,W] on the device -> (img [B,3,S,S], img_blocks: one [B, n_s, 3, S, S] per scale): the two entries a
    DatasetWrapperWithBlock item carries (data_manager.py:336-341), for a whole batch of equally sized images at once."""

    def __init__(self, size: int = 224, multi_scale: Sequence[int] = (2, 3, 4, 5), mean=CLIP_PIXEL_MEAN, std=CLIP_PIXEL_STD,
                 dtype: torch.dtype = torch.float32):
        self.size, self.multi_scale, self.mean, self.std, self.dtype = int(size), tuple(multi_scale), tuple(mean), tuple(std), dtype
        self._cache = {}

    def windows(self, height: int, width: int, device):
        key = (height, width, str(device))
        if key not in self._cache:
            per_scale = enumerate_windows(height, width, self.multi_scale) if self.multi_scale else []
            flat = np.concatenate([full_image_window(height, width)] + per_scale)
            check_windows(flat, self.size)
            self._cache[key] = (torch.from_numpy(flat).to(device), [len(p) for p in per_scale])
        return self._cache[key]

    def __call__(self, src_u8: torch.Tensor):
        from .hip import ops
        if src_u8.dim() == 3:
            src_u8 = src_u8.unsqueeze(0)
        b, _, h, w = src_u8.shape
        win, counts = self.windows(h, w, src_u8.device)
        out = ops.crop_resize(src_u8.contiguous(), win, self.size, self.mean, self.std, self.dtype)    # [B, 1 + sum n_s, 3, S, S]
        img = out[:, 0]
        blocks, o = [], 1
        for n in counts:
            blocks.append(out[:, o:o + n])
            o += n
        return img, blocks
