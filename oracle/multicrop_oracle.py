"""CPU oracle for the multi-crop test path: window enumeration and the per-window test transform.

TEST INFRASTRUCTURE ONLY (imported by tests/ alone).  Two restatements:

* ``windows(h, w, multi_scale)`` - the crop rectangles of ``DatasetWrapperWithBlock._transform_image``
  (dassl/data/data_manager.py:348-492), integer arithmetic only.  Pinned: ``oracle/make_golden.py`` executes the
  reference's own source lines with recording stubs and stores every window's footprint (``tests/golden/multicrop.npz``).
* ``transform_window(...)`` - what the reference's ``tfm`` does to a window (dassl/data/transforms/transforms.py:379-400 with
  the shipped cfg: torchvision ``Resize(224, bicubic)`` on the smaller edge, ``CenterCrop(224)``, ``ToTensor``,
  ``Normalize``).  The resize is Pillow's ``ImagingResample`` (third-party: Pillow, any version with the 8-bit fixed-point
  resampler of src/libImaging/Resample.c - 22 fractional bits, horizontal pass then vertical pass, uint8 in between;
  torchvision 0.12.0 for the size rule) restated in numpy.  Pinned against Pillow itself through the same fixture file.
"""
from __future__ import annotations

import math
from typing import List, Sequence

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _add(out, y0, x0, bh, bw, top, h, w, bottom=0):
    hp = h + top + bottom
    bh = min(bh, hp - y0)
    bw = min(bw, w - x0)
    assert bh > 0 and bw > 0
    out.append((y0, x0, bh, bw, top))


def windows(h: int, w: int, multi_scale: Sequence[int] = (2, 3, 4, 5)) -> List[np.ndarray]:
    """Per scale: int64 [n, 5] rows (y0, x0, bh, bw, pad_top).  y0 counts rows of the image after ``pad_top`` reflected rows
    were put on top (and reflected rows below as needed); columns are never padded, windows are cut at the right edge.
    data_manager.py:359-490; the square windows' `F.pad(img, (0, padding_w, 0, padding_h), 'reflect')` is torchvision's
    (left, top, right, bottom) order: `padding_w` rows on TOP, `padding_h` rows at the bottom, no columns."""
    res = []
    for bs in multi_scale:
        out = []
        # square sliding windows (:385-399)
        slide = bs * 2
        bh, bw = h // bs, w // bs
        sh, sw = ((bs - 1) * bh) // (slide - 1) + 1, ((bs - 1) * bw) // (slide - 1) + 1
        pad_h = sh * (slide - 1) - ((bs - 1) * bh) - h % bs
        pad_w = sw * (slide - 1) - ((bs - 1) * bw) - w % bs
        # (a negative amount crops instead - torch's pad semantics - which the row mapping r - pad_top and the cut at the padded
        # height h + pad_w + pad_h cover as they stand)
        for i in range(slide):
            for j in range(slide):
                _add(out, i * sh, j * sw, bh, bw, pad_w, h, w, pad_h)
        # 1x2 / 2x1, 2x3 / 3x2 and (bs >= 3) 2x3-of-bs windows (:401-488): no padding, cut at the borders, empty ones skipped
        groups = [([(h // bs, w * 2 // bs), (h * 2 // bs, w // bs)], [(bs * 2, bs), (bs, bs * 2)]),
                  ([(h // bs, w * 3 // (2 * bs)), (h * 3 // (2 * bs), w // bs)], [(bs * 2 // 1, bs * 2 * 2 // 3), (bs * 2 * 2 // 3, bs * 2 // 1)])]
        if bs >= 3:
            groups.append(([(h * 2 // bs, w * 3 // bs), (h * 3 // bs, w * 2 // bs)], [(bs * 2 // 2, bs * 2 // 3), (bs * 2 // 3, bs * 2 // 2)]))
        for blocks, slides in groups:
            for (bh, bw), (nh, nw) in zip(blocks, slides):
                sh, sw = ((bs - 1) * bh) // (nh - 1) + 1, ((bs - 1) * bw) // (nw - 1) + 1
                for i in range(nh):
                    for j in range(nw):
                        ch, cw = min(bh, h - i * sh), min(bw, w - j * sw)
                        if ch <= 0 or cw <= 0:
                            continue
                        out.append((i * sh, j * sw, ch, cw, 0))
        res.append(np.array(out, dtype=np.int64))
    return res


def source_rows(y0: int, bh: int, pad_top: int, h: int) -> np.ndarray:
    """Padded-row index -> source row (reflect padding without repeating the edge row, top and bottom)."""
    r = np.arange(y0, y0 + bh) - pad_top
    r = np.where(r < 0, -r, r)
    return np.where(r > h - 1, 2 * (h - 1) - r, r)


def footprint(win, h: int) -> np.ndarray:
    """The 8 numbers make_golden's recording `tfm` stores per window: (rows, cols, y[0,0], x[0,0], y[-1,-1], x[-1,-1], sum y, sum x)."""
    y0, x0, bh, bw, top = (int(v) for v in win)
    rows = source_rows(y0, bh, top, h)
    return np.array([bh, bw, rows[0], x0, rows[-1], x0 + bw - 1, int(rows.sum()) * bw, (x0 * bw + bw * (bw - 1) // 2) * bh], dtype=np.int64)


# ------------------------------------------------------------------------------------------- Pillow's bicubic resampler
def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _coeffs(in_size: int, out_size: int):
    """precompute_coeffs + normalize_coeffs_8bpc (Resample.c): per output index (xmin, int32 taps)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ss = 1.0 / filterscale
    res = []
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in k:
            ww += v
        if ww != 0.0:
            k = [v / ww for v in k]
        kk = [int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)) for v in k]
        res.append((xmin, np.array(kk, dtype=np.int64)))
    return res


def _resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """One pass over `axis` of a uint8 [C, H, W] array: sum of taps in 32-bit fixed point, + half, >> 22, clip to 0..255."""
    in_size = img.shape[axis]
    if in_size == out_size:
        return img
    src = np.moveaxis(img, axis, -1).astype(np.int64)
    out = np.empty(src.shape[:-1] + (out_size,), dtype=np.uint8)
    for xx, (xmin, kk) in enumerate(_coeffs(in_size, out_size)):
        acc = (src[..., xmin:xmin + len(kk)] * kk).sum(-1) + (1 << (PRECISION_BITS - 1))
        out[..., xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, -1, axis)


def resized_size(h: int, w: int, size: int):
    """torchvision 0.12.0 transforms/functional.py `_compute_resized_output_size` for an int size (smaller edge -> size)."""
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long_ / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)     # (new_h, new_w)


def transform_window(src: np.ndarray, win, size: int, mean, std):
    """src uint8 [3, H, W]; returns (uint8 [3, size, size] after resize + centre crop, float32 normalised tensor)."""
    _, h, w = src.shape
    y0, x0, bh, bw, top = (int(v) for v in win)
    crop = src[:, source_rows(y0, bh, top, h)][:, :, x0:x0 + bw]
    nh, nw = resized_size(bh, bw, size)
    r = _resample_axis(crop, nw, 2)          # horizontal pass first (ImagingResample), uint8 between the passes
    r = _resample_axis(r, nh, 1)
    t, l = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))     # CenterCrop: Python round (half to even)
    u8 = r[:, t:t + size, l:l + size]
    f = u8.astype(np.float32) / np.float32(255)
    f = (f - np.asarray(mean, np.float32)[:, None, None]) / np.asarray(std, np.float32)[:, None, None]
    return u8, f.astype(np.float32)
