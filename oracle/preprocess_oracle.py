"""CPU oracle for the image input pipeline: decoded uint8 image -> CLIP pixel_values.

TEST INFRASTRUCTURE ONLY (see vit_oracle.py header for the import rule).

What the reference does (/root/reference/libra/models/clip/image_processing_clip.py:219-337, driven by
/root/reference/libra/data/datasets/laion_dataset.py:130-136):
    [expand2square with the mean colour]  (caption_datasets.py:45-56)
    -> resize so that the SHORTEST edge is `size` (aspect kept, long edge = int(size * long / short)), PIL BICUBIC on uint8
    -> center crop to crop x crop -> * 1/255 -> (x - mean) / std  (float32)  -> CHW;  LibraTokenizer then casts to bf16.

The resize is Pillow's `ImagingResample` (third-party, not in /root/reference; pinned version in this image: Pillow 12.2.0,
src/libImaging/Resample.c): a separable two-pass filter on 8-bit pixels with
    scale = in / out, filterscale = max(scale, 1), support = 2 * filterscale (bicubic, a = -0.5),
    per output pixel: center = (i + 0.5) * scale, taps [xmin, xmin + n) = [int(center - support + 0.5), int(center + support + 0.5))
    clipped to the image, weights bicubic((x + xmin - center + 0.5) / filterscale) normalised to sum 1, converted to fixed point
    int(w * 2^22 +- 0.5), pixel = clip8((2^21 + sum tap * coeff) >> 22); horizontal pass first (only the rows the vertical pass
    reads), uint8 in between.
Restated here in numpy (integer arithmetic: bit-exact), pinned by tests/test_preprocess_cpu.py against Pillow itself on random
images and against pixel_values produced by the reference's own CLIPImageProcessor (tests/golden/clip_preprocess.safetensors)."""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2
OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """-> bounds int32 [out, 2] (first tap, tap count), coeffs int32 [out, ksize] (fixed point, 22 fractional bits)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray, axis: int) -> np.ndarray:
    """one separable pass over `axis` (0 = rows / vertical, 1 = columns / horizontal) of a uint8 [H, W, C] image."""
    src = img.astype(np.int64)
    out_n = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = out_n
    out = np.empty(shape, dtype=np.uint8)
    for i in range(out_n):
        x0, n = int(bounds[i, 0]), int(bounds[i, 1])
        k = kk[i, :n].astype(np.int64)
        if axis == 1:
            acc = (src[:, x0:x0 + n, :] * k[None, :, None]).sum(1)
        else:
            acc = (src[x0:x0 + n, :, :] * k[:, None, None]).sum(0)
        acc = (acc + (1 << (PRECISION_BITS - 1))) >> PRECISION_BITS
        v = np.clip(acc, 0, 255).astype(np.uint8)
        if axis == 1:
            out[:, i, :] = v
        else:
            out[i] = v
    return out


def pil_bicubic_resize(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """uint8 [H, W, C] -> uint8 [out_h, out_w, C], equal to PIL.Image.resize((out_w, out_h), BICUBIC) bit for bit."""
    h, w = img.shape[:2]
    if w != out_w:
        bh, kh = resample_coeffs(w, out_w)
        if h != out_h:                                       # the horizontal pass only produces the rows the vertical pass reads
            bv, kv = resample_coeffs(h, out_h)
            first, last = int(bv[0, 0]), int(bv[-1, 0] + bv[-1, 1])
            tmp = _pass(img[first:last], bh, kh, 1)
            bv = bv.copy(); bv[:, 0] -= first
            return _pass(tmp, bv, kv, 0)
        return _pass(img, bh, kh, 1)
    if h != out_h:
        bv, kv = resample_coeffs(h, out_h)
        return _pass(img, bv, kv, 0)
    return img.copy()


def expand2square(img: np.ndarray, background: Sequence[int]) -> np.ndarray:
    h, w = img.shape[:2]
    if h == w:
        return img
    n = max(h, w)
    out = np.empty((n, n, img.shape[2]), dtype=np.uint8)
    out[:] = np.asarray(background, dtype=np.uint8)
    if w > h:
        out[(w - h) // 2:(w - h) // 2 + h] = img
    else:
        out[:, (h - w) // 2:(h - w) // 2 + w] = img
    return out


def shortest_edge_size(h: int, w: int, size: int) -> Tuple[int, int]:
    """-> (new_h, new_w): shortest edge = size, long edge = int(size * long / short)  (HF get_resize_output_image_size)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def normalize_lut(mean=OPENAI_CLIP_MEAN, std=OPENAI_CLIP_STD, rescale: float = 1 / 255) -> np.ndarray:
    """float32 [3, 256]: the value every uint8 level maps to - `image * scale` is float64 (uint8 array times a Python float), cast
    to float32, then (x - mean) / std with mean / std in float32 (image_transforms.normalize)."""
    lv = (np.arange(256, dtype=np.uint8) * rescale).astype(np.float32)
    m, s = np.asarray(mean, dtype=np.float32), np.asarray(std, dtype=np.float32)
    return ((lv[None, :] - m[:, None]) / s[:, None]).astype(np.float32)


def clip_preprocess(img: np.ndarray, *, size: int = 336, crop: int = 336, pad_to_square: bool = False, mean=OPENAI_CLIP_MEAN,
                    std=OPENAI_CLIP_STD) -> np.ndarray:
    """uint8 [H, W, 3] -> float32 [3, crop, crop] (the reference's pixel_values before the bf16 cast)."""
    if pad_to_square:
        img = expand2square(img, tuple(int(x * 255) for x in mean))
    h, w = img.shape[:2]
    nh, nw = shortest_edge_size(h, w, size)
    r = pil_bicubic_resize(img, nw, nh)
    top, left = (nh - crop) // 2, (nw - crop) // 2
    if top < 0 or left < 0:
        raise ValueError("image smaller than the crop after resizing (the reference zero-pads; not a Libra configuration)")
    c = r[top:top + crop, left:left + crop]
    lut = normalize_lut(mean, std)
    return np.stack([lut[ch][c[:, :, ch]] for ch in range(3)])
