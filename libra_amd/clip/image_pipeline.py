"""Image input pipeline on MI355X: decoded uint8 images -> bf16 `pixel_values` (or directly the patch-embedding
GEMM's im2col rows), replacing the reference's per-image CPU chain
    [expand2square(mean colour)] -> CLIPImageProcessor(resize shortest edge, BICUBIC -> center crop -> /255 -> normalise) -> .to(bf16)
(/root/reference/libra/models/clip/image_processing_clip.py:219-337, libra/data/datasets/laion_dataset.py:130-136,
caption_datasets.py:45-56, libra/models/libra/tokenization_libra.py:258 `images.to(self.dtype)`).

Bit-exact with the reference's pixel_values: the resize is Pillow's 8-bit two-pass resampler, whose tap tables (fixed point, 22
fractional bits) are built here on the host exactly as Pillow's `precompute_coeffs` / `normalize_coeffs_8bpc` do (a few hundred
integers per image side, cached per size pair); the integer passes, crop, normalisation (a 3 x 256 look-up table holding the
reference's float32 arithmetic for every uint8 level) and layout run in two kernels per image (`libra_resample_h_u8`,
`libra_resample_v_u8_norm`).  Host -> device traffic is the decoded uint8 image itself, 6x less than fp32 pixel_values.
"""
from __future__ import annotations

import functools
import math
from typing import Sequence, Tuple, Union

import numpy as np
import torch

from .. import kernels as K

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
_BITS = 32 - 8 - 2


def _cubic(x: float) -> float:                                    # Pillow's bicubic_filter, a = -0.5
    x = -x if x < 0 else x
    if x < 1.0:
        return (1.5 * x - 2.5) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * -0.5
    return 0.0


@functools.lru_cache(maxsize=512)
def _taps(n_in: int, n_out: int) -> Tuple[np.ndarray, np.ndarray]:
    """Pillow's tap table for resizing n_in -> n_out samples: bounds [n_out, 2] (first source index, count), coeffs [n_out, ksize]."""
    scale = n_in / n_out
    fs = scale if scale > 1.0 else 1.0
    support = 2.0 * fs
    ss = 1.0 / fs                   # Pillow's precompute_coeffs multiplies by the reciprocal (a division can differ by one ulp
    ksize = int(math.ceil(support)) * 2 + 1          # for fs != 1 and, rarely, flip a 22-bit fixed-point coefficient)
    bounds = np.zeros((n_out, 2), dtype=np.int32)
    coeffs = np.zeros((n_out, ksize), dtype=np.int32)
    for i in range(n_out):
        center = (i + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = lo if lo > 0 else 0
        hi = int(center + support + 0.5)
        hi = hi if hi < n_in else n_in
        n = hi - lo
        w = [_cubic((j + lo - center + 0.5) * ss) for j in range(n)]
        tot = sum(w)
        for j in range(n):
            v = w[j] / tot if tot != 0.0 else w[j]
            coeffs[i, j] = int(v * (1 << _BITS) - 0.5) if v < 0 else int(v * (1 << _BITS) + 0.5)
        bounds[i] = (lo, n)
    return bounds, coeffs


def _out_size(h: int, w: int, size: int) -> Tuple[int, int]:
    short, long = (w, h) if w <= h else (h, w)
    new_long = int(size * long / short)
    return (new_long, size) if w <= h else (size, new_long)


class CLIPImagePipeline:
    """`pipeline(images) -> bf16 [B, 3, crop, crop]` on `device`; `images`: uint8 [H, W, 3] arrays / tensors (host or device) or PIL
    images, any sizes.  `pad_to_square=True` first centres each image on a square canvas of the mean colour (the reference's
    dataset code does this for image-to-text samples).  `as_patches=P` returns the patch-embedding operand [B * (crop/P)^2, Kpad]."""

    def __init__(self, size: int = 336, crop: int = 336, image_mean=OPENAI_CLIP_MEAN, image_std=OPENAI_CLIP_STD,
                 rescale_factor: float = 1 / 255, device="cuda"):
        self.size, self.crop, self.device = size, crop, torch.device(device)
        self.image_mean, self.image_std = tuple(image_mean), tuple(image_std)
        # the reference's arithmetic per uint8 level: uint8 * python float -> float64 -> float32; (x - mean) / std in float32; -> bf16
        lv = (np.arange(256, dtype=np.uint8) * rescale_factor).astype(np.float32)
        m, s = np.asarray(image_mean, dtype=np.float32), np.asarray(image_std, dtype=np.float32)
        lut = ((lv[None, :] - m[:, None]) / s[:, None]).astype(np.float32)
        self.lut = torch.from_numpy(lut).to(torch.bfloat16).to(self.device).contiguous()
        self.background = tuple(int(x * 255) for x in image_mean)
        self._dev_taps = {}

    def _taps_dev(self, n_in: int, n_out: int):
        key = (n_in, n_out)
        if key not in self._dev_taps:
            if len(self._dev_taps) > 256:
                self._dev_taps.clear()
            b, c = _taps(n_in, n_out)
            self._dev_taps[key] = (b, torch.from_numpy(b).to(self.device), torch.from_numpy(c).to(self.device))
        return self._dev_taps[key]

    def _one(self, img, pad_to_square: bool, out: torch.Tensor, patch: int, kpad: int):
        if not isinstance(img, torch.Tensor):
            img = torch.from_numpy(np.ascontiguousarray(np.asarray(img)))
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
            raise ValueError("images must be uint8 [H, W, 3] (RGB; convert PIL images with .convert('RGB'))")
        img = img.to(self.device, non_blocking=True).contiguous()
        h, w = int(img.shape[0]), int(img.shape[1])
        pad_y = pad_x = 0
        if pad_to_square and h != w:
            n = max(h, w)
            pad_y, pad_x = ((n - h) // 2, 0) if w > h else (0, (n - w) // 2)
            h = w = n
        nh, nw = _out_size(h, w, self.size)
        top, left = (nh - self.crop) // 2, (nw - self.crop) // 2
        if top < 0 or left < 0:
            raise ValueError("image smaller than the crop after resizing (not a Libra configuration)")
        bvh, bv, cv = self._taps_dev(h, nh)
        _, bh, ch = self._taps_dev(w, nw)
        # the horizontal pass only produces the canvas rows the vertical pass reads (Pillow does the same)
        first, last = int(bvh[0, 0]), int(bvh[-1, 0] + bvh[-1, 1])
        tmp = K.resample_h_u8(img, pad_y, pad_x, self.background, bh, ch, last - first, first)
        K.resample_v_u8_norm(tmp, first, bv, cv, top, left, self.crop, self.lut, out, patch, kpad)

    @torch.no_grad()
    def __call__(self, images: Sequence, pad_to_square: Union[bool, Sequence[bool]] = False, as_patches: int = 0) -> torch.Tensor:
        if not isinstance(images, (list, tuple)):
            images = [images]
        B, c = len(images), self.crop
        pads = list(pad_to_square) if isinstance(pad_to_square, (list, tuple)) else [bool(pad_to_square)] * B
        if as_patches:
            g = c // as_patches
            kpad = K.round_up(3 * as_patches * as_patches, 64)
            out = torch.zeros((B * g * g, kpad), dtype=torch.bfloat16, device=self.device)
            for i, img in enumerate(images):
                self._one(img, pads[i], out[i * g * g:(i + 1) * g * g], as_patches, kpad)
            return out
        out = torch.empty((B, 3, c, c), dtype=torch.bfloat16, device=self.device)
        for i, img in enumerate(images):
            self._one(img, pads[i], out[i], 0, 0)
        return out
