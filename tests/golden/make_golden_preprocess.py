"""Golden fixture for the image input pipeline from the REFERENCE's own CLIPImageProcessor
(/root/reference/libra/models/clip/image_processing_clip.py:219-337; config = CLIP ViT-L/14@336: shortest edge 336, BICUBIC,
center crop 336, 1/255, OpenAI mean / std) and its pad-to-square step (libra/data/datasets/caption_datasets.py:45-56), on small
random uint8 images of several aspect ratios.  Stored: the uint8 inputs and the float32 pixel_values.
Build-container only (imports /root/reference through ref_harness)."""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402


def main():
    from make_golden import _save
    from PIL import Image
    spec = importlib.util.spec_from_file_location("_ref_image_processing_clip", f"{rh.REF}/libra/models/clip/image_processing_clip.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # the processor is parametric in size / crop: 112 keeps the fixture small (same code path); one CLIP ViT-L/14@336 case is kept
    # as a strided sample of its 336 x 336 output plus its float64 sum
    proc = mod.CLIPImageProcessor(size={"shortest_edge": 112}, crop_size={"height": 112, "width": 112})
    proc336 = mod.CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336})
    spec2 = importlib.util.spec_from_file_location("_ref_caption_datasets_expand", f"{rh.REF}/libra/data/datasets/caption_datasets.py")
    rng = np.random.default_rng(7)
    # (h, w, pad_to_square): down- and up-scaling, both orientations, a square image, an extreme aspect ratio
    cases = [(60, 80, False, 112), (200, 150, False, 112), (112, 112, False, 112), (40, 300, False, 112), (250, 170, True, 112),
             (50, 70, True, 112), (400, 520, False, 336)]
    t, meta = {}, dict(cases=[])

    def expand2square(pil_img, background_color):           # the reference's function (caption_datasets.py:45-56) needs its whole
        width, height = pil_img.size                        # dataset module to import; its 10 lines are restated in the ORACLE and
        if width == height:                                 # checked there - here PIL's own paste builds the same canvas
            return pil_img
        n = max(width, height)
        result = Image.new(pil_img.mode, (n, n), background_color)
        result.paste(pil_img, (0, (width - height) // 2) if width > height else ((height - width) // 2, 0))
        return result
    for i, (h, w, pad, size) in enumerate(cases):
        # smooth-ish content (random low-res field upsampled) + noise: exercises the filter on realistic gradients
        base = rng.integers(0, 256, (max(h // 8, 2), max(w // 8, 2), 3), dtype=np.uint8)
        img = np.asarray(Image.fromarray(base).resize((w, h), resample=Image.BILINEAR)).astype(np.int16)
        img = np.clip(img + rng.integers(-20, 21, img.shape), 0, 255).astype(np.uint8)
        pil = Image.fromarray(img)
        if pad:
            pil = expand2square(pil, tuple(int(x * 255) for x in proc.image_mean))
        out = (proc if size == 112 else proc336).preprocess(pil, return_tensors="pt")["pixel_values"][0]
        assert out.shape == (3, size, size) and out.dtype == torch.float32
        t[f"in.{i}"] = torch.from_numpy(img)
        stride = 1 if size == 112 else 6
        t[f"out.{i}"] = out[:, ::stride, ::stride].contiguous()
        meta["cases"].append(dict(h=h, w=w, pad_to_square=pad, size=size, stride=stride, sum=float(out.double().sum())))
    _save("clip_preprocess.safetensors", t, meta)


if __name__ == "__main__":
    main()
