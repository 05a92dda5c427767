"""Image input pipeline on device against the oracle (itself pinned on Pillow and on the reference's
CLIPImageProcessor, tests/test_preprocess_cpu.py) and against the reference fixture: bit-exact after the bf16 cast."""
import numpy as np
import pytest
import torch

from helpers import load_golden, parity_report

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def test_pipeline_vs_reference_fixture_bit_exact():
    from libra_amd.clip import CLIPImagePipeline
    t, meta = load_golden("clip_preprocess.safetensors")
    pipes = {s: CLIPImagePipeline(size=s, crop=s) for s in {c["size"] for c in meta["cases"]}}
    n = 0
    for i, case in enumerate(meta["cases"]):
        out = pipes[case["size"]]([t[f"in.{i}"]], pad_to_square=case["pad_to_square"])[0].cpu()
        want = t[f"out.{i}"].to(BF)                                   # LibraTokenizer: images.to(self.dtype)
        got = out[:, ::case["stride"], ::case["stride"]]
        assert got.shape == want.shape and torch.equal(got, want), (i, case)
        n += want.numel()
    parity_report(f"[f3 image pipeline] {len(meta['cases'])} images (down / up-scaling, both orientations, pad-to-square): "
                  f"{n} bf16 pixel values bit-identical to the reference CLIPImageProcessor's (cast to bf16)")


def test_batch_of_mixed_sizes_and_patch_layout_vs_oracle():
    from libra_amd import kernels as K
    from libra_amd.clip import CLIPImagePipeline
    from oracle import preprocess_oracle as PO
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, s, dtype=np.uint8) for s in [(500, 375, 3), (336, 600, 3), (97, 131, 3), (336, 336, 3)]]
    pads = [False, True, False, False]
    pipe = CLIPImagePipeline()
    out = pipe(imgs, pad_to_square=pads)
    assert out.shape == (4, 3, 336, 336) and out.dtype == BF
    for i, (img, pad) in enumerate(zip(imgs, pads)):
        want = torch.from_numpy(PO.clip_preprocess(img, pad_to_square=pad)).to(BF)
        assert torch.equal(out[i].cpu(), want), i
    # device-resident uint8 input and the fused patch layout: exactly what patch_im2col makes of the pixel_values
    cols = pipe([torch.from_numpy(im).cuda() for im in imgs], pad_to_square=pads, as_patches=14)
    ref_cols = K.patch_im2col(out.contiguous(), 14, K.round_up(3 * 14 * 14, 64))
    assert cols.shape == ref_cols.shape and torch.equal(cols, ref_cols)
    with pytest.raises(ValueError):
        pipe([np.zeros((10, 10), dtype=np.uint8)])
