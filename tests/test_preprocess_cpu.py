"""CPU pin of the image-pipeline oracle: against Pillow itself (the third-party resampler the reference calls:
image_processing_clip.py `resize` -> transformers.image_transforms.resize -> PIL.Image.resize) and against pixel_values produced by
the reference's own CLIPImageProcessor (tests/golden/clip_preprocess.safetensors, made by tests/golden/make_golden_preprocess.py)."""
import numpy as np
import pytest
import torch

from helpers import load_golden
from oracle import preprocess_oracle as PO


@pytest.mark.parametrize("h,w,oh,ow", [(37, 53, 20, 31), (100, 80, 336, 420), (500, 375, 448, 336), (336, 400, 336, 336),
                                       (64, 64, 64, 64), (9, 700, 5, 336), (123, 77, 123, 40)])
def test_resize_equals_pillow_bit_for_bit(h, w, oh, ow):
    from PIL import Image
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    want = np.asarray(Image.fromarray(img).resize((ow, oh), resample=Image.BICUBIC, reducing_gap=None))
    got = PO.pil_bicubic_resize(img, ow, oh)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_expand2square_and_sizes():
    img = np.arange(2 * 5 * 3, dtype=np.uint8).reshape(2, 5, 3)
    sq = PO.expand2square(img, (122, 116, 104))
    assert sq.shape == (5, 5, 3) and np.array_equal(sq[1:3], img) and tuple(sq[0, 0]) == (122, 116, 104)
    assert PO.shortest_edge_size(480, 640, 336) == (336, 448) and PO.shortest_edge_size(640, 480, 336) == (448, 336)
    assert PO.shortest_edge_size(333, 1000, 336) == (336, int(336 * 1000 / 333))


def test_pipeline_matches_reference_processor_fixture():
    t, meta = load_golden("clip_preprocess.safetensors")
    for i, case in enumerate(meta["cases"]):
        img = t[f"in.{i}"].numpy()
        got = PO.clip_preprocess(img, size=case["size"], crop=case["size"], pad_to_square=case["pad_to_square"])
        want = t[f"out.{i}"].numpy()
        sub = got[:, ::case["stride"], ::case["stride"]]
        assert sub.shape == want.shape
        assert np.array_equal(sub, want), (i, case, float(np.abs(sub - want).max()))
        assert abs(float(got.astype(np.float64).sum()) - case["sum"]) < 1e-6 * max(1.0, abs(case["sum"]))


def test_product_tap_tables_equal_the_oracle():
    """Host logic of the product pipeline (libra_amd/clip/image_pipeline.py): Pillow's tap tables and the output sizing."""
    from libra_amd.clip.image_pipeline import _out_size, _taps
    for a, b in [(53, 31), (80, 420), (375, 336), (400, 336), (64, 64), (700, 336), (77, 40), (1000, 336), (336, 1000)]:
        b1, c1 = _taps(a, b)
        b2, c2 = PO.resample_coeffs(a, b)
        assert np.array_equal(b1, b2) and np.array_equal(c1, c2), (a, b)
    for h, w in [(480, 640), (640, 480), (333, 1000), (336, 336), (90, 700)]:
        assert _out_size(h, w, 336) == PO.shortest_edge_size(h, w, 336)
