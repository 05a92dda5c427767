"""Throughput of the two "next" rows that so far had parity only (missing #5):
  f3  the CLIP image pipeline on the device (uint8 HxWx3 -> bicubic resize -> centre crop -> normalise -> bf16 336x336,
      libra_amd/clip/image_pipeline.py): images/s and algorithmic GB/s (uint8 in + bf16 out);
  f2  the VQ decoder (LFQ codes -> post_quant_conv -> taming Decoder -> 336 px image): images/s at a STATED synthetic decoder
      configuration (the released `ddconfig` ships only inside the checkpoint, SURVEY 7: "missing artefacts") - 24 x 24 codes,
      z_channels 256, ch 128, ch_mult (1, 2, 4), 2 res blocks per level, attention at 24 x 24, 24 -> 48 -> 336 px.
One JSON line each."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch


def timeit(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def f3():
    from libra_amd.clip import CLIPImagePipeline
    rng = np.random.default_rng(0)
    B = 32
    imgs = [torch.from_numpy(rng.integers(0, 256, (500, 375, 3), dtype=np.uint8)).cuda() for _ in range(B)]
    pipe = CLIPImagePipeline()
    dt = timeit(lambda: pipe(imgs), 10)
    by = B * (500 * 375 * 3 + 3 * 336 * 336 * 2)
    dtp = timeit(lambda: pipe(imgs, as_patches=14), 10)
    print(json.dumps({"row": "f3 image pipeline", "batch": B, "input": "500x375x3 uint8 on the device", "images_per_s": round(B / dt, 1),
                      "ms_per_batch": round(dt * 1e3, 3), "algorithmic_GBps": round(by / dt / 1e9, 2),
                      "as_patches_images_per_s": round(B / dtp, 1),
                      "note": "two separable bicubic passes per image, one launch pair per image: launch-bound at this size"}), flush=True)


def f2():
    from test_vq_decode_gpu import _build
    dd = dict(select_layer=[-2, -3], z_channels=256, ch=128, out_ch=3, ch_mult=[1, 2, 4], num_res_blocks=2, attn_resolutions=[24],
              in_channels=3, resolution=336, dropout=0.0, double_z=False, initial_resolution=24)
    torch.manual_seed(0)
    m = _build(dd, 512).to(torch.bfloat16).eval().cuda()
    B = 8
    idx = torch.randint(0, 512, (B, 24, 24, 2)).cuda()
    img = m.decode_code(idx)
    assert img.shape == (B, 3, 336, 336), img.shape
    dt = timeit(lambda: m.decode_code(idx), 5)
    nparam = sum(p.numel() for n, p in m.named_parameters() if n.startswith(("decoder.", "post_quant_conv.")))
    print(json.dumps({"row": "f2 VQ decode", "batch": B, "config": {k: dd[k] for k in ("z_channels", "ch", "ch_mult", "num_res_blocks", "attn_resolutions", "resolution")},
                      "decoder_params_M": round(nparam / 1e6, 1), "images_per_s": round(B / dt, 2), "ms_per_batch": round(dt * 1e3, 2)}), flush=True)


if __name__ == "__main__":
    for fn in (f3, f2):
        try:
            fn()
        except Exception as e:
            print(json.dumps({"row": fn.__name__, "error": repr(e)[:300]}), flush=True)
