"""Golden fixture for the VQ image decoder from the REFERENCE's own classes: ImageTokenizer.decode
(image_tokenizer.py:97-124) -> VQModel.decode_code (vqgan.py:127-130) -> LFQ.indices_to_codes
(lookup_free_quantization.py:129-158) -> post_quant_conv -> taming Decoder (diffusionmodules/model.py:474-588: conv_in, mid
ResnetBlock / AttnBlock / ResnetBlock, two resolutions with an attention level and a nearest-neighbour Upsample + conv, GroupNorm-32,
swish, conv_out).  Tiny random-init model; inputs, decoder-side weights and outputs are stored.
Build-container only (imports /root/reference through ref_harness)."""
import os
import sys
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402
from make_golden import VIT_TINY, _randomize, _save  # noqa: E402

DD = dict(select_layer=[-2, -3], z_channels=32, ch=32, out_ch=3, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[4],
          in_channels=3, resolution=8, dropout=0.0, double_z=False)


def main(embed_dim=32):
    cfgm, mc = rh.clip_modules()
    vq, lfq, it = rh.vq_modules()
    cfg = cfgm.CLIPVisionConfig(**VIT_TINY)
    torch.manual_seed(0)
    clip = mc.CLIPVisionModel(cfg).eval()
    with tempfile.TemporaryDirectory(prefix="tiny_clip_") as d:
        clip.save_pretrained(d)
        torch.manual_seed(9)
        model = vq.VQModel(ddconfig=dict(DD, encoder_name=d), embed_dim=embed_dim, codebook_size=512, num_codebook=2).eval()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith(("decoder.", "post_quant_conv.", "quantize.project_out")):
                if p.ndim == 1:
                    p.copy_((1.0 if "norm" in n and n.endswith("weight") else 0.0) + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    fan = p[0].numel()
                    p.copy_(torch.randn(p.shape, generator=g) / fan ** 0.5)
    B, Hh, Q = 2, 4, 2
    idx = torch.randint(0, 512, (B, Hh, Hh, Q), generator=g)
    with torch.no_grad():
        codes = model.quantize.indices_to_codes(idx)                      # [B, E, h, w]
        z = model.post_quant_conv(codes)
        img = model.decode_code(idx)
        # token-id entry point: [Q, B, 1 + h*w + 1] with BOI / EOI and the text-vocabulary offset
        tok = it.ImageTokenizer.__new__(it.ImageTokenizer)
        torch.nn.Module.__init__(tok)
        tok.model = model
        tok.codebook_size, tok.num_codebook, tok.offset = 512, 2, 32000
        tok.boi_token_id, tok.eoi_token_id = 32000 + 512, 32000 + 513
        ids = idx.permute(3, 0, 1, 2).reshape(Q, B, Hh * Hh) + 32000
        ids = torch.cat([torch.full((Q, B, 1), tok.boi_token_id), ids, torch.full((Q, B, 1), tok.eoi_token_id)], dim=2)
        type(tok).device = property(lambda self: torch.device("cpu"))
        img2 = tok.decode(ids.clone())
        assert torch.equal(img, img2)
    t = {"in.indices": idx, "in.token_ids": ids, "out.codes": codes, "out.z": z, "out.image": img}
    for k, v in model.state_dict().items():
        if k.startswith(("decoder.", "post_quant_conv.", "quantize.project_out")) and v.is_floating_point():
            t["w." + k] = v
    _save("vq_decode_tiny.safetensors", t, dict(dd=DD, embed_dim=embed_dim, codebook_size=512, num_codebook=2, offset=32000,
                                                boi=32000 + 512, eoi=32000 + 513))
    print("image", tuple(img.shape), float(img.abs().max()))


if __name__ == "__main__":
    main()
