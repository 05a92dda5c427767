"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own
modules (imported from /root/reference via ref_harness).  Run in the build
container only:   python tests/golden/make_golden.py

Fixtures are data only (seeded inputs, random-init weights under the reference's
state-dict key names, the reference's outputs and autograd gradients), stored as
small .safetensors files.  They pin the CPU oracle in oracle/ (see
tests/test_oracle_golden.py) and, transitively, the HIP path.
"""
import json
import os
import sys
import tempfile

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402


def _save(name, tensors, meta):
    tensors = {k: v.detach().contiguous().clone() for k, v in tensors.items()}
    save_file(tensors, os.path.join(HERE, name), metadata={"meta": json.dumps(meta)})
    sz = os.path.getsize(os.path.join(HERE, name)) / 1e6
    print(f"wrote {name}: {len(tensors)} tensors, {sz:.2f} MB")


# head_dim must be 64 (as in CLIP ViT-L/14) for the gfx950 attention kernel the fixtures also check
VIT_TINY = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                image_size=56, patch_size=14)


def _randomize(model, seed):
    """The reference initialisers leave LN at (1,0) and biases at 0; perturb every
    parameter so that each term of the computation is numerically live."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim == 1 and ("norm" in n):
                p.add_(torch.randn(p.shape, generator=g) * 0.1)
            elif p.ndim == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 / p[0].numel() ** 0.5) * 2)


def make_vit():
    cfgm, mc = rh.clip_modules()
    cfg = cfgm.CLIPVisionConfig(**VIT_TINY)
    torch.manual_seed(0)
    m = mc.CLIPVisionModel(cfg).eval()
    _randomize(m, 1)
    g = torch.Generator().manual_seed(42)
    x = torch.randn(2, 3, 56, 56, generator=g).requires_grad_(True)
    out = m(x, output_hidden_states=True)
    hs = out.hidden_states
    # cotangent on the two-layer channel concat the tower selects (clip_encoder.py:31-45)
    sel = torch.cat([hs[-2], hs[-3]], dim=-1)[:, 1:]
    ct = torch.randn(sel.shape, generator=g)
    (sel * ct).sum().backward()
    t = {"in.pixel_values": x.detach(), "in.cotangent": ct}
    for k, v in m.state_dict().items():
        if v.is_floating_point():
            t["w." + k] = v
    for i, h in enumerate(hs):
        t[f"out.hidden_states.{i}"] = h
    t["grad.pixel_values"] = x.grad
    for n, p in m.named_parameters():
        if p.grad is not None:
            t["grad." + n] = p.grad
    _save("vit_tiny.safetensors", t, dict(cfg=VIT_TINY, select_layer=[-2, -3], eps=cfg.layer_norm_eps,
                                          act=cfg.hidden_act))
    return m, cfg


def make_vq(embed_dim):
    cfgm, mc = rh.clip_modules()
    vq, lfq, it = rh.vq_modules()
    cfg = cfgm.CLIPVisionConfig(**VIT_TINY)
    torch.manual_seed(0)
    clip = mc.CLIPVisionModel(cfg).eval()
    _randomize(clip, 1)
    with tempfile.TemporaryDirectory(prefix="tiny_clip_") as d:
        # VQModel picks the CLIP tower iff "clip" is in encoder_name (vqgan.py:44-45)
        clip.save_pretrained(d)
        dd = dict(encoder_name=d, select_layer=[-2, -3], z_channels=32, ch=32, out_ch=3, ch_mult=[1],
                  num_res_blocks=1, attn_resolutions=[], in_channels=3, resolution=8, dropout=0.0,
                  double_z=False)
        torch.manual_seed(5)
        model = vq.VQModel(ddconfig=dd, embed_dim=embed_dim, codebook_size=512, num_codebook=2).eval()
        # from_pretrained re-inits nothing here, but make sure tower == our randomised clip
        model.encoder.vision_tower.load_state_dict(clip.state_dict())
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():
            model.quant_conv.weight.copy_(torch.randn(model.quant_conv.weight.shape, generator=g) * 0.1)
            model.quant_conv.bias.copy_(torch.randn(model.quant_conv.bias.shape, generator=g) * 0.05)
        x = torch.randn(3, 3, 56, 56, generator=g)
        quant, aux, idx, feat = model.encode(x, return_encoder_feat=True)
        assert idx.dtype == torch.int64
        # ImageTokenizer.encode glue, exercised on the reference class without its __init__
        tok = it.ImageTokenizer.__new__(it.ImageTokenizer)
        torch.nn.Module.__init__(tok)
        tok.model = model
        tok.codebook_size, tok.num_codebook, tok.offset = 512, 2, 32000
        tok.boi_token_id = 32000 + 514 - 2
        tok.eoi_token_id = 32000 + 514 - 1
        enc = tok.encode(x)
        t = {"in.pixel_values": x, "out.quant": quant, "out.aux": aux.reshape(1), "out.indices": idx - 0,
             "out.encoder_feat": feat, "tok.input_ids": enc["input_ids"],
             "tok.attention_mask": enc["attention_mask"], "tok.encoder_feat": enc["encoder_feat"]}
        # NOTE: ImageTokenizer.encode adds the offset IN PLACE on `indices` (image_tokenizer.py:82);
        # re-run encode for the pristine indices.
        _, _, idx2, _ = model.encode(x, return_encoder_feat=True)
        t["out.indices"] = idx2
        for k, v in model.state_dict().items():
            if k.startswith(("quant_conv.", "quantize.project")) and v.is_floating_point():
                t["w." + k] = v
        for k, v in clip.state_dict().items():
            if v.is_floating_point():
                t["clip." + k] = v
        _save(f"vq_tiny_E{embed_dim}.safetensors", t,
              dict(cfg=VIT_TINY, select_layer=[-2, -3], embed_dim=embed_dim, offset=32000,
                   image_size=enc["image_size"]))


if __name__ == "__main__":
    which = sys.argv[1:] or ["vit", "vq", "libra"]
    if "vit" in which:
        make_vit()
    if "vq" in which:
        make_vq(18)
        make_vq(32)
    if "libra" in which:
        from make_golden_libra import make_libra
        make_libra()
