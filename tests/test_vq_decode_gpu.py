"""VQ image decoder on device: row kernels against torch statements with the reference's bf16 rounding points,
the whole decode path against the fixture made by the reference's own ImageTokenizer.decode -> VQModel.decode_code -> taming
Decoder (tests/golden/make_golden_vq_decode.py) and against the CPU oracle on a deeper three-level configuration."""
import pytest
import torch
import torch.nn.functional as F

from helpers import load_golden, parity_report, rel_err, sub

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _tiny_clip():
    from transformers import CLIPVisionConfig
    from libra_amd.clip import CLIPVisionModel
    return CLIPVisionModel(CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                                            image_size=56, patch_size=14))


def test_lfq_codes_and_softmax_rows():
    from libra_amd import kernels as K
    g = torch.Generator().manual_seed(0)
    idx = torch.randint(0, 512, (300, 2), generator=g)
    codes = K.lfq_codes(idx.cuda(), 9, 64).cpu().float()
    bits = ((idx[..., None] >> torch.arange(8, -1, -1)) & 1).float() * 2 - 1
    assert torch.equal(codes[:, :18], bits.reshape(300, 18)) and float(codes[:, 18:].abs().max()) == 0.0
    x = (torch.randn(70, 128, generator=g) * 3).to(BF)
    got = K.softmax_rows_(x.clone().cuda(), 100, 0.125).cpu().float()
    want = torch.softmax((x[:, :100].float() * 0.125).to(BF).float(), dim=-1).to(BF).float()
    assert float((got[:, :100] - want).abs().max()) <= 2 ** -8 * float(want.max()) and float(got[:, 100:].abs().max()) == 0.0


@pytest.mark.parametrize("C,H,up", [(32, 4, 1.0), (64, 6, 2.0), (128, 5, 1.6), (512, 3, 1.0)])
def test_groupnorm_and_conv_gather(C, H, up):
    """GroupNorm(32) statistics + the fused normalise / swish / nearest-upsample / 3x3 gather against torch ops in the reference's
    order (GroupNorm -> bf16, swish -> bf16, F.interpolate nearest, unfold)."""
    from libra_amd import kernels as K
    g = torch.Generator().manual_seed(C)
    B, W = 2, H + 1
    x = (torch.randn(B, C, H, W, generator=g) * 2 + 0.5).to(BF)
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).to(BF), (0.2 * torch.randn(C, generator=g)).to(BF)
    xn = x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().cuda()
    sc, sh = K.groupnorm_affine(xn, gamma.cuda(), beta.cuda(), B, H * W, 32, 1e-6)
    y = F.group_norm(x.float(), 32, gamma.float(), beta.float(), eps=1e-6)
    y_aff = x.float() * sc.cpu()[:, :, None, None] + sh.cpu()[:, :, None, None]
    assert rel_err(y_aff, y) < 2e-5
    yb = y.to(BF)
    sw = (yb.float() * torch.sigmoid(yb.float()).to(BF).float()).to(BF)
    ups = F.interpolate(sw.float(), scale_factor=up, mode="nearest") if up != 1.0 else sw.float()
    Ho, Wo = ups.shape[-2:]
    cols = F.unfold(ups, 3, padding=1).view(B, C, 9, Ho * Wo).permute(0, 3, 2, 1).reshape(B * Ho * Wo, 9 * C)   # tap-major, then c
    kp = K.round_up(9 * C, 64)
    got = K.conv_gather(xn, B, H, W, Ho, Wo, 3, kp, scale=sc, shift=sh, swish=True, inv_scale=1.0 / up).cpu().float()
    assert float(got[:, 9 * C:].abs().max()) == 0.0 if kp > 9 * C else True
    d = (got[:, :9 * C] - cols).abs()
    assert float(d.max()) <= 2 ** -7 * float(cols.abs().max()) + 1e-6, float(d.max())      # one bf16 step (fp32 stats order)
    assert float((d > 0).float().mean()) < 0.02


def _build(meta_dd, embed_dim):
    from libra_amd.libra.vqgan import VQModel
    dd = dict(meta_dd); dd["encoder_name"] = "clip_tiny"
    return VQModel(ddconfig=dd, embed_dim=embed_dim, codebook_size=512, num_codebook=2, vision_model=_tiny_clip())


def test_vq_decode_tiny_vs_reference_fixture():
    from libra_amd.libra.image_tokenizer import ImageTokenizer
    from oracle import vq_decode_oracle as DO
    t, meta = load_golden("vq_decode_tiny.safetensors")
    m = _build(meta["dd"], meta["embed_dim"])
    missing, unexpected = m.load_state_dict(sub(t, "w."), strict=False)
    assert not unexpected and not [k for k in missing if k.startswith(("decoder.", "post_quant_conv.", "quantize.project_out"))]
    m = m.to(BF).cuda().eval()
    idx = t["in.indices"].cuda()
    codes = m.quantize.indices_to_codes(idx)
    assert codes.shape == t["out.codes"].shape and rel_err(codes.float().cpu(), t["out.codes"]) < 6e-3
    img = m.decode_code(idx)
    assert img.shape == t["out.image"].shape and img.dtype == BF
    # yardstick: the oracle run op by op in bf16 (the reference's own bf16 arithmetic) against the fp32 fixture
    wb = {k: v.to(BF) for k, v in sub(t, "w.").items()}
    dd = meta["dd"]
    _, _, theirs_img = DO.decode_code(wb, t["in.indices"], codebook_size=512, ch_mult=dd["ch_mult"],
                                      num_res_blocks=dd["num_res_blocks"], resolution=dd["resolution"])
    ours, theirs = rel_err(img.float().cpu(), t["out.image"]), rel_err(theirs_img.float(), t["out.image"])
    parity_report(f"[f2 VQ decode, tiny fixture] image vs the reference's fp32 run: ours {ours:.3e} theirs(bf16 op-by-op) {theirs:.3e}")
    assert ours < max(2 * theirs, 1e-2), (ours, theirs)
    # token-id entry point (image_tokenizer.py:97-124): BOI / EOI stripped, offset removed, same image
    tok = ImageTokenizer.__new__(ImageTokenizer)
    torch.nn.Module.__init__(tok)
    tok.model, tok.offset, tok.boi_token_id, tok.eoi_token_id = m, meta["offset"], meta["boi"], meta["eoi"]
    assert torch.equal(tok.decode(t["in.token_ids"].cuda()), img)
    with pytest.raises(ValueError, match="square"):
        tok.decode(t["in.token_ids"].cuda()[:, :, :-3])


def test_vq_decode_three_levels_vs_oracle():
    """ch 64, ch_mult (1, 2, 2), attention at the 8x8 level, 8 -> 16 -> 32 px (a x2 level and the final jump to `resolution`),
    E = 512 with project_out: against the CPU oracle in fp32, yardstick = the oracle in bf16."""
    from oracle import vq_decode_oracle as DO
    dd = dict(select_layer=[-2, -3], z_channels=64, ch=64, out_ch=3, ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[8],
              in_channels=3, resolution=32, dropout=0.0, double_z=False)
    torch.manual_seed(5)
    m = _build(dd, 512)
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.startswith(("decoder.", "post_quant_conv.", "quantize.project_out")):
                if p.ndim == 1:
                    p.copy_((1.0 if "norm" in n and n.endswith("weight") else 0.0) + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(torch.randn(p.shape, generator=g) / p[0].numel() ** 0.5)
    m = m.to(BF).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items() if k.startswith(("decoder.", "post_quant_conv.", "quantize.project_out"))}
    m = m.cuda()
    idx = torch.randint(0, 512, (2, 8, 8, 2), generator=g)
    img = m.decode_code(idx.cuda())
    kw = dict(codebook_size=512, ch_mult=dd["ch_mult"], num_res_blocks=1, resolution=32)
    _, _, ref = DO.decode_code({k: v.float() for k, v in sd.items()}, idx, **kw)
    _, _, refb = DO.decode_code(sd, idx, **kw)
    assert img.shape == ref.shape == (2, 3, 32, 32)
    ours, theirs = rel_err(img.float().cpu(), ref), rel_err(refb.float(), ref)
    parity_report(f"[f2 VQ decode, ch 64 x (1,2,2), 8->32 px, E=512] image vs fp32 oracle: ours {ours:.3e} theirs(bf16 op-by-op) {theirs:.3e}")
    assert ours < max(2 * theirs, 1e-2), (ours, theirs)
    q = m.quantize.indices_to_codes(idx.cuda())
    assert torch.equal(m.decode(q), img)
