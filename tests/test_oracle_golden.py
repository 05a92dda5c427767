"""The CPU oracle (oracle/) against fixtures produced by the reference's own
modules (tests/golden/make_golden.py).  CPU only."""
import pytest
import torch

from helpers import load_golden, rel_err, sub
from oracle import vit_oracle as VO
from oracle import vq_oracle as QO


@pytest.fixture(scope="module")
def vit():
    return load_golden("vit_tiny.safetensors")


def _vit_kw(meta):
    c = meta["cfg"]
    return dict(patch=c["patch_size"], heads=c["num_attention_heads"], layers=c["num_hidden_layers"],
                eps=meta["eps"])


def test_vit_hidden_states_match_reference(vit):
    t, meta = vit
    sd = sub(t, "w.")
    hs = VO.vit_hidden_states(sd, t["in.pixel_values"], **_vit_kw(meta))
    assert len(hs) == meta["cfg"]["num_hidden_layers"] + 1
    for i, h in enumerate(hs):
        ref = t[f"out.hidden_states.{i}"]
        assert h.shape == ref.shape
        assert rel_err(h, ref) < 2e-6, (i, rel_err(h, ref))


def test_vit_backward_matches_reference_autograd(vit):
    t, meta = vit
    sd = {k: v.clone().requires_grad_(True) for k, v in sub(t, "w.").items()}
    x = t["in.pixel_values"].clone().requires_grad_(True)
    hs = VO.vit_hidden_states(sd, x, **_vit_kw(meta))
    sel = VO.feature_select(hs, meta["select_layer"], square=False)
    (sel * t["in.cotangent"]).sum().backward()
    assert rel_err(x.grad, t["grad.pixel_values"]) < 1e-5
    n = 0
    for k, g in sub(t, "grad.").items():
        if k == "pixel_values":
            continue
        assert sd[k].grad is not None, k
        if float(g.abs().max()) < 1e-5:
            # k_proj.bias: softmax is invariant to a per-query constant, the true gradient is 0
            assert float(sd[k].grad.abs().max()) < 1e-5, k
        else:
            assert rel_err(sd[k].grad, g) < 2e-5, (k, rel_err(sd[k].grad, g))
        n += 1
    assert n >= 37  # every parameter that feeds hs[-2], hs[-3] (the last layer and post_layernorm get none)


@pytest.mark.parametrize("E", [18, 32])
def test_vq_encode_matches_reference(E):
    t, meta = load_golden(f"vq_tiny_E{E}.safetensors")
    c = meta["cfg"]
    hs = VO.vit_hidden_states(sub(t, "clip."), t["in.pixel_values"], patch=c["patch_size"],
                              heads=c["num_attention_heads"], layers=c["num_hidden_layers"])
    feat = VO.feature_select(hs, meta["select_layer"])
    assert rel_err(feat, t["out.encoder_feat"]) < 2e-6
    # from the reference's own feat so the sign decisions are not perturbed by 1e-6 noise upstream
    quant, aux, idx, feat2, pre = QO.vq_encode(sub(t, "w."), t["out.encoder_feat"])
    assert idx.dtype == torch.int64 and idx.shape == t["out.indices"].shape
    bad = idx != t["out.indices"]
    assert int(bad.sum()) == 0, f"{int(bad.sum())} index mismatches, min margin {pre.abs().min()}"
    assert rel_err(quant, t["out.quant"]) < 2e-6
    assert float(aux) == float(t["out.aux"]) == 0.0
    ids, am, ef = QO.image_tokenizer_encode(idx, feat2, offset=meta["offset"])
    assert torch.equal(ids, t["tok.input_ids"])
    assert torch.equal(am, t["tok.attention_mask"])
    assert torch.equal(ef, t["tok.encoder_feat"])
    assert list(quant.shape[2:]) == meta["image_size"]
    # bits: MSB-first packing, range
    assert int(idx.min()) >= 0 and int(idx.max()) < 512


def test_vq_image_decoder_matches_reference():
    """§8f-2 groundwork: token ids -> LFQ codes -> post_quant_conv -> taming Decoder against the reference's own
    ImageTokenizer.decode / VQModel.decode_code run (codes bit-exact; z and image to fp32 round-off)."""
    from oracle import vq_decode_oracle as VD
    t, meta = load_golden("vq_decode_tiny.safetensors")
    sd = sub(t, "w.")
    dd = meta["dd"]
    idx = VD.token_ids_to_indices(t["in.token_ids"], offset=meta["offset"], boi_token_id=meta["boi"])
    assert torch.equal(idx, t["in.indices"])
    codes, z, img = VD.decode_code(sd, idx, codebook_size=meta["codebook_size"], ch_mult=dd["ch_mult"],
                                   num_res_blocks=dd["num_res_blocks"], resolution=dd["resolution"])
    assert rel_err(codes, t["out.codes"]) < 1e-6
    assert rel_err(z, t["out.z"]) < 2e-6
    assert img.shape == t["out.image"].shape
    assert rel_err(img, t["out.image"]) < 2e-5
