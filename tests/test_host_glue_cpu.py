"""CPU tests of the host-side integer glue (a11 tensor assembly, a22 labels, a23 freeze policy) against the fixtures
produced by the reference's own LibraTokenizer.forward / get_labels, and against the reference's parameter counts."""
import torch

from helpers import load_golden


def test_assemble_inputs_matches_reference_tokenizer():
    from libra_amd.libra import assemble_inputs
    t, meta = load_golden("libra_tokenizer_assembly.safetensors")
    out = assemble_inputs(t["in.text_ids"], t["in.attention_mask"],
                          {"input_ids": t["in.image_ids"], "encoder_feat": t["in.encoder_feat"]},
                          img_ph_token_id=meta["img_ph"], img_gen_token_id=meta["img_gen"], boi_token_id=meta["boi"],
                          num_codebook=meta["Q"], max_vision_token_length=meta["L"],
                          contiguous_ignore_signs=meta["ignore"], truncation=True, max_length=meta["max_length"])
    assert torch.equal(out["input_ids"], t["out.input_ids"])
    assert torch.equal(out["attention_mask"], t["out.attention_mask"])
    assert torch.equal(out["vision_indices"], t["out.vision_indices"])
    assert torch.equal(out["coninous_signal"], t["out.signal"])


def test_get_labels_matches_reference():
    from libra_amd.libra import get_labels
    t, meta = load_golden("libra_tiny.safetensors")
    lab = get_labels({"input_ids": t["in.input_ids"], "attention_mask": t["in.attention_mask"]},
                     [[tuple(s) for s in sp] for sp in meta["spans"]], boi_token_id=meta["boi"], bos_token_id=1)
    assert torch.equal(lab, t["in.labels"])


def test_freeze_policy_counts_match_reference():
    """LibraConfig() defaults are Libra-11B: 11.007 B parameters, 4.269 B of them carry "vision" in their name and are the
    only trainable ones in pretraining (measured from the reference)."""
    from libra_amd.libra import LibraConfig, LibraForCausalLM, apply_freeze_policy
    with torch.device("meta"):
        m = LibraForCausalLM(LibraConfig())
    total = sum(p.numel() for p in m.parameters())
    assert abs(total / 1e9 - 11.007) < 0.001, total
    apply_freeze_policy(m, frozen_language=True)
    trainable = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert abs(trainable / 1e9 - 4.269) < 0.001, trainable
    assert all(("vision" in n) == p.requires_grad for n, p in m.named_parameters())


def test_generation_glue_matches_reference():
    """prepare_inputs_for_generation / _update_model_kwargs_for_generation (pure tensor logic) against outputs of the
    reference's own methods on a toy state (mid-image, at EOI, in text), stored by make_golden_libra_decode.py."""
    import types
    import torch
    from helpers import load_golden
    from libra_amd.libra.modeling_libra import LibraForCausalLM
    t, meta = load_golden("libra_tiny_decode.safetensors")
    me = types.SimpleNamespace(max_vision_token_length=meta["cfg"]["max_vision_token_length"])
    out = types.SimpleNamespace(past_hidden_states=None, past_vision_flag=None, past_key_values=("cache",))
    kw = LibraForCausalLM._update_model_kwargs_for_generation(
        me, out, {"attention_mask": t["glue.attention_mask0"].clone(), "vision_indices": t["glue.vision_indices0"].clone()})
    assert torch.equal(kw["vision_indices"], t["glue.vision_indices1"]) and torch.equal(kw["attention_mask"], t["glue.attention_mask1"])
    assert kw["past_key_values"] == ("cache",)
    kw2 = LibraForCausalLM._update_model_kwargs_for_generation(me, out, dict(kw))
    assert torch.equal(kw2["vision_indices"], t["glue.vision_indices2"])
    ids0 = t["glue.input_ids0"]
    prep = LibraForCausalLM.prepare_inputs_for_generation(
        me, torch.cat([ids0, ids0[:, :, -1:] + 1], -1), past_key_values=("cache",), attention_mask=kw["attention_mask"],
        vision_indices=kw["vision_indices"], contiguous_signal=torch.zeros(3, 5, 4), use_cache=True)
    assert prep["contiguous_signal"] is None and prep["use_cache"] is True
    for k in ("input_ids", "position_ids", "vision_indices"):
        assert torch.equal(prep[k], t[f"glue.prep.{k}"]), k
    prep0 = LibraForCausalLM.prepare_inputs_for_generation(me, ids0, past_key_values=None, attention_mask=t["glue.attention_mask0"],
                                                           vision_indices=t["glue.vision_indices0"], use_cache=True)
    assert torch.equal(prep0["position_ids"], t["glue.prep0.position_ids"]) and torch.equal(prep0["input_ids"], t["glue.prep0.input_ids"])


def test_kv_cache_container_and_reorder():
    """KVCache bookkeeping that needs no GPU: buffer shapes, beam re-ordering of every buffer (modeling_libra.py:1284-1289) and
    that captured decode graphs are dropped when the buffers they point at are replaced."""
    import torch
    from libra_amd import decoder_engine as DE
    from libra_amd.libra.modeling_libra import LibraForCausalLM
    c = DE.KVCache(layers=3, B=4, capacity=10, H=256, device="cpu")
    assert len(c.layers) == 3 and all(len(l) == 4 and l[0].shape == (4, 10, 256) for l in c.layers)
    assert c.flag.shape == (4, 10) and c.flag.dtype == torch.uint8 and c.get_seq_length() == 0
    for li, layer in enumerate(c.layers):
        for bi, buf in enumerate(layer):
            buf.copy_(torch.arange(4, dtype=torch.float32)[:, None, None].expand(4, 10, 256) + 8 * li + 32 * bi)
    c.flag.copy_(torch.arange(4, dtype=torch.uint8)[:, None].expand(4, 10))
    c.length = 7
    c.graphs[(False, True, False, False)] = ("graph", {}, {})
    beam = torch.tensor([2, 2, 0, 3])
    out = LibraForCausalLM._reorder_cache(c, beam)
    assert out is c and c.graphs == {} and c.B == 4 and c.get_seq_length() == 7
    for li, layer in enumerate(c.layers):
        for bi, buf in enumerate(layer):
            assert torch.equal(buf[:, 0, 0].float(), beam.float() + 8 * li + 32 * bi)
    assert torch.equal(c.flag[:, 0], beam.to(torch.uint8))
