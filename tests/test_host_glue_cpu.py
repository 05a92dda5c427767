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
