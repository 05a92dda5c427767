"""Tiny width x FULL depth (32 decoder layers) golden fixture from the REFERENCE's own LibraForCausalLM:
the 32-iteration layer loop (modeling_libra.py:781-807) with every hidden state, the logits, the loss and the reference's own
autograd gradients.  Build-container only (imports /root/reference through ref_harness).

The 35 M weights are not stored: both sides derive them from tests/golden/seeded_weights.py (per-parameter seeds, bf16-rounded
values) and the fixture carries their checksum."""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402
from make_golden_libra import TINY, build_inputs  # noqa: E402
from seeded_weights import checksum, seeded_state  # noqa: E402

DEEP = dict(TINY, num_hidden_layers=32)
SEED = 32
# gradients kept: every parameter of at most 4096 elements (all norm weights, all rank-8 bridge matrices) in every layer, and
# every parameter of these layers + the embeddings / heads / final norms
FULL_LAYERS = (0, 15, 31)


def keep_grad(name: str, numel: int) -> bool:
    if numel <= 4096 or not name.startswith("model.layers."):
        return True
    return int(name.split(".")[2]) in FULL_LAYERS


def main():
    from make_golden import _save
    cfgm, ml, ll = rh.libra_modules()
    cfg = cfgm.LibraConfig(**DEEP)
    torch.manual_seed(0)
    model = ml.LibraForCausalLM(cfg).eval()
    sd = seeded_state([(n, p.shape) for n, p in model.named_parameters()], SEED)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(sd[n])
    g = torch.Generator().manual_seed(11)
    ids, am, vi, sig, spans, boi, eoi = build_inputs(DEEP, g)
    sig = sig.to(torch.bfloat16).float()
    fake = types.SimpleNamespace(tokenizer=types.SimpleNamespace(
        image_tokenizer=types.SimpleNamespace(boi_token_id=boi), text_tokenizer=types.SimpleNamespace(bos_token_id=1)))
    labels = ml.LibraTrainWrapper.get_labels(fake, {"input_ids": ids, "attention_mask": am}, spans)
    out = model(input_ids=ids, attention_mask=am, vision_indices=vi, contiguous_signal=sig, labels=labels,
                use_cache=False, output_hidden_states=True)
    assert torch.isfinite(out.loss), out.loss
    assert len(out.hidden_states) == 33
    out.loss.backward()
    t = {"in.input_ids": ids, "in.attention_mask": am, "in.vision_indices": vi, "in.signal": sig, "in.labels": labels,
         "out.logits": out.logits, "out.loss": out.loss.reshape(1),
         "out.hidden_states": torch.stack([h.detach() for h in out.hidden_states])}       # [33, B, S, H]: embeddings, layers 0..30, normed layer 31
    kept = 0
    for n, p in model.named_parameters():
        if p.grad is not None and keep_grad(n, p.numel()):
            t["grad." + n] = p.grad.to(torch.float32 if p.numel() <= 4096 else torch.float16)
            kept += 1
    names = [n for n, _ in model.named_parameters()]
    _save("libra_tiny_depth32.safetensors", t,
          dict(cfg=DEEP, spans=spans, boi=boi, eoi=eoi, seed=SEED, checksum=checksum(sd), n_params=len(names), n_grads=kept,
               full_layers=list(FULL_LAYERS)))


if __name__ == "__main__":
    main()
