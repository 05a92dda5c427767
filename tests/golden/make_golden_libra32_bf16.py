"""The REFERENCE's own LibraForCausalLM executed in bf16 (`model.to(torch.bfloat16)`, as train.py:31-32 / the yaml's
torch_dtype do) on the tiny-width x 32-layer case of make_golden_libra32.py: same seeded weights, same inputs.  This pins the
yardstick "theirs" of the model-level parity gates (ours <= max(k x theirs, floor)) to the reference's arithmetic instead of to the
oracle run in bf16.  Build-container only (imports /root/reference through ref_harness).

Stored: every hidden state, the logits and the loss of the bf16 run (bf16 values), plus the reference's own bf16 autograd
gradients of the small parameters (norm weights, rank-8 bridge matrices) and of layers 0 / 15 / 31."""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402
from make_golden_libra import build_inputs  # noqa: E402
from make_golden_libra32 import DEEP, SEED, keep_grad  # noqa: E402
from seeded_weights import checksum, seeded_state  # noqa: E402


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
    model = model.to(torch.bfloat16)
    g = torch.Generator().manual_seed(11)
    ids, am, vi, sig, spans, boi, eoi = build_inputs(DEEP, g)
    sig = sig.to(torch.bfloat16)
    fake = types.SimpleNamespace(tokenizer=types.SimpleNamespace(
        image_tokenizer=types.SimpleNamespace(boi_token_id=boi), text_tokenizer=types.SimpleNamespace(bos_token_id=1)))
    labels = ml.LibraTrainWrapper.get_labels(fake, {"input_ids": ids, "attention_mask": am}, spans)
    out = model(input_ids=ids, attention_mask=am, vision_indices=vi, contiguous_signal=sig, labels=labels,
                use_cache=False, output_hidden_states=True)
    assert torch.isfinite(out.loss), out.loss
    assert len(out.hidden_states) == 33 and out.hidden_states[-1].dtype == torch.bfloat16
    out.loss.backward()
    t = {"out.logits": out.logits.detach(), "out.loss": out.loss.detach().float().reshape(1),
         "out.hidden_states": torch.stack([h.detach() for h in out.hidden_states])}
    kept = 0
    for n, p in model.named_parameters():
        if p.grad is not None and keep_grad(n, p.numel()):
            t["grad." + n] = p.grad.detach()
            kept += 1
    _save("libra_tiny_depth32_bf16.safetensors", t,
          dict(cfg=DEEP, seed=SEED, checksum=checksum(sd), dtype="bfloat16", n_grads=kept,
               note="same weights / inputs as libra_tiny_depth32.safetensors; the reference run under model.to(torch.bfloat16)"))


if __name__ == "__main__":
    main()
