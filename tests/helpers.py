"""Shared test helpers (fixture loading, tolerances)."""
import json
import os

import torch
from safetensors import safe_open

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    path = os.path.join(GOLDEN, name)
    t = {}
    with safe_open(path, framework="pt") as f:
        meta = json.loads(f.metadata()["meta"])
        for k in f.keys():
            t[k] = f.get_tensor(k)
    return t, meta


def sub(t, prefix):
    return {k[len(prefix):]: v for k, v in t.items() if k.startswith(prefix)}


def is_ref_buffer(name: str) -> bool:
    """The reference state dict's persistent BUFFERS (modeling_libra.py:870-882, modeling_llama.py:139): carried by the fixtures'
    `w.` tensors since round 6 (the unfiltered key set), but not parameters - no gradient, no freeze policy, -inf values."""
    return name.endswith(("naive_placeholder", "logits_placeholder", "rotary_emb.inv_freq"))


def sub_params(t, prefix):
    """`sub` without the reference's persistent buffers: the PARAMETERS of the fixture's model."""
    return {k: v for k, v in sub(t, prefix).items() if not is_ref_buffer(k)}


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max|a-b| / max|b|  — the max-norm relative error SURVEY §8(d) gates on."""
    a = a.double()
    b = b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def word_level_tokenizer(words, model_max_length=64):
    """A real HF `PreTrainedTokenizerFast` (word-level model built in memory) standing in for the LLaMA sentencepiece
    tokenizer, which does not exist offline: ids 0..2 = <unk>, <s>, </s> like LLaMA; BOS is prepended like
    LlamaTokenizerFast's post-processor does."""
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2}
    for w in words:
        vocab.setdefault(w, len(vocab))
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tk.post_processor = processors.TemplateProcessing(single="<s> $A", special_tokens=[("<s>", 1)])
    return PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="<unk>", bos_token="<s>", eos_token="</s>",
                                   model_max_length=model_max_length, padding_side="right")


class FakeImageTokenizer:
    """Image-tokenizer stand-in returning prebuilt ids / features (fixture-driven tests of the tensor assembly)."""

    def __init__(self, image_ids, feat, boi, L, Q, device="cpu", dtype=None):
        import torch
        self.image_ids, self.feat = image_ids.to(device), feat.to(device)
        self.boi_token_id, self.eoi_token_id, self.max_vision_token_length, self.num_codebook = boi, boi + 1, L, Q
        self.device, self.dtype = torch.device(device), dtype or feat.dtype

    def __call__(self, images):
        return {"input_ids": self.image_ids.clone(), "encoder_feat": self.feat.clone()}

    def get_token_length(self, images):
        return self.max_vision_token_length


class FakeTextTokenizer:
    """Text-tokenizer stand-in returning prebuilt ids (the fixtures' text ids were built by hand: no sentencepiece model)."""
    bos_token_id, eos_token_id, pad_token_id, unk_token = 1, 2, 0, "<unk>"

    def __init__(self, ids, am, vocab_size, ph, gen, model_max_length):
        self.ids, self.am, self.vocab_size, self.model_max_length = ids, am, vocab_size, model_max_length
        self._ph, self._gen = ph, gen

    def add_tokens(self, t):
        return 1

    def convert_tokens_to_ids(self, t):
        return {"<img_ph>": self._ph, "<img_gen>": self._gen}[t]

    def __call__(self, texts, return_tensors="pt", return_length=True, **kw):
        from transformers import BatchEncoding
        return BatchEncoding({"input_ids": self.ids.clone(), "attention_mask": self.am.clone(), "length": self.am.sum(1)})


def torch_sumsq(x, out, *, accumulate=False):
    """Plain-torch statement of `libra_sumsq_bf16` (injected into dp.FlatAdamW / GradBuckets.grad_norm_sq by the CPU tests)."""
    s = x.float().pow(2).sum()
    out.copy_(out + s if accumulate else s.reshape(1))
    return out


def torch_adamw_update(master, m, v, grad, param, *, lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2,
                       grad_scale=1.0, grad_norm_sq=None, max_grad_norm=0.0):
    """Plain-torch statement of `libra_adamw_step` (torch.optim.AdamW's arithmetic on an fp32 master): the checker for the
    HIP kernel, and the `update_fn` the CPU (gloo) tests inject into dp.FlatAdamW - the product has no CPU optimizer."""
    import math
    if grad_norm_sq is not None:                   # torch.nn.utils.clip_grad_norm_: coef = max / (norm + 1e-6), clamped to 1
        grad_scale = grad_scale * min(1.0, max_grad_norm / (math.sqrt(float(grad_norm_sq)) + 1e-6))
    g = grad.float() * grad_scale
    master.mul_(1.0 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    denom = v.sqrt() / math.sqrt(bias_corr2) + eps
    master.addcdiv_(m, denom, value=-lr / bias_corr1)
    param.copy_(master)


def parity_report(line: str):
    """Append one line to the parity report of this GPU visit (gpurun_out/parity_report.txt; copied to profiles/rNN_parity.txt
    by the builder): the measured ours / theirs numbers behind the tolerance floors, per config."""
    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_report.txt"), "a") as f:
            f.write(line.rstrip() + "\n")
    except OSError:
        pass
    print(line)
