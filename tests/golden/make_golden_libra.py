"""Golden fixtures for the routed decoder rows (a11-a22) from the REFERENCE's own classes.
Build-container only (imports /root/reference through ref_harness)."""
import importlib.util
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

TINY = dict(vocab_size=96, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
            max_position_embeddings=64, vision_vocab_size=18, vision_codebook_num=2, max_vision_token_length=6,
            image_feature_resolution=2, contiguous_signal_size=64, bridge_rank=8, vision_down_ratio=4,
            rms_norm_eps=1e-6, pad_token_id=0, bos_token_id=1, eos_token_id=2)


def build_inputs(cfg, g):
    V, L = cfg["vocab_size"], cfg["max_vision_token_length"]
    boi, eoi = V + 16, V + 17
    B, S, Q = 2, 16, 2

    def img():
        return torch.stack([torch.cat([torch.tensor([boi]), V + torch.randint(0, 16, (4,), generator=g), torch.tensor([eoi])])
                            for _ in range(Q)])
    ids = torch.zeros(Q, B, S, dtype=torch.long)
    am = torch.ones(B, S, dtype=torch.long)
    vi = torch.full((B, S), L, dtype=torch.long)
    # sample 0: BOS | image | 9 text tokens
    t0 = torch.randint(3, V, (9,), generator=g)
    i0 = img()
    for q in range(Q):
        ids[q, 0] = torch.cat([torch.tensor([1]), i0[q], t0])
    vi[0, 1:7] = torch.arange(6)
    # sample 1: BOS | 3 text | image | 4 text | 2 pad   (right padding, pad id 0)
    t1a, t1b = torch.randint(3, V, (3,), generator=g), torch.randint(3, V, (4,), generator=g)
    i1 = img()
    for q in range(Q):
        ids[q, 1] = torch.cat([torch.tensor([1]), t1a, i1[q], t1b, torch.zeros(2, dtype=torch.long)])
    vi[1, 4:10] = torch.arange(6)
    am[1, 14:] = 0
    sig = torch.zeros(B, S, cfg["contiguous_signal_size"])
    sig[0, 2:6] = torch.randn(4, cfg["contiguous_signal_size"], generator=g)
    sig[1, 5:9] = torch.randn(4, cfg["contiguous_signal_size"], generator=g)
    spans = [[(7, 8)], [(1, 4), (10, 11)]]          # label_mask_position_map: includes first text token after an image
    return ids, am, vi, sig, spans, boi, eoi


def make_libra():
    from make_golden import _save
    cfgm, ml, ll = rh.libra_modules()
    cfg = cfgm.LibraConfig(**TINY)
    torch.manual_seed(0)
    model = ml.LibraForCausalLM(cfg).eval()
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim == 1:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g) if "norm" in n else 0.02 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / p.shape[-1] ** 0.5))
    ids, am, vi, sig, spans, boi, eoi = build_inputs(TINY, g)
    fake = types.SimpleNamespace(tokenizer=types.SimpleNamespace(
        image_tokenizer=types.SimpleNamespace(boi_token_id=boi), text_tokenizer=types.SimpleNamespace(bos_token_id=1)))
    labels = ml.LibraTrainWrapper.get_labels(fake, {"input_ids": ids, "attention_mask": am}, spans)
    out = model(input_ids=ids, attention_mask=am, vision_indices=vi, contiguous_signal=sig, labels=labels,
                use_cache=False, output_hidden_states=True)
    assert torch.isfinite(out.loss), out.loss
    out.loss.backward()
    t = {"in.input_ids": ids, "in.attention_mask": am, "in.vision_indices": vi, "in.signal": sig, "in.labels": labels,
         "out.logits": out.logits, "out.loss": out.loss.reshape(1), "out.hidden": out.hidden_states[-1],
         "out.embeds": out.hidden_states[0], "out.layer0": out.hidden_states[1]}
    # the UNFILTERED reference state dict: parameters AND its persistent buffers (naive_placeholder, the three -inf logits
    # placeholders, modeling_libra.py:870-882; one rotary_emb.inv_freq per attention layer, modeling_llama.py:139) - the key set
    # a real checkpoint carries and `load_state_dict(strict=True)` of the product model must accept (boundary)
    for k, v in model.state_dict().items():
        t["w." + k] = v
    for n, p in model.named_parameters():
        if p.grad is not None:
            t["grad." + n] = p.grad.to(torch.float16)      # halves the fixture; compared at 1e-3
    _save("libra_tiny.safetensors", t, dict(cfg=TINY, spans=spans, boi=boi, eoi=eoi))
    make_tokenizer_assembly(TINY, g)


def make_tokenizer_assembly(cfg, g):
    """a11: run the reference's real LibraTokenizer.forward (tensor-assembly half) with stand-in text / image
    tokenizers that return prebuilt ids (no sentencepiece model exists offline)."""
    from make_golden import _save
    rh.install()
    import transformers.tokenization_utils as tu
    for nm in ("TextInput",):
        if not hasattr(tu, nm):
            setattr(tu, nm, str)
    import re as _re
    for nm, val in (("re", _re), ("AddedToken", getattr(tu, "AddedToken", object))):
        if not hasattr(tu, nm):
            setattr(tu, nm, val)
    if not hasattr(tu, "logger"):
        import logging
        tu.logger = logging.getLogger("tu")
    if "omegaconf" not in sys.modules:
        om = types.ModuleType("omegaconf"); om.OmegaConf = type("OmegaConf", (), {}); sys.modules["omegaconf"] = om
    fast = types.ModuleType("libra.models.llama.tokenization_llama_fast")
    fast.LlamaTokenizerFast = type("LlamaTokenizerFast", (), {})
    sys.modules["libra.models.llama.tokenization_llama_fast"] = fast
    spec = importlib.util.spec_from_file_location("_ref_tokenization_libra",
                                                  f"{rh.REF}/libra/models/libra/tokenization_libra.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    V, L, Q = cfg["vocab_size"], cfg["max_vision_token_length"], 2
    PH, GEN = V - 1, V - 2
    boi, eoi = V + 16, V + 17
    B, S = 3, 14
    text = torch.zeros(B, S, dtype=torch.long)
    am = torch.ones(B, S, dtype=torch.long)
    text[0] = torch.cat([torch.tensor([1]), torch.full((L,), PH), torch.randint(3, V - 2, (7,), generator=g)])
    text[1] = torch.cat([torch.tensor([1]), torch.randint(3, V - 2, (2,), generator=g), torch.full((L,), PH),
                         torch.randint(3, V - 2, (3,), generator=g), torch.zeros(2, dtype=torch.long)])
    am[1, 12:] = 0
    text[2] = torch.cat([torch.tensor([1]), torch.randint(3, V - 2, (4,), generator=g), torch.full((L,), PH),
                         torch.randint(3, V - 2, (3,), generator=g)])
    n_img = 3
    image_ids = torch.stack([torch.cat([torch.full((n_img, 1), boi), V + torch.randint(0, 16, (n_img, 4), generator=g),
                                        torch.full((n_img, 1), eoi)], 1) for _ in range(Q)])
    feat = torch.randn(n_img, 4, cfg["contiguous_signal_size"], generator=g)

    class Enc(dict):
        def to(self, device):
            return self

    class FakeText:
        img_ph_token_id, img_gen_token_id, model_max_length = PH, GEN, 12

        def __call__(self, texts, return_tensors="pt", return_length=True, **kw):
            return Enc(input_ids=text.clone(), attention_mask=am.clone(), length=am.sum(1))

    class FakeImg:
        boi_token_id, max_vision_token_length, num_codebook = boi, L, Q

        def __call__(self, images):
            return {"input_ids": image_ids.clone(), "encoder_feat": feat.clone()}

        def get_token_length(self, images):
            return L
    fake = types.SimpleNamespace(text_tokenizer=FakeText(), image_tokenizer=FakeImg(),
                                 img_indices_ph=torch.arange(0, L)[None, :], raw_output=True,
                                 device=torch.device("cpu"), dtype=torch.float32)
    samples = [{"language": "a", "vision": torch.zeros(3, 8, 8), "contiguous_ignore_sign": False},
               {"language": "b", "vision": torch.zeros(3, 8, 8), "contiguous_ignore_sign": True},
               {"language": "c", "vision": torch.zeros(3, 8, 8), "contiguous_ignore_sign": False}]
    out = mod.LibraTokenizer.forward(fake, samples, padding="longest", truncation=True, max_length=12)
    t = {"in.text_ids": text, "in.attention_mask": am, "in.image_ids": image_ids, "in.encoder_feat": feat,
         "out.input_ids": out["input_ids"], "out.attention_mask": out["attention_mask"],
         "out.vision_indices": out["vision_indices"], "out.signal": out["coninous_signal"]}
    _save("libra_tokenizer_assembly.safetensors", t,
          dict(img_ph=PH, img_gen=GEN, boi=boi, eoi=eoi, Q=Q, L=L, ignore=[False, True, False], max_length=12))


if __name__ == "__main__":
    make_libra()
