"""Golden fixtures for the alternative decoder configurations from the REFERENCE's own LibraForCausalLM:
`use_2d_rope` (modeling_libra.py:43-49, :576-587, :663-678), `unified_head` (:1054-1064), `vision_prediction_mode="2d"`
(:942-1014), `use_bridge=False` (:258, :311-317, :394) and the embedding-stage switches `use_vision_position_embedding` (:564-566,
:636-638), `norm_signals=False` (:558, :641-644), `concat_signals=False` (:561-562, :753-754), and `addition_mode` (cal_language_vision :111-127 on the q / k / v / o projections), each on libra_tiny's inputs and weights (the 2d-prediction heads have their own [18, 2*256] weights, stored).
Stored per variant: loss, final hidden state, logits, and the reference autograd's gradients of a sample of parameters.
Build-container only (imports /root/reference through ref_harness)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_harness as rh  # noqa: E402
import make_golden_libra as mg  # noqa: E402

SAMPLE = ["model.layers.0.self_attn.q_proj.weight", "model.layers.1.self_attn.vision_k_proj.weight_A",
          "model.layers.0.self_attn.vision_k_bridge_on_language.weight_B", "model.layers.1.mlp.vision_down_proj.weight_B",
          "model.embed_tokens.weight", "model.vision_embed_tokens.1.weight", "model.norm.weight", "model.vision_norm.weight",
          "lm_head.weight", "vision_lm_head.heads.0.weight", "vision_lm_head.heads.1.weight", "vision_hidden_placeholder"]


def main():
    from helpers import load_golden, sub
    from make_golden import _save
    cfgm, ml, ll = rh.libra_modules()
    t0, meta0 = load_golden("libra_tiny.safetensors")
    w = sub(t0, "w.")
    ids, am, vi, sig, labels = t0["in.input_ids"], t0["in.attention_mask"], t0["in.vision_indices"], t0["in.signal"], t0["in.labels"]
    out_t, meta = {}, dict(cfg=mg.TINY, variants={})
    for name, over in (("rope2d", dict(use_2d_rope=True)), ("unified", dict(unified_head=True)),
                       ("pred2d", dict(vision_prediction_mode="2d")), ("rope2d_pred2d", dict(use_2d_rope=True, vision_prediction_mode="2d")),
                       ("nobridge", dict(use_bridge=False)), ("vispos", dict(use_vision_position_embedding=True)),
                       ("nonorm", dict(norm_signals=False)), ("noconcat", dict(concat_signals=False)),
                       ("addition", dict(addition_mode=True))):
        cfg = cfgm.LibraConfig(**dict(mg.TINY, **over))
        torch.manual_seed(0)
        model = ml.LibraForCausalLM(cfg).eval()
        sd = dict(w)
        if over.get("use_bridge") is False:                               # :258 - the layer has no bridge parameters at all
            sd = {k: v for k, v in sd.items() if "_bridge_on_" not in k}
        extra = {}
        gx = torch.Generator().manual_seed(23)
        H_, Cs_, L_ = mg.TINY["hidden_size"], mg.TINY["contiguous_signal_size"], mg.TINY["max_vision_token_length"]
        if over.get("use_vision_position_embedding"):                     # :564-566 Embedding(max_vision_token_length, hidden)
            extra["model.vision_position_embedding.weight"] = torch.randn(L_, H_, generator=gx) * 0.5
        if over.get("norm_signals") is False:                             # :558 - no vision_signal_norm module
            sd.pop("model.vision_signal_norm.weight")
        if over.get("concat_signals") is False:                           # :561-562 Linear(contiguous_signal_size -> hidden), no norm
            sd.pop("model.vision_signal_norm.weight")
            extra["model.vision_contiguous_signal_processor.weight"] = torch.randn(H_, Cs_, generator=gx) * Cs_ ** -0.5
        sd.update(extra)
        if over.get("vision_prediction_mode") == "2d":                    # heads take cat(up, left): [Vv, 2 * hidden]
            g = torch.Generator().manual_seed(17)
            for q in range(2):
                sd[f"vision_lm_head.heads.{q}.weight"] = torch.randn(mg.TINY["vision_vocab_size"], 2 * mg.TINY["hidden_size"], generator=g) * (2 * mg.TINY["hidden_size"]) ** -0.5
            sd["vision_hidden_placeholder"] = torch.randn(mg.TINY["hidden_size"], generator=g) * 0.5
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        out = model(input_ids=ids, attention_mask=am, vision_indices=vi, contiguous_signal=sig, labels=labels, use_cache=False,
                    output_hidden_states=True)
        assert torch.isfinite(out.loss), (name, out.loss)
        out.loss.backward()
        out_t[f"{name}.loss"] = out.loss.detach().reshape(1)
        out_t[f"{name}.hidden"] = out.hidden_states[-1].detach()
        out_t[f"{name}.logits"] = out.logits.detach()
        grads = dict(model.named_parameters())
        more = ["model.layers.1.self_attn.o_proj.weight", "model.layers.0.self_attn.vision_o_proj.weight_A",
                "model.layers.0.self_attn.v_proj.weight", "model.layers.1.self_attn.vision_q_proj.weight_B",
                "model.layers.0.self_attn.vision_k_bridge_on_language.weight_A",
                "model.layers.0.input_layernorm.weight", "model.layers.1.vision_input_layernorm.weight"] if name == "addition" else []
        for n in SAMPLE + more:
            if n in grads and grads[n].grad is not None:
                out_t[f"{name}.grad.{n}"] = grads[n].grad.detach().clone()
        if over.get("vision_prediction_mode") == "2d":
            for q in range(2):
                out_t[f"{name}.w.vision_lm_head.heads.{q}.weight"] = sd[f"vision_lm_head.heads.{q}.weight"]
            out_t[f"{name}.w.vision_hidden_placeholder"] = sd["vision_hidden_placeholder"]
        for k_, v_ in extra.items():
            out_t[f"{name}.w.{k_}"] = v_
            if grads[k_].grad is not None:
                out_t[f"{name}.grad.{k_}"] = grads[k_].grad.detach().clone()
        if over.get("use_2d_rope"):
            out_t[f"{name}.position_ids"] = model.model.get_2d_position_ids(vi)
        meta["variants"][name] = over
        print(name, "loss", float(out.loss))
    _save("libra_tiny_f4.safetensors", out_t, meta)


if __name__ == "__main__":
    main()
