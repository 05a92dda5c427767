"""Golden fixture for the KV-cache decode path from the REFERENCE's own LibraForCausalLM:
prefill with use_cache=True, then one token per call with past_key_values / position_ids / past_hidden_states /
past_vision_flag, exactly as prepare_inputs_for_generation + _update_model_kwargs_for_generation drive it
(modeling_libra.py:1190-1282).  Two sequences:
  A. BOS | image | text prompt  -> text continuation (teacher-forced ids);
  B. BOS | text prompt          -> an image, token by token (BOI, codes, EOI): decoded vision tokens carry NO encoder
     signal (contiguous_signal=None -> zeros, :1216-1218 and :646-653).
The model and its weights are libra_tiny's (same seeds; checked equal), so only inputs and per-step logits are stored.
Build-container only (imports /root/reference through ref_harness)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_harness as rh  # noqa: E402
import make_golden_libra as mg  # noqa: E402


def main():
    from helpers import load_golden, sub
    from make_golden import _save
    cfgm, ml, ll = rh.libra_modules()
    cfg = cfgm.LibraConfig(**mg.TINY)
    torch.manual_seed(0)
    model = ml.LibraForCausalLM(cfg).eval()
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim == 1:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g) if "norm" in n else 0.02 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / p.shape[-1] ** 0.5))
    ids, am, vi, sig, spans, boi, eoi = mg.build_inputs(mg.TINY, g)
    ref_w = sub(load_golden("libra_tiny.safetensors")[0], "w.")
    sdm = model.state_dict()
    for k, v in ref_w.items():
        assert torch.equal(sdm[k], v), f"weights drifted from libra_tiny.safetensors: {k}"

    L = mg.TINY["max_vision_token_length"]
    cases = {
        # name: (batch row, sequence length used, prefill length)
        "A": (0, 16, 10),        # BOS + 6 image tokens + 3 text -> continue with 6 text tokens
        "B": (1, 10, 4),         # BOS + 3 text -> BOI, 4 codes, EOI
    }
    t = {}
    meta = dict(cfg=mg.TINY, cases={}, newline_token_id=int(model.newline_token_id))
    with torch.no_grad():
        for name, (row, S, P) in cases.items():
            b = slice(row, row + 1)
            cids, cvi = ids[:, b, :S].clone(), vi[b, :S].clone()
            csig = sig[b, :S].clone()
            csig[:, P:] = 0                                   # decoded positions never have an encoder signal
            cam = torch.ones(1, S, dtype=torch.long)
            out = model(input_ids=cids[:, :, :P], attention_mask=cam[:, :P], vision_indices=cvi[:, :P],
                        contiguous_signal=csig[:, :P], use_cache=True)
            steps = [out.logits]                              # [Q, 1, P, V + Vv]
            past, ph, pf = out.past_key_values, out.past_hidden_states, out.past_vision_flag
            for s in range(P, S):
                o = model(input_ids=cids[:, :, s:s + 1], attention_mask=cam[:, :s + 1], vision_indices=cvi[:, s:s + 1],
                          contiguous_signal=None, past_key_values=past, use_cache=True, position_ids=torch.tensor([[s]]),
                          past_hidden_states=ph, past_vision_flag=pf)
                past, ph, pf = o.past_key_values, o.past_hidden_states, o.past_vision_flag
                steps.append(o.logits)                        # [Q, 1, 1, V + Vv]
            inc = torch.cat(steps, dim=2)
            # the reference's own full forward on the same inputs (signal zeroed at the decoded positions)
            full = model(input_ids=cids, attention_mask=cam, vision_indices=cvi, contiguous_signal=csig, use_cache=False)
            # every cached step must reproduce the full forward, except a step whose input token is EOI: there the cached
            # path overwrites the logits with the "append a newline" placeholder (:1141-1144) - recorded as it is
            forced = torch.zeros(S, dtype=torch.bool)
            forced[P:] = cvi[0, P:] == L - 1
            keep = ~forced
            fi, ff = torch.isfinite(inc[:, :, keep]), torch.isfinite(full.logits[:, :, keep])
            assert torch.equal(fi, ff)
            d = inc[:, :, keep] - full.logits[:, :, keep]
            gap = float(d[torch.isfinite(d)].abs().max())
            assert gap < 1e-4, gap
            for s_ in forced.nonzero().flatten().tolist():
                row_ = inc[:, 0, s_]
                assert torch.isposinf(row_[:, model.newline_token_id]).all() and torch.isneginf(row_).sum() == row_.numel() - row_.shape[0]
            kc = past[0]                                      # layer 0 cache: ([K_for_vision, K_for_language], V, V_bridge, flag)
            t.update({f"{name}.input_ids": cids, f"{name}.vision_indices": cvi, f"{name}.signal": csig,
                      f"{name}.logits_incremental": inc, f"{name}.cache0.k_for_vision": kc[0][0],
                      f"{name}.cache0.k_for_language": kc[0][1], f"{name}.cache0.v": kc[1], f"{name}.cache0.v_bridge": kc[2],
                      f"{name}.cache0.flag": kc[3].to(torch.uint8)})
            meta["cases"][name] = dict(prefill=P, length=S, incremental_vs_full_forward_max_abs=gap,
                                       eoi_forced_steps=forced.nonzero().flatten().tolist())
            print(name, "incremental vs full:", gap)
    # generation glue (pure tensor logic): the reference's own prepare_inputs_for_generation /
    # _update_model_kwargs_for_generation on a toy state - mid-image, at EOI and in text
    import types
    vi0 = torch.tensor([[L, 0, 1, 2], [L, L, L, L], [L, 3, 4, 5]])
    am0 = torch.ones(3, 4, dtype=torch.long)
    ids0 = torch.arange(2 * 3 * 4).reshape(2, 3, 4)
    fake_out = types.SimpleNamespace(past_hidden_states=None, past_vision_flag=None, past_key_values=("cache",))
    model._extract_past_from_model_output = lambda outputs, standardize_cache_format=False: outputs.past_key_values
    kw = model._update_model_kwargs_for_generation(fake_out, {"attention_mask": am0.clone(), "vision_indices": vi0.clone()})
    kw2 = model._update_model_kwargs_for_generation(fake_out, {"attention_mask": kw["attention_mask"], "vision_indices": kw["vision_indices"]})
    prep = model.prepare_inputs_for_generation(torch.cat([ids0, ids0[:, :, -1:] + 1], -1), past_key_values=("cache",),
                                               attention_mask=kw["attention_mask"], vision_indices=kw["vision_indices"],
                                               contiguous_signal=torch.zeros(3, 5, 4), use_cache=True)
    assert prep["contiguous_signal"] is None
    prep0 = model.prepare_inputs_for_generation(ids0, past_key_values=None, attention_mask=am0, vision_indices=vi0, use_cache=True)
    t.update({"glue.vision_indices0": vi0, "glue.attention_mask0": am0, "glue.input_ids0": ids0,
              "glue.vision_indices1": kw["vision_indices"], "glue.attention_mask1": kw["attention_mask"],
              "glue.vision_indices2": kw2["vision_indices"],
              "glue.prep.input_ids": prep["input_ids"], "glue.prep.position_ids": prep["position_ids"],
              "glue.prep.vision_indices": prep["vision_indices"], "glue.prep0.position_ids": prep0["position_ids"],
              "glue.prep0.input_ids": prep0["input_ids"]})
    _save("libra_tiny_decode.safetensors", t, meta)


if __name__ == "__main__":
    main()
