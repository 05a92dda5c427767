"""The decoder oracle (oracle/libra_oracle.py) against fixtures produced by the reference's own
LibraForCausalLM / LibraTrainWrapper.get_labels / LibraTokenizer.forward.  CPU only."""
import torch

from helpers import load_golden, rel_err, sub, sub_params
from oracle import libra_oracle as LO


def _run(t, meta, sd, dtype=torch.float32):
    c = meta["cfg"]
    hid, flag = LO.model_forward(sd, t["in.input_ids"], t["in.attention_mask"], t["in.vision_indices"],
                                 t["in.signal"].to(dtype), layers=c["num_hidden_layers"], heads=c["num_attention_heads"],
                                 vocab=c["vocab_size"], max_vision_token_length=c["max_vision_token_length"],
                                 eps=c["rms_norm_eps"], max_pos=c["max_position_embeddings"])
    logits = LO.vl_logits(sd, hid, flag, c["vision_codebook_num"])
    return hid, flag, logits


def test_libra_forward_matches_reference():
    t, meta = load_golden("libra_tiny.safetensors")
    sd = sub(t, "w.")
    hid, flag, logits = _run(t, meta, sd)
    assert rel_err(hid, t["out.hidden"]) < 5e-6
    ref = t["out.logits"]
    assert logits.shape == ref.shape
    fin = torch.isfinite(ref)
    assert torch.equal(fin, torch.isfinite(logits))            # the -inf padding pattern is identical
    assert rel_err(logits[fin], ref[fin]) < 5e-6
    loss = LO.causal_lm_loss(logits, t["in.labels"])
    assert abs(float(loss) - float(t["out.loss"])) < 1e-5 * abs(float(t["out.loss"]))


def test_libra_embeddings_and_first_layer():
    t, meta = load_golden("libra_tiny.safetensors")
    sd = sub(t, "w.")
    c = meta["cfg"]
    flag = t["in.vision_indices"] < c["max_vision_token_length"]
    emb = LO.input_embeds(sd, t["in.input_ids"], flag, t["in.signal"], c["vocab_size"], c["rms_norm_eps"])
    assert rel_err(emb, t["out.embeds"]) < 2e-6
    S = emb.shape[1]
    cos, sin = LO.rope_tables(c["hidden_size"] // c["num_attention_heads"], c["max_position_embeddings"])
    pos = torch.arange(S).unsqueeze(0).expand(emb.shape[0], S)
    mask = LO.additive_mask(t["in.attention_mask"], S, emb.dtype)
    x1 = LO.decoder_layer(sd, 0, emb, flag, mask, pos, c["num_attention_heads"], c["rms_norm_eps"], cos, sin)
    assert rel_err(x1, t["out.layer0"]) < 5e-6


def test_libra_backward_matches_reference_autograd():
    t, meta = load_golden("libra_tiny.safetensors")
    sd = {k: v.clone().requires_grad_(True) for k, v in sub_params(t, "w.").items()}
    hid, flag, logits = _run(t, meta, sd)
    LO.causal_lm_loss(logits, t["in.labels"]).backward()
    n = 0
    for k, g in sub(t, "grad.").items():
        if k == "vision_hidden_placeholder":
            continue
        assert sd[k].grad is not None, k
        gmax = float(g.float().abs().max())
        if gmax < 1e-6:
            assert float(sd[k].grad.abs().max()) < 1e-5, k
        else:
            assert rel_err(sd[k].grad, g.float()) < 2e-3, (k, rel_err(sd[k].grad, g.float()))    # fixture grads are fp16
        n += 1
    assert n > 60


def test_get_labels_matches_reference():
    t, meta = load_golden("libra_tiny.safetensors")
    lab = LO.get_labels(t["in.input_ids"], t["in.attention_mask"], [[tuple(s) for s in sp] for sp in meta["spans"]],
                        boi_token_id=meta["boi"], bos_token_id=1)
    assert torch.equal(lab, t["in.labels"])


def test_tokenizer_tensor_assembly_matches_reference():
    t, meta = load_golden("libra_tokenizer_assembly.safetensors")
    ids, am, vi, sig = LO.assemble_inputs(t["in.text_ids"], t["in.attention_mask"], t["in.image_ids"], t["in.encoder_feat"],
                                          img_ph_token_id=meta["img_ph"], img_gen_token_id=meta["img_gen"],
                                          boi_token_id=meta["boi"], num_codebook=meta["Q"],
                                          max_vision_token_length=meta["L"], contiguous_ignore_signs=meta["ignore"],
                                          max_length=meta["max_length"])
    assert torch.equal(ids, t["out.input_ids"])
    assert torch.equal(am, t["out.attention_mask"])
    assert torch.equal(vi, t["out.vision_indices"])
    assert torch.equal(sig, t["out.signal"])


def test_kv_cache_decode_matches_reference():
    """§8f-1 groundwork: the oracle's cached step (prefill, then one token per call) against the reference's own
    past_key_values run - per-step logits incl. the EOI -> newline rule, the layer-0 cache in the reference's layout,
    and consistency with the oracle's uncached forward."""
    t, meta = load_golden("libra_tiny_decode.safetensors")
    w = sub(load_golden("libra_tiny.safetensors")[0], "w.")
    c = meta["cfg"]
    kw = dict(layers=c["num_hidden_layers"], heads=c["num_attention_heads"], vocab=c["vocab_size"],
              max_vision_token_length=c["max_vision_token_length"], eps=c["rms_norm_eps"], max_pos=c["max_position_embeddings"])
    Q, L = c["vision_codebook_num"], c["max_vision_token_length"]
    for name, info in meta["cases"].items():
        ids, vi, sig = t[f"{name}.input_ids"], t[f"{name}.vision_indices"], t[f"{name}.signal"]
        P, S = info["prefill"], info["length"]
        valid = torch.ones(1, S, dtype=torch.bool)
        pos = torch.arange(S).unsqueeze(0)
        hid, flag, caches = LO.model_step(w, ids[:, :, :P], vi[:, :P], sig[:, :P], None, pos[:, :P], valid[:, :P], **kw)
        steps = [LO.cached_logits(w, hid, flag, vi[:, :P], Q, had_past=False, max_vision_token_length=L,
                                  newline_token_id=meta["newline_token_id"])]
        for s in range(P, S):
            hid, flag, caches = LO.model_step(w, ids[:, :, s:s + 1], vi[:, s:s + 1], None, caches, pos[:, s:s + 1], valid[:, :s + 1], **kw)
            steps.append(LO.cached_logits(w, hid, flag, vi[:, s:s + 1], Q, had_past=True, max_vision_token_length=L,
                                          newline_token_id=meta["newline_token_id"]))
        inc = torch.cat(steps, dim=2)
        ref = t[f"{name}.logits_incremental"]
        assert torch.equal(torch.isfinite(inc), torch.isfinite(ref)), name
        assert torch.equal(torch.isposinf(inc), torch.isposinf(ref)), name            # the forced newline
        fin = torch.isfinite(ref)
        assert rel_err(inc[fin], ref[fin]) < 2e-5, (name, rel_err(inc[fin], ref[fin]))
        assert info["eoi_forced_steps"] == ([S - 1] if name == "B" else [])
        kv, kl, v, vb, fl = LO.as_reference_cache(caches[0])
        for got, key in ((kv, "k_for_vision"), (kl, "k_for_language"), (v, "v"), (vb, "v_bridge")):
            assert rel_err(got, t[f"{name}.cache0.{key}"]) < 2e-5, (name, key)
        assert torch.equal(fl.to(torch.uint8), t[f"{name}.cache0.flag"])
        # and the cached path agrees with the oracle's own uncached forward (signal zero at the decoded positions)
        full, fflag = LO.model_forward(w, ids, torch.ones(1, S, dtype=torch.long), vi, sig, **kw)
        zf = LO.vl_logits(w, full, fflag, Q)
        keep = torch.ones(S, dtype=torch.bool); keep[info["eoi_forced_steps"]] = False
        a, b = inc[:, :, keep], zf[:, :, keep]
        m = torch.isfinite(b)
        assert rel_err(a[m], b[m]) < 2e-5, name


def test_greedy_generation_matches_reference_greedy_search():
    """§8f-1: the oracle's generation loop (left-padded prompts, ValidImageLogitsProcessor rule, EOI -> newline, pad after EOS)
    against the reference's own greedy_search run (make_golden_libra_generate.py): identical sequences, per-step scores."""
    t, meta = load_golden("libra_tiny_generate.safetensors")
    w = sub(load_golden("libra_tiny.safetensors")[0], "w.")
    c = meta["cfg"]
    seq, scores = LO.greedy_generate(
        w, t["in.input_ids"], t["in.attention_mask"], t["in.vision_indices"], t["in.signal"], steps=meta["steps"],
        Q=c["vision_codebook_num"], layers=c["num_hidden_layers"], heads=c["num_attention_heads"], vocab=c["vocab_size"],
        max_vision_token_length=c["max_vision_token_length"], newline_token_id=meta["newline_token_id"],
        pad_token_id=meta["pad_token_id"], eos_token_id=meta["eos_token_id"], eps=c["rms_norm_eps"], max_pos=c["max_position_embeddings"],
        image_rule=dict(valid_image_token_length=meta["valid_image_token_length"], boi=meta["boi"], eoi=meta["eoi"], offset=c["vocab_size"]))
    assert torch.equal(seq, t["out.sequences"])
    ref = t["out.scores"]
    assert torch.equal(torch.isfinite(scores), torch.isfinite(ref)) and torch.equal(torch.isposinf(scores), torch.isposinf(ref))
    fin = torch.isfinite(ref)
    assert rel_err(scores[fin], ref[fin]) < 2e-5


def test_f4_variants_match_reference():
    """§8f-4: use_2d_rope, unified_head, vision_prediction_mode='2d' (and 2d rope + 2d prediction together), use_bridge=False, the embedding-stage switches, addition_mode - the oracle's
    restatements against the reference's own forward + autograd (tests/golden/make_golden_libra_f4.py)."""
    t, meta = load_golden("libra_tiny_f4.safetensors")
    t0, _ = load_golden("libra_tiny.safetensors")
    c = meta["cfg"]
    L, res, Q = c["max_vision_token_length"], c["image_feature_resolution"], c["vision_codebook_num"]
    kw = dict(layers=c["num_hidden_layers"], heads=c["num_attention_heads"], vocab=c["vocab_size"], max_vision_token_length=L,
              eps=c["rms_norm_eps"], max_pos=c["max_position_embeddings"])
    ids, am, vi, sig, labels = t0["in.input_ids"], t0["in.attention_mask"], t0["in.vision_indices"], t0["in.signal"], t0["in.labels"]
    for name, over in meta["variants"].items():
        sd = {k: v.clone() for k, v in sub(t0, "w.").items()}
        if over.get("use_bridge") is False:                    # the reference layer then has no bridge parameters (:258)
            sd = {k: v for k, v in sd.items() if "_bridge_on_" not in k}
        if over.get("norm_signals") is False or over.get("concat_signals") is False:      # no vision_signal_norm module (:558)
            sd.pop("model.vision_signal_norm.weight")
        for k, v in sub(t, f"{name}.w.").items():
            sd[k] = v.clone()
        sd = {k: v.requires_grad_(True) for k, v in sd.items()}
        hid, flag = LO.model_forward(sd, ids, am, vi, sig, rope_2d_res=res if over.get("use_2d_rope") else None,
                                     addition=bool(over.get("addition_mode")), **kw)
        if over.get("use_2d_rope"):
            assert torch.equal(LO.position_ids_2d(vi, L, res), t[f"{name}.position_ids"])
        if over.get("unified_head"):
            z = LO.vl_logits_unified(sd, hid, Q)
        elif over.get("vision_prediction_mode") == "2d":
            z = LO.vl_logits_2d(sd, hid, flag, Q, L, res)
        else:
            z = LO.vl_logits(sd, hid, flag, Q)
        loss = LO.causal_lm_loss(z, labels)
        loss.backward()
        assert rel_err(hid.detach(), t[f"{name}.hidden"]) < 2e-5, name
        ref = t[f"{name}.logits"]
        assert torch.equal(torch.isfinite(z), torch.isfinite(ref)), name
        fin = torch.isfinite(ref)
        assert rel_err(z.detach()[fin], ref[fin]) < 2e-5, name
        assert abs(float(loss) - float(t[f"{name}.loss"])) < 1e-5 * abs(float(t[f"{name}.loss"])), name
        n = 0
        for k, g in sub(t, f"{name}.grad.").items():
            assert sd[k].grad is not None, (name, k)
            assert rel_err(sd[k].grad, g) < 2e-4, (name, k, rel_err(sd[k].grad, g))
            n += 1
        assert n >= (10 if over.get("use_bridge") is False else 11), (name, n)
        for k in sub(t, f"{name}.w."):                           # the variant's own extra weights have their reference gradient too
            assert f"{name}.grad.{k}" in t or "placeholder" in k or "heads" in k, (name, k)


# ---- tiny width x FULL depth (32 layers): SURVEY §8c(i) ------------------------------------------------------------------
def depth32_state(meta, names_shapes):
    """The fixture's weights, re-derived (tests/golden/seeded_weights.py) and checked against the generating run's checksum."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from seeded_weights import checksum, seeded_state
    sd = seeded_state(names_shapes, meta["seed"])
    cs, ref = checksum(sd), meta["checksum"]
    assert cs["n"] == ref["n"] == meta["n_params"] and cs["first"] == ref["first"]
    assert abs(cs["sum"] - ref["sum"]) <= 1e-9 * ref["abs_sum"] and abs(cs["abs_sum"] - ref["abs_sum"]) <= 1e-9 * ref["abs_sum"]
    return sd


def depth32_names_shapes(c):
    """(name, shape) of every parameter of LibraForCausalLM(c): the reference's state-dict layout (checked against the fixture's
    parameter count and the product model's own parameters in the GPU test)."""
    H, I, V, Vv, Q = c["hidden_size"], c["intermediate_size"], c["vocab_size"], c["vision_vocab_size"], c["vision_codebook_num"]
    r, rg, rank, Cs = H // c["vision_down_ratio"], I // c["vision_down_ratio"], c["bridge_rank"], c["contiguous_signal_size"]
    out = [("model.embed_tokens.weight", (V, H))] + [(f"model.vision_embed_tokens.{q}.weight", (Vv, H // Q)) for q in range(Q)]
    out += [("model.vision_signal_norm.weight", (H + Cs,)), ("model.vision_contiguous_signal_processor.weight", (H, H + Cs))]
    for i in range(c["num_hidden_layers"]):
        a, m = f"model.layers.{i}.self_attn.", f"model.layers.{i}.mlp."
        for n in "qkvo":
            out += [(a + f"{n}_proj.weight", (H, H)), (a + f"vision_{n}_proj.weight_A", (r, H)), (a + f"vision_{n}_proj.weight_B", (H, r))]
        for kv in "kv":
            for w in ("language", "vision"):
                out += [(a + f"vision_{kv}_bridge_on_{w}.weight_A", (rank, H)), (a + f"vision_{kv}_bridge_on_{w}.weight_B", (H, rank))]
        out += [(m + "gate_proj.weight", (I, H)), (m + "up_proj.weight", (I, H)), (m + "down_proj.weight", (H, I)),
                (m + "vision_gate_proj.weight_A", (rg, H)), (m + "vision_gate_proj.weight_B", (I, rg)),
                (m + "vision_up_proj.weight_A", (rg, H)), (m + "vision_up_proj.weight_B", (I, rg)),
                (m + "vision_down_proj.weight_A", (r, I)), (m + "vision_down_proj.weight_B", (H, r))]
        p = f"model.layers.{i}."
        out += [(p + n, (H,)) for n in ("input_layernorm.weight", "vision_input_layernorm.weight",
                                        "post_attention_layernorm.weight", "vision_post_attention_layernorm.weight")]
    out += [("model.norm.weight", (H,)), ("model.vision_norm.weight", (H,)), ("lm_head.weight", (V, H))]
    out += [(f"vision_lm_head.heads.{q}.weight", (Vv, H)) for q in range(Q)]
    out += [("vision_hidden_placeholder", (H,))]
    return out


def test_depth32_oracle_matches_reference_at_every_depth():
    """The oracle against the reference's own 32-layer run: all 33 hidden states, logits, loss and autograd gradients."""
    t, meta = load_golden("libra_tiny_depth32.safetensors")
    c = meta["cfg"]
    ns = depth32_names_shapes(c)
    assert len(ns) == meta["n_params"]
    sd = {k: v.requires_grad_(True) for k, v in depth32_state(meta, ns).items()}
    hs = []
    hid, flag = LO.model_forward(sd, t["in.input_ids"], t["in.attention_mask"], t["in.vision_indices"], t["in.signal"],
                                 layers=c["num_hidden_layers"], heads=c["num_attention_heads"], vocab=c["vocab_size"],
                                 max_vision_token_length=c["max_vision_token_length"], eps=c["rms_norm_eps"],
                                 max_pos=c["max_position_embeddings"], hidden_states=hs)
    ref_hs = t["out.hidden_states"]
    assert len(hs) == ref_hs.shape[0] == 33
    valid = t["in.attention_mask"].bool()
    # the reference's tuple = embeddings, the outputs of layers 0..30, and the NORMED output of layer 31 (:809-813)
    worst = max(rel_err(h.detach()[valid], ref_hs[l][valid]) for l, h in enumerate(hs[:32]))
    worst = max(worst, rel_err(hid.detach()[valid], ref_hs[32][valid]))
    assert worst < 2e-5, worst
    logits = LO.vl_logits(sd, hid, flag, c["vision_codebook_num"])
    fin = torch.isfinite(t["out.logits"])
    assert torch.equal(fin, torch.isfinite(logits)) and rel_err(logits.detach()[fin], t["out.logits"][fin]) < 2e-5
    loss = LO.causal_lm_loss(logits, t["in.labels"])
    assert abs(float(loss) - float(t["out.loss"])) < 1e-5 * abs(float(t["out.loss"]))
    loss.backward()
    n = 0
    for k, g in sub(t, "grad.").items():
        if k == "vision_hidden_placeholder":
            continue
        gmax = float(g.float().abs().max())
        if gmax < 1e-6:
            assert float(sd[k].grad.abs().max()) < 1e-5, k
        else:
            tol = 2e-3 if g.dtype == torch.float16 else 2e-4
            assert rel_err(sd[k].grad, g.float()) < tol, (k, rel_err(sd[k].grad, g.float()))
        n += 1
    assert n == meta["n_grads"] - (1 if "grad.vision_hidden_placeholder" in t else 0) and n > 400


def bf16_yardsticks(meta32, t32):
    """-> (per-depth error of the REFERENCE run in bf16, per-depth error of the ORACLE run in bf16), both against the reference's
    fp32 hidden states of libra_tiny_depth32.safetensors; and the two bf16 runs' hidden states for a direct comparison."""
    tb, mb = load_golden("libra_tiny_depth32_bf16.safetensors")
    assert mb["seed"] == meta32["seed"] and mb["checksum"]["abs_sum"] == meta32["checksum"]["abs_sum"]
    c = meta32["cfg"]
    sdb = {k: v.to(torch.bfloat16) for k, v in depth32_state(meta32, depth32_names_shapes(c)).items()}
    hsb = []
    hidb, _ = LO.model_forward(sdb, t32["in.input_ids"], t32["in.attention_mask"], t32["in.vision_indices"],
                               t32["in.signal"].to(torch.bfloat16), layers=c["num_hidden_layers"], heads=c["num_attention_heads"],
                               vocab=c["vocab_size"], max_vision_token_length=c["max_vision_token_length"], eps=c["rms_norm_eps"],
                               max_pos=c["max_position_embeddings"], hidden_states=hsb)
    hsb = hsb[:32] + [hidb]
    valid = t32["in.attention_mask"].bool()
    ref32, refb = t32["out.hidden_states"], tb["out.hidden_states"]
    e_ref = [rel_err(refb[l].float()[valid], ref32[l][valid]) for l in range(33)]
    e_orc = [rel_err(hsb[l].float()[valid], ref32[l][valid]) for l in range(33)]
    return e_ref, e_orc, refb, hsb, tb


def test_depth32_bf16_yardstick_is_the_references_own_bf16_run():
    """"theirs" in the model-level parity gates (ours <= max(k x theirs, floor)) is the bf16 rounding error of the REFERENCE
    itself: tests/golden/libra_tiny_depth32_bf16.safetensors holds the reference's LibraForCausalLM run under
    `model.to(torch.bfloat16)` (train.py:31-32) on the depth-32 case.  Pinned here: (i) the oracle executed op by op in bf16 - the
    yardstick the full-size GPU tests have to use, the reference being absent there - makes an error of the same size at every
    depth (within 1.5x of the reference's either way; the two runs round at slightly different points, e.g. inside the norms, so
    they differ from each other by about the sum of their errors, never more); (ii) the size of that error: it grows to ~3e-2 over 32 layers, which is why north_star's 1e-3 is a
    per-kernel bound, not a model-level one."""
    t, meta = load_golden("libra_tiny_depth32.safetensors")
    e_ref, e_orc, refb, hsb, tb = bf16_yardsticks(meta, t)
    valid = t["in.attention_mask"].bool()
    for l in range(33):
        assert e_orc[l] <= 1.5 * e_ref[l] + 1e-4 and e_ref[l] <= 1.5 * e_orc[l] + 1e-4, (l, e_ref[l], e_orc[l])
        same = rel_err(hsb[l].float()[valid], refb[l].float()[valid])
        assert same <= 1.25 * (e_ref[l] + e_orc[l]), (l, same, e_ref[l], e_orc[l])
    assert 5e-3 < e_ref[32] < 1e-1 and e_ref[1] < 8e-3, (e_ref[1], e_ref[32])
    loss32, lossb = float(t["out.loss"]), float(tb["out.loss"])
    assert abs(lossb - loss32) < 3e-2 * abs(loss32)
