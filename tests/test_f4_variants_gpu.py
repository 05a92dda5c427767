"""SURVEY §8f-4 on device: use_2d_rope, unified_head, vision_prediction_mode="2d" (and 2d RoPE + 2d prediction together), use_bridge=False
the embedding-stage switches (use_vision_position_embedding, norm_signals=False, concat_signals=False) and addition_mode against
the fixture produced by the reference's own forward + autograd (tests/golden/make_golden_libra_f4.py) and the fp32 oracle on the
same bf16-rounded weights; the cached generation path of each variant against the uncached forward."""
import pytest
import torch

from helpers import load_golden, parity_report, rel_err, sub

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
VARIANTS = ("rope2d", "unified", "pred2d", "rope2d_pred2d", "nobridge", "vispos", "nonorm", "noconcat", "addition")


def _build(name):
    from libra_amd.libra import LibraConfig, LibraForCausalLM
    t, meta = load_golden("libra_tiny_f4.safetensors")
    t0, _ = load_golden("libra_tiny.safetensors")
    over = meta["variants"][name]
    w = dict(sub(t0, "w."))
    if over.get("use_bridge") is False:                                    # no bridge parameters in the state dict (:258)
        w = {k: v for k, v in w.items() if "_bridge_on_" not in k}
    if over.get("norm_signals") is False or over.get("concat_signals") is False:      # no vision_signal_norm module (:558)
        w.pop("model.vision_signal_norm.weight")
    w.update(sub(t, f"{name}.w."))
    m = LibraForCausalLM(LibraConfig(**dict(meta["cfg"], **over)))
    m.load_state_dict(w, strict=True)
    return m.to(BF).cuda(), w, t, t0, meta, over


def _oracle_logits(LO, sd, hid, flag, over, c):
    Q, L, res = c["vision_codebook_num"], c["max_vision_token_length"], c["image_feature_resolution"]
    if over.get("unified_head"):
        return LO.vl_logits_unified(sd, hid, Q)
    if over.get("vision_prediction_mode") == "2d":
        return LO.vl_logits_2d(sd, hid, flag, Q, L, res)
    return LO.vl_logits(sd, hid, flag, Q)


@pytest.mark.parametrize("name", VARIANTS)
def test_f4_forward_backward_vs_reference_fixture(name):
    from oracle import libra_oracle as LO
    m, w, t, t0, meta, over = _build(name)
    c = meta["cfg"]
    m.requires_grad_(True)
    ids, am, vi = t0["in.input_ids"].cuda(), t0["in.attention_mask"].cuda(), t0["in.vision_indices"].cuda()
    sig, lab = t0["in.signal"].to(BF).cuda(), t0["in.labels"].cuda()
    out = m(input_ids=ids, attention_mask=am, vision_indices=vi, contiguous_signal=sig, labels=lab, output_hidden_states=True)
    out.loss.backward()
    # fp32 oracle on the bf16-rounded weights / signal: only the arithmetic differs
    sdf = {k: v.to(BF).float().requires_grad_(True) for k, v in w.items()}
    kw = dict(layers=c["num_hidden_layers"], heads=c["num_attention_heads"], vocab=c["vocab_size"],
              max_vision_token_length=c["max_vision_token_length"], eps=c["rms_norm_eps"], max_pos=c["max_position_embeddings"],
              rope_2d_res=c["image_feature_resolution"] if over.get("use_2d_rope") else None, addition=bool(over.get("addition_mode")))
    hid, flag = LO.model_forward(sdf, t0["in.input_ids"], t0["in.attention_mask"], t0["in.vision_indices"],
                                 t0["in.signal"].to(BF).float(), **kw)
    z = _oracle_logits(LO, sdf, hid, flag, over, c)
    ref_loss = LO.causal_lm_loss(z, t0["in.labels"])
    ref_loss.backward()
    valid = t0["in.attention_mask"].bool()
    e_h = rel_err(out.hidden_states[-1].float().cpu()[valid], hid.detach()[valid])
    with torch.no_grad():                                               # the same oracle run op by op in bf16: the noise floor
        hidb, _ = LO.model_forward({k: v.to(BF) for k, v in w.items()}, t0["in.input_ids"], t0["in.attention_mask"],
                                   t0["in.vision_indices"], t0["in.signal"].to(BF), **kw)
    theirs = rel_err(hidb.float()[valid], hid.detach()[valid])
    assert e_h < max(1.5 * theirs, 3e-3), (name, e_h, theirs)
    # against the reference's own run (fp32, UN-rounded weights): the bf16 op-by-op oracle's distance to it is the yardstick
    floor = rel_err(hidb.float()[valid], t[f"{name}.hidden"][valid])
    assert rel_err(out.hidden_states[-1].float().cpu()[valid], t[f"{name}.hidden"][valid]) < max(2e-2, 1.5 * floor), (name, floor)
    logits = out.logits.float().cpu()
    assert logits.shape == t[f"{name}.logits"].shape
    assert torch.equal(torch.isfinite(logits), torch.isfinite(t[f"{name}.logits"])), name
    fin = torch.isfinite(z.detach()) & valid[None, :, :, None]
    e_z = rel_err(logits[fin], z.detach()[fin])
    assert e_z < max(2.0 * theirs, 6e-3), (name, e_z, theirs)
    assert abs(float(out.loss) - float(ref_loss)) < 2e-2 * abs(float(ref_loss)), (name, float(out.loss), float(ref_loss))
    assert abs(float(out.loss) - float(t[f"{name}.loss"])) < 5e-2 * abs(float(t[f"{name}.loss"]))
    worst, n = ("", 0.0), 0
    for pname, p in m.named_parameters():
        ref = sdf[pname].grad
        if pname == "vision_hidden_placeholder" and over.get("vision_prediction_mode") != "2d":
            continue
        assert p.grad is not None and ref is not None, (name, pname)
        if float(ref.abs().max()) < 1e-7:
            assert float(p.grad.float().abs().max()) < 1e-4, pname
            continue
        e = rel_err(p.grad.float().cpu(), ref)
        worst = (pname, e) if e > worst[1] else worst
        assert e < 6e-2, (name, pname, e)        # bf16 activations + bf16 gradients through 2 layers (rank-8 bridge A is the noisiest)
        key = f"{name}.grad.{pname}"
        if key in t:                                                     # the reference's own autograd (fp32 weights)
            assert rel_err(p.grad.float().cpu(), t[key].float()) < 8e-2, (name, pname)
            n += 1
    assert n >= (10 if over.get("use_bridge") is False else 11), (name, n)
    parity_report(f"[f4 {name}, tiny] loss ours {float(out.loss):.4f} reference {float(t[f'{name}.loss']):.4f}; hidden rel err {e_h:.2e} (bf16 op-by-op oracle {theirs:.2e}), "
                  f"logits {e_z:.2e}; worst of {sum(1 for _ in m.parameters())} weight gradients vs fp32 oracle {worst[1]:.2e} at {worst[0]}")


@pytest.mark.parametrize("name", VARIANTS)
@pytest.mark.parametrize("graphs", (True, False))
def test_f4_cached_steps_match_uncached_forward(name, graphs):
    """Prefill + token-by-token cached steps (the generation glue: prepare_inputs_for_generation /
    _update_model_kwargs_for_generation) must reproduce the uncached forward over the whole sequence - which the test above pins
    to the reference - for a left-padded batch that generates an image (2d positions, the cell above / to the left, </img>)."""
    m, w, t, t0, meta, over = _build(name)
    m.eval()
    m.decode_graphs = graphs
    g, gmeta = load_golden("libra_tiny_generate.safetensors")
    S, T = gmeta["prompt_len"], gmeta["steps"]
    seq = g["out.sequences"].cuda()                                        # [Q, B, S + T]: a well-formed continuation
    kwargs = dict(attention_mask=g["in.attention_mask"].cuda(), vision_indices=g["in.vision_indices"].cuda(),
                  contiguous_signal=g["in.signal"].to(BF).cuda(), use_cache=True)
    cached = []
    with torch.no_grad():
        for step in range(T):
            inputs = m.prepare_inputs_for_generation(seq[:, :, :S + step], **kwargs)
            out = m(**inputs, max_cache_len=S + T + 1)
            cached.append(out.logits[:, :, -1].float().cpu())
            kwargs = m._update_model_kwargs_for_generation(out, kwargs)
        vi_full = kwargs["vision_indices"][:, :S + T - 1]
        am_full = kwargs["attention_mask"][:, :S + T - 1]
        sig = torch.zeros((seq.shape[1], S + T - 1, g["in.signal"].shape[-1]), dtype=BF, device="cuda")
        sig[:, :S] = g["in.signal"].to(BF).cuda()
        full = m(input_ids=seq[:, :, :S + T - 1], attention_mask=am_full, vision_indices=vi_full, contiguous_signal=sig)
    dense = full.logits.float().cpu()
    L = gmeta["cfg"]["max_vision_token_length"]
    worst = 0.0
    for step in range(T):
        a, b = cached[step], dense[:, :, S - 1 + step]
        eoi_in = (vi_full[:, S - 1 + step] == L - 1).cpu()                 # cached branch: </img> predicts only a newline (:1141-1144)
        fin = torch.isfinite(a) & ~eoi_in[None, :, None]
        assert bool(torch.isfinite(b[fin]).all()), (name, step)
        if not over.get("unified_head"):                                   # (the unified head's uncached rows are dense, cached masked)
            assert torch.equal(torch.isfinite(a)[:, ~eoi_in], torch.isfinite(b)[:, ~eoi_in]), (name, step)
        e = rel_err(a[fin], b[fin])
        worst = max(worst, e)
        assert e < 2e-2, (name, step, e)
    parity_report(f"[f4 {name}, tiny, graphs={graphs}] {T} cached steps vs the uncached forward: worst logits rel err {worst:.2e}")
