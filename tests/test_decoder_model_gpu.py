"""The routed decoder (forward) on MI355X against the reference fixture / the CPU oracle."""
import pytest
import torch

from helpers import load_golden, rel_err, sub, sub_params

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def test_libra_tiny_forward_vs_reference_fixture():
    from libra_amd.libra import LibraConfig, LibraForCausalLM
    from oracle import libra_oracle as LO
    t, meta = load_golden("libra_tiny.safetensors")
    c = meta["cfg"]
    m = LibraForCausalLM(LibraConfig(**c))
    missing, unexpected = m.load_state_dict(sub(t, "w."), strict=True)
    m = m.to(BF).cuda().eval()
    ids, am, vi = t["in.input_ids"].cuda(), t["in.attention_mask"].cuda(), t["in.vision_indices"].cuda()
    sig, lab = t["in.signal"].to(BF).cuda(), t["in.labels"].cuda()
    with torch.no_grad():
        out = m(input_ids=ids, attention_mask=am, vision_indices=vi, contiguous_signal=sig, labels=lab,
                output_hidden_states=True)
    # fp32 oracle on the bf16-rounded weights / signal: only the arithmetic differs
    sdf = {k: v.to(BF).float() for k, v in sub(t, "w.").items()}
    sdb = {k: v.to(BF) for k, v in sub(t, "w.").items()}
    kw = dict(layers=c["num_hidden_layers"], heads=c["num_attention_heads"], vocab=c["vocab_size"],
              max_vision_token_length=c["max_vision_token_length"], eps=c["rms_norm_eps"], max_pos=c["max_position_embeddings"])
    hid, flag = LO.model_forward(sdf, t["in.input_ids"], t["in.attention_mask"], t["in.vision_indices"],
                                 t["in.signal"].to(BF).float(), **kw)
    hidb, _ = LO.model_forward(sdb, t["in.input_ids"], t["in.attention_mask"], t["in.vision_indices"], t["in.signal"].to(BF), **kw)
    valid = t["in.attention_mask"].bool()
    ours = rel_err(out.hidden_states[-1].float().cpu()[valid], hid[valid])
    theirs = rel_err(hidb.float()[valid], hid[valid])
    assert ours < max(1.5 * theirs, 3e-3), (ours, theirs)
    # logits: identical -inf pattern; finite part close; loss within bf16-logit noise of the fp32 loss
    logits = LibraForCausalLM.materialize_logits(out).float().cpu()
    ref_logits = LO.vl_logits(sdf, hid, flag, c["vision_codebook_num"])
    assert logits.shape == t["out.logits"].shape
    assert torch.equal(torch.isfinite(logits), torch.isfinite(ref_logits))
    fin = torch.isfinite(ref_logits) & valid[None, :, :, None]
    assert rel_err(logits[fin], ref_logits[fin]) < max(2.0 * theirs, 6e-3)
    ref_loss = float(LO.causal_lm_loss(ref_logits, t["in.labels"]))
    assert abs(float(out.loss) - ref_loss) < 2e-2 * abs(ref_loss), (float(out.loss), ref_loss)
    assert abs(float(out.loss) - float(t["out.loss"])) < 5e-2 * abs(float(t["out.loss"]))


def test_libra_tiny_backward_vs_reference_fixture():
    """Hand-written decoder backward vs (a) autograd through the fp32 oracle on the same bf16-rounded weights and
    (b) the reference's own autograd gradients stored in the fixture."""
    from libra_amd.libra import LibraConfig, LibraForCausalLM
    from oracle import libra_oracle as LO
    t, meta = load_golden("libra_tiny.safetensors")
    c = meta["cfg"]
    m = LibraForCausalLM(LibraConfig(**c))
    m.load_state_dict(sub(t, "w."), strict=True)
    m = m.to(BF).cuda()
    m.requires_grad_(True)
    ids, am, vi = t["in.input_ids"].cuda(), t["in.attention_mask"].cuda(), t["in.vision_indices"].cuda()
    sig, lab = t["in.signal"].to(BF).cuda(), t["in.labels"].cuda()
    out = m(input_ids=ids, attention_mask=am, vision_indices=vi, contiguous_signal=sig, labels=lab)
    out.loss.backward()
    sdf = {k: v.to(BF).float().requires_grad_(True) for k, v in sub_params(t, "w.").items()}
    kw = dict(layers=c["num_hidden_layers"], heads=c["num_attention_heads"], vocab=c["vocab_size"],
              max_vision_token_length=c["max_vision_token_length"], eps=c["rms_norm_eps"], max_pos=c["max_position_embeddings"])
    hid, flag = LO.model_forward(sdf, t["in.input_ids"], t["in.attention_mask"], t["in.vision_indices"],
                                 t["in.signal"].to(BF).float(), **kw)
    LO.causal_lm_loss(LO.vl_logits(sdf, hid, flag, c["vision_codebook_num"]), t["in.labels"]).backward()
    worst, n = ("", 0.0), 0
    for name, p in m.named_parameters():
        if name == "vision_hidden_placeholder":
            continue
        ref = sdf[name].grad
        assert p.grad is not None, name
        gmax = float(ref.abs().max())
        if gmax < 1e-7:
            assert float(p.grad.float().abs().max()) < 1e-4, name
            continue
        e = rel_err(p.grad.float().cpu(), ref)
        if e > worst[1]:
            worst = (name, e)
        assert e < 4e-2, (name, e)                                   # bf16 activations + bf16 gradients, 2 layers
        e2 = rel_err(p.grad.float().cpu(), t["grad." + name].float())
        assert e2 < 8e-2, (name, e2)                                 # vs the reference's fp32-weight autograd run
        n += 1
    print("decoder backward: worst rel err", worst)
    assert n > 60


def test_libra_full_width_single_layer_vs_oracle():
    """One full-width Libra-11B decoder layer (H=4096, I=11008, 32 heads x 128, rank-8 bridges) at S=256:
    the hot-loop body of BASELINE config 3, against the fp32 oracle on the host."""
    from libra_amd import decoder_engine as DE
    from oracle import libra_oracle as LO
    g = torch.Generator().manual_seed(5)
    H, I, heads, r, rg = 4096, 11008, 32, 1024, 2752
    B, S, L = 2, 256, 70
    sd = {}

    def rn(*shape, std):
        return (torch.randn(*shape, generator=g) * std).to(BF)
    p = "model.layers.0."
    for n in ("q", "k", "v", "o"):
        sd[p + f"self_attn.{n}_proj.weight"] = rn(H, H, std=H ** -0.5)
        sd[p + f"self_attn.vision_{n}_proj.weight_A"] = rn(r, H, std=H ** -0.5)
        sd[p + f"self_attn.vision_{n}_proj.weight_B"] = rn(H, r, std=r ** -0.5)
    for kv in ("k", "v"):
        for w in ("language", "vision"):
            sd[p + f"self_attn.vision_{kv}_bridge_on_{w}.weight_A"] = rn(8, H, std=H ** -0.5)
            sd[p + f"self_attn.vision_{kv}_bridge_on_{w}.weight_B"] = rn(H, 8, std=0.3)
    sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"] = rn(I, H, std=H ** -0.5), rn(I, H, std=H ** -0.5)
    sd[p + "mlp.down_proj.weight"] = rn(H, I, std=I ** -0.5)
    for n in ("gate", "up"):
        sd[p + f"mlp.vision_{n}_proj.weight_A"], sd[p + f"mlp.vision_{n}_proj.weight_B"] = rn(rg, H, std=H ** -0.5), rn(I, rg, std=rg ** -0.5)
    sd[p + "mlp.vision_down_proj.weight_A"], sd[p + "mlp.vision_down_proj.weight_B"] = rn(r, I, std=I ** -0.5), rn(H, r, std=r ** -0.5)
    for n in ("input_layernorm", "post_attention_layernorm", "vision_input_layernorm", "vision_post_attention_layernorm"):
        sd[p + n + ".weight"] = (1 + 0.1 * torch.randn(H, generator=g)).to(BF)
    sd["model.embed_tokens.weight"] = rn(8, H, std=1.0)
    x = rn(B, S, H, std=1.0)
    vi = torch.full((B, S), L, dtype=torch.long)
    vi[0, 3:3 + L] = torch.arange(L); vi[1, 100:100 + L] = torch.arange(L)
    am = torch.ones(B, S, dtype=torch.long); am[1, 200:] = 0
    d = DE.DecDims(hidden=H, inter=I, layers=1, heads=heads, vocab=32000, vision_vocab=514, codebooks=2, max_vision_len=L,
                   signal=2048)
    dsd = {k: v.cuda() for k, v in sd.items()}
    pk = DE.pack(dsd, d)
    flag, li, vidx, lens, _ = DE.route(vi.cuda(), am.cuda(), d)
    cos, sin = DE.rope_tables(128, 2048, "cuda")
    y = DE.layer_forward(dsd, pk[0], 0, d, x.view(B * S, H).cuda(), flag, li, vidx, lens, cos, sin, B, S).view(B, S, H)
    f = vi < L
    cosr, sinr = LO.rope_tables(128, 2048)
    pos = torch.arange(S).unsqueeze(0).expand(B, S)
    sdf = {k: v.float() for k, v in sd.items()}
    ref = LO.decoder_layer(sdf, 0, x.float(), f, LO.additive_mask(am, S, torch.float32), pos, heads, 1e-6, cosr, sinr)
    refb = LO.decoder_layer(sd, 0, x, f, LO.additive_mask(am, S, BF), pos, heads, 1e-6, cosr.to(BF), sinr.to(BF))
    valid = am.bool()
    ours, theirs = rel_err(y.float().cpu()[valid], ref[valid]), rel_err(refb.float()[valid], ref[valid])
    print(f"full-width layer: ours {ours:.3e}, reference-style bf16 {theirs:.3e}")
    assert ours < max(1.5 * theirs, 3e-3), (ours, theirs)


def test_libra_backward_emits_trainable_gradients_to_a_capturing_reducer():
    """Same contract as the ViT test for the decoder under the pretraining freeze policy: only the trainable
    ("vision") gradients are exchanged, all of them leave during the backward (weight-gradient GEMMs writing straight into
    the flat buckets), results equal the plain backward; a second step reuses the buckets."""
    from libra_amd import decoder_engine as DE
    from libra_amd import dp
    from libra_amd.libra import LibraConfig, LibraForCausalLM, apply_freeze_policy
    t, meta = load_golden("libra_tiny.safetensors")
    m = LibraForCausalLM(LibraConfig(**meta["cfg"]))
    m.load_state_dict(sub(t, "w."), strict=True)
    m = m.to(BF).cuda()
    apply_freeze_policy(m, frozen_language=True)
    kw = dict(input_ids=t["in.input_ids"].cuda(), attention_mask=t["in.attention_mask"].cuda(),
              vision_indices=t["in.vision_indices"].cuda(), contiguous_signal=t["in.signal"].to(BF).cuda(),
              labels=t["in.labels"].cuda())
    m(**kw).loss.backward()
    named = [(n, p) for n, p in m.named_parameters() if p.requires_grad and n != "vision_hidden_placeholder"]
    plain = {n: p.grad.clone() for n, p in named if p.grad is not None}
    assert plain and all("vision" in n for n in plain) and set(plain) == {n for n, _ in named}
    m.zero_grad(set_to_none=True)
    L = meta["cfg"]["num_hidden_layers"]
    st = dp.GradBuckets(named, bucket_bytes=1 << 14, group_fn=lambda n: DE.emit_group(n, L))
    for _ in range(2):
        with st.capture():
            m(**kw).loss.backward()
        captured, sent = set(st._seen), st._next
        st.finish_into(named)
        assert captured == set(plain), set(plain) ^ captured
        assert sent == len(st.buckets)                     # every bucket left while the backward was running
        for n, p in named:
            assert torch.equal(p.grad, plain[n]), n
            assert p.grad.data_ptr() == st.view(n).data_ptr()


def test_libra_text_only_batch_emits_zero_vision_gradients():
    """a micro-batch without a single vision token must still produce (zero) gradients for every trainable
    vision parameter - the reference runs the vision modules on empty tensors and autograd yields zeros - so that all
    data-parallel ranks exchange the same buckets; text parameters get their usual gradients."""
    from libra_amd.libra import LibraConfig, LibraForCausalLM
    t, meta = load_golden("libra_tiny.safetensors")
    c = meta["cfg"]
    m = LibraForCausalLM(LibraConfig(**c))
    m.load_state_dict(sub(t, "w."), strict=True)
    m = m.to(BF).cuda()
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(3, c["vocab_size"], (1, 2, 12), generator=g).repeat(2, 1, 1).cuda()
    ids[:, :, 0] = 1
    vi = torch.full((2, 12), c["max_vision_token_length"], dtype=torch.long).cuda()
    labels = ids.clone(); labels[:, :, 0] = -100
    out = m(input_ids=ids, attention_mask=torch.ones(2, 12, dtype=torch.long).cuda(), vision_indices=vi, contiguous_signal=None,
            labels=labels)
    assert torch.isfinite(out.loss)
    out.loss.backward()
    for n, p in m.named_parameters():
        if n == "vision_hidden_placeholder":
            assert p.grad is None
        elif "vision" in n:
            assert p.grad is not None and float(p.grad.float().abs().max()) == 0.0, n
        else:
            assert p.grad is not None and float(p.grad.float().abs().max()) > 0.0, n


def test_libra_tiny_cached_decode_vs_reference_fixture():
    """Generation path: prefill with use_cache, then one token per call with past_key_values, against the
    per-step logits of the reference's own cached run (fixture) and against the fp32 oracle's cached step on the bf16-rounded
    weights; plus consistency with this model's own uncached forward."""
    from libra_amd.libra import LibraConfig, LibraForCausalLM
    from oracle import libra_oracle as LO
    t, meta = load_golden("libra_tiny_decode.safetensors")
    w = sub(load_golden("libra_tiny.safetensors")[0], "w.")
    c = meta["cfg"]
    m = LibraForCausalLM(LibraConfig(**c))
    m.load_state_dict(w, strict=True)
    m = m.to(BF).cuda().eval()
    sdf = {k: v.to(BF).float() for k, v in w.items()}
    kw = dict(layers=c["num_hidden_layers"], heads=c["num_attention_heads"], vocab=c["vocab_size"],
              max_vision_token_length=c["max_vision_token_length"], eps=c["rms_norm_eps"], max_pos=c["max_position_embeddings"])
    Q, L = c["vision_codebook_num"], c["max_vision_token_length"]
    for name, info in meta["cases"].items():
        ids, vi, sig = t[f"{name}.input_ids"].cuda(), t[f"{name}.vision_indices"].cuda(), t[f"{name}.signal"].to(BF).cuda()
        P, S = info["prefill"], info["length"]
        with torch.no_grad():
            out = m(input_ids=ids[:, :, :P], vision_indices=vi[:, :P], contiguous_signal=sig[:, :P], use_cache=True)
            steps, past = [out.logits], out.past_key_values
            assert past.get_seq_length() == P
            for s in range(P, S):
                out = m(input_ids=ids[:, :, s:s + 1], vision_indices=vi[:, s:s + 1], past_key_values=past, use_cache=True,
                        position_ids=torch.tensor([[s]], device="cuda"))
                past = out.past_key_values
                steps.append(out.logits)
            assert past.get_seq_length() == S
            full = m(input_ids=ids, vision_indices=vi, contiguous_signal=sig)
            # the same steps launched kernel by kernel (no hipGraph capture): bit-identical
            m.decode_graphs = False
            out2 = m(input_ids=ids[:, :, :P], vision_indices=vi[:, :P], contiguous_signal=sig[:, :P], use_cache=True)
            steps2, past2 = [out2.logits], out2.past_key_values
            for s in range(P, S):
                out2 = m(input_ids=ids[:, :, s:s + 1], vision_indices=vi[:, s:s + 1], past_key_values=past2, use_cache=True)
                steps2.append(out2.logits)
            m.decode_graphs = True
            a2, b2 = torch.cat(steps2, dim=2), torch.cat(steps, dim=2)
            assert torch.equal(torch.nan_to_num(a2.float(), posinf=1e30, neginf=-1e30), torch.nan_to_num(b2.float(), posinf=1e30, neginf=-1e30)), name
        inc = torch.cat(steps, dim=2).float().cpu()
        ref = t[f"{name}.logits_incremental"]
        assert inc.shape == ref.shape
        assert torch.equal(torch.isfinite(inc), torch.isfinite(ref)), name             # -inf padding pattern
        assert torch.equal(torch.isposinf(inc), torch.isposinf(ref)), name             # EOI -> newline
        # fp32 oracle cached run on the bf16-rounded weights: only the arithmetic differs
        valid = torch.ones(1, S, dtype=torch.bool)
        pos = torch.arange(S).unsqueeze(0)
        sgf = t[f"{name}.signal"].to(BF).float()
        hid, flag, caches = LO.model_step(sdf, t[f"{name}.input_ids"][:, :, :P], t[f"{name}.vision_indices"][:, :P], sgf[:, :P], None,
                                          pos[:, :P], valid[:, :P], **kw)
        osteps = [LO.vl_logits(sdf, hid, flag, Q)]
        for s in range(P, S):
            hid, flag, caches = LO.model_step(sdf, t[f"{name}.input_ids"][:, :, s:s + 1], t[f"{name}.vision_indices"][:, s:s + 1], None,
                                              caches, pos[:, s:s + 1], valid[:, :s + 1], **kw)
            osteps.append(LO.vl_logits(sdf, hid, flag, Q))
        oref = torch.cat(osteps, dim=2)
        fin = torch.isfinite(ref)
        # the same cached run in the reference's own op-by-op bf16 arithmetic: the yardstick for "bf16 noise" on these logits
        sdb = {k: v.to(BF) for k, v in w.items()}
        sgb = t[f"{name}.signal"].to(BF)
        hb, fb, cb = LO.model_step(sdb, t[f"{name}.input_ids"][:, :, :P], t[f"{name}.vision_indices"][:, :P], sgb[:, :P], None,
                                   pos[:, :P], valid[:, :P], **kw)
        bsteps = [LO.vl_logits(sdb, hb, fb, Q)]
        for s in range(P, S):
            hb, fb, cb = LO.model_step(sdb, t[f"{name}.input_ids"][:, :, s:s + 1], t[f"{name}.vision_indices"][:, s:s + 1], None, cb,
                                       pos[:, s:s + 1], valid[:, :s + 1], **kw)
            bsteps.append(LO.vl_logits(sdb, hb, fb, Q))
        theirs = rel_err(torch.cat(bsteps, dim=2).float()[fin], oref[fin])
        # the cached and the uncached path of this model agree (same kernels except the attention / rope entry points)
        fl = LibraForCausalLM.materialize_logits(full).float().cpu()
        keep = torch.ones(S, dtype=torch.bool); keep[info["eoi_forced_steps"]] = False
        a, b = inc[:, :, keep], fl[:, :, keep]
        mfin = torch.isfinite(b)
        assert torch.equal(torch.isfinite(a), mfin)
        e_self = rel_err(a[mfin], b[mfin])
        e_oracle = rel_err(inc[fin], oref[fin])
        e_fix = rel_err(inc[fin], ref[fin])
        report = dict(case=name, cached_vs_uncached=e_self, vs_fp32_oracle=e_oracle, vs_reference_fixture=e_fix, reference_bf16=theirs)
        assert e_self < max(1.5 * theirs, 6e-3), report
        assert e_oracle < max(2.0 * theirs, 6e-3), report            # the forward test's criterion
        assert e_fix < max(3.0 * theirs, 3e-2), report               # vs the reference's fp32-weight run: + weight rounding


def test_libra_model_path_at_seq_4096_vs_oracle():
    """BASELINE configs[4] sequence length (4096 > the reference's 2048-entry RoPE cache, which it extends on demand): the whole
    model path - embeddings, 2 routed layers, heads, CE, backward - at S = 4096 with libra_tiny's width and weights, against
    autograd through the fp32 oracle (loss, final hidden state, gradients of every parameter family)."""
    from libra_amd.libra import LibraConfig, LibraForCausalLM
    from oracle import libra_oracle as LO
    t, meta = load_golden("libra_tiny.safetensors")
    c = dict(meta["cfg"], max_position_embeddings=4096)
    m = LibraForCausalLM(LibraConfig(**c))
    m.load_state_dict(sub(t, "w."), strict=True)
    m = m.to(BF).cuda()
    m.requires_grad_(True)
    V, L, S = c["vocab_size"], c["max_vision_token_length"], 4096
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(3, V, (1, 1, S), generator=g).repeat(2, 1, 1)
    ids[:, 0, 0] = 1
    vi = torch.full((1, S), L, dtype=torch.long)
    for start in (5, 2000):                                          # two images inside the sequence
        ids[0, 0, start:start + L] = torch.cat([torch.tensor([meta["boi"]]), V + torch.randint(0, 16, (L - 2,), generator=g), torch.tensor([meta["eoi"]])])
        ids[1, 0, start:start + L] = torch.cat([torch.tensor([meta["boi"]]), V + torch.randint(0, 16, (L - 2,), generator=g), torch.tensor([meta["eoi"]])])
        vi[0, start:start + L] = torch.arange(L)
    am = torch.ones(1, S, dtype=torch.long); am[0, 3900:] = 0
    sig = torch.zeros(1, S, c["contiguous_signal_size"])
    sig[0, 6:6 + L - 2] = torch.randn(L - 2, c["contiguous_signal_size"], generator=g)
    labels = LO.get_labels(ids, am, [[(5 + L, 6 + L), (2000 + L, 2001 + L)]], boi_token_id=meta["boi"], bos_token_id=1)
    out = m(input_ids=ids.cuda(), attention_mask=am.cuda(), vision_indices=vi.cuda(), contiguous_signal=sig.to(BF).cuda(),
            labels=labels.cuda(), output_hidden_states=True)
    out.loss.backward()
    sdf = {k: v.to(BF).float().requires_grad_(True) for k, v in sub_params(t, "w.").items()}
    kw = dict(layers=c["num_hidden_layers"], heads=c["num_attention_heads"], vocab=V, max_vision_token_length=L, eps=c["rms_norm_eps"],
              max_pos=4096)
    hid, flag = LO.model_forward(sdf, ids, am, vi, sig.to(BF).float(), **kw)
    ref_loss = LO.causal_lm_loss(LO.vl_logits(sdf, hid, flag, 2), labels)
    ref_loss.backward()
    valid = am.bool()
    e_h = rel_err(out.hidden_states[-1].float().cpu()[valid], hid.detach()[valid])
    assert e_h < 2e-2, e_h
    assert abs(float(out.loss) - float(ref_loss)) < 2e-2 * abs(float(ref_loss)), (float(out.loss), float(ref_loss))
    worst = ("", 0.0)
    for name, p in m.named_parameters():
        if name == "vision_hidden_placeholder":
            continue
        ref = sdf[name].grad
        e = rel_err(p.grad.float().cpu(), ref)
        worst = max(worst, (name, e), key=lambda x: x[1])
        assert e < 6e-2, (name, e)
    from helpers import parity_report
    parity_report(f"[configs[4] sequence length, tiny width] S=4096 model path: hidden {e_h:.3e}, loss {float(out.loss):.4f} vs {float(ref_loss):.4f}, "
                  f"worst parameter gradient {worst[1]:.3e} at {worst[0]}")


def test_libra_depth32_vs_reference_fixture_error_growth():
    """Tiny width x FULL depth: the 32-iteration routed layer loop (modeling_libra.py:781-807) against the
    reference's own 32-layer run - every one of the 33 hidden states, the loss, and the reference's autograd gradients; the error
    per depth, ours (HIP, bf16) next to theirs (the oracle executed op by op in bf16 = the reference's arithmetic), goes to the
    parity report.  Gate: at every depth ours <= max(1.5 x theirs, 4e-3)."""
    from helpers import parity_report
    from libra_amd.libra import LibraConfig, LibraForCausalLM
    from oracle import libra_oracle as LO
    from test_oracle_libra_golden import depth32_names_shapes, depth32_state
    t, meta = load_golden("libra_tiny_depth32.safetensors")
    c = meta["cfg"]
    m = LibraForCausalLM(LibraConfig(**c))
    ours_ns = {n: tuple(p.shape) for n, p in m.named_parameters()}
    assert ours_ns == {n: tuple(s) for n, s in depth32_names_shapes(c)}       # the product's parameters = the reference layout
    sd32 = depth32_state(meta, list(ours_ns.items()))
    m.load_state_dict(sd32, strict=False)
    m = m.to(BF).cuda()
    m.requires_grad_(True)
    ids, am, vi = t["in.input_ids"].cuda(), t["in.attention_mask"].cuda(), t["in.vision_indices"].cuda()
    sig, lab = t["in.signal"].to(BF).cuda(), t["in.labels"].cuda()
    out = m(input_ids=ids, attention_mask=am, vision_indices=vi, contiguous_signal=sig, labels=lab, output_hidden_states=True)
    assert len(out.hidden_states) == 33
    out.loss.backward()
    # theirs: the oracle op by op in bf16 (the reference's own arithmetic under torch_dtype=bfloat16)
    sdb = {k: v.to(BF) for k, v in sd32.items()}
    hsb = []
    kw = dict(layers=c["num_hidden_layers"], heads=c["num_attention_heads"], vocab=c["vocab_size"],
              max_vision_token_length=c["max_vision_token_length"], eps=c["rms_norm_eps"], max_pos=c["max_position_embeddings"])
    hidb, _ = LO.model_forward(sdb, t["in.input_ids"], t["in.attention_mask"], t["in.vision_indices"], t["in.signal"].to(BF),
                               hidden_states=hsb, **kw)
    hsb = hsb[:32] + [hidb]
    valid = t["in.attention_mask"].bool()
    ref = t["out.hidden_states"]
    # theirs, pinned: the REFERENCE's own run under model.to(torch.bfloat16) (tests/golden/make_golden_libra32_bf16.py)
    refb = load_golden("libra_tiny_depth32_bf16.safetensors")[0]["out.hidden_states"]
    lines, worst = [], 0.0
    for l in range(33):
        e_o = rel_err(out.hidden_states[l].float().cpu()[valid], ref[l][valid])
        e_t = rel_err(hsb[l].float()[valid], ref[l][valid])
        e_r = rel_err(refb[l].float()[valid], ref[l][valid])
        worst = max(worst, e_o)
        if l % 4 == 0 or l >= 31:
            lines.append(f"{l}:{e_o:.2e}/{e_r:.2e}/{e_t:.2e}")
        assert e_o < max(1.5 * e_r, 4e-3), (l, e_o, e_r, e_t)
    loss_ref = float(t["out.loss"])
    assert abs(float(out.loss) - loss_ref) < 3e-2 * abs(loss_ref), (float(out.loss), loss_ref)
    # theirs for the gradients: autograd through the bf16 oracle (what the reference's bf16 training run computes)
    sdg = {k: v.to(BF).requires_grad_(True) for k, v in sd32.items()}
    hg, fg = LO.model_forward(sdg, t["in.input_ids"], t["in.attention_mask"], t["in.vision_indices"], t["in.signal"].to(BF), **kw)
    LO.causal_lm_loss(LO.vl_logits(sdg, hg, fg, c["vision_codebook_num"]).float(), t["in.labels"]).backward()
    gw, gname, gt, n = 0.0, "", 0.0, 0
    params = dict(m.named_parameters())
    for k, g in sub(t, "grad.").items():
        if k == "vision_hidden_placeholder":
            continue
        gmax = float(g.float().abs().max())
        if gmax < 1e-6:
            continue
        e = rel_err(params[k].grad.float().cpu(), g.float())
        et = rel_err(sdg[k].grad.float(), g.float())
        gt = max(gt, et)
        if e > gw:
            gw, gname = e, k
        # (through 32 bf16 layers single gradients are noisy on both sides - first visit: ours 6.6e-2 where theirs was 3.2e-2 on a
        #  layer-0 bridge matrix, and the other way round elsewhere; the worst of each side is what the report carries)
        assert e < max(3.0 * et, 1e-1), (k, e, et)
        n += 1
    lines.append(f"| worst gradient ours {gw:.2e} theirs {gt:.2e}")
    parity_report(f"[a20 depth 32, tiny width, vs the reference's own 32-layer run] hidden-state rel err by depth ours / theirs (the reference "
                  f"itself in bf16, fixture) / the bf16 oracle: {' '.join(lines)}; loss {float(out.loss):.5f} vs {loss_ref:.5f}; worst of {n} reference gradients "
                  f"{gw:.2e} ({gname})")
    assert n > 400, n


def test_libra_depth32_error_trend_over_weight_seeds():
    """Round-3 review item 8: on the one 32-layer fixture ours crossed ABOVE the reference's bf16 arithmetic in layers 28-32 (2.7e-2
    vs 2.3e-2).  Is that a rounding point of ours or the draw?  The same inputs, FIVE other weight draws (the fixture's CRC-seeded
    generator with other seeds), each against the fp32 oracle (= the reference's fp32 run to 2e-5, test_oracle_libra_golden.py):
    error of ours and of the oracle run op by op in bf16 at depths 24 / 28 / 31 / 32.  Gate: over the draws the mean ratio
    ours / theirs at depth >= 28 stays <= 1.25 and no single draw exceeds 1.6; the table goes to the parity report."""
    import os
    import sys
    from helpers import parity_report
    from libra_amd.libra import LibraConfig, LibraForCausalLM
    from oracle import libra_oracle as LO
    from test_oracle_libra_golden import depth32_names_shapes
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from seeded_weights import seeded_state
    t, meta = load_golden("libra_tiny_depth32.safetensors")
    c = meta["cfg"]
    kw = dict(layers=c["num_hidden_layers"], heads=c["num_attention_heads"], vocab=c["vocab_size"],
              max_vision_token_length=c["max_vision_token_length"], eps=c["rms_norm_eps"], max_pos=c["max_position_embeddings"])
    ids, am, vi = t["in.input_ids"], t["in.attention_mask"], t["in.vision_indices"]
    sig = t["in.signal"].to(BF)
    valid = am.bool()
    depths = (24, 28, 31, 32)
    rows, ratios = [], []
    for seed in (101, 202, 303, 404, 505):
        sd = seeded_state(depth32_names_shapes(c), seed)
        m = LibraForCausalLM(LibraConfig(**c))
        m.load_state_dict(sd, strict=False)
        m = m.to(BF).cuda()
        with torch.no_grad():
            out = m(input_ids=ids.cuda(), attention_mask=am.cuda(), vision_indices=vi.cuda(), contiguous_signal=sig.cuda(),
                    output_hidden_states=True)
            ours = [h.float().cpu() for h in out.hidden_states]
            ref, hs32 = [], []
            hid32, _ = LO.model_forward({k: v.float() for k, v in sd.items()}, ids, am, vi, sig.float(), hidden_states=hs32, **kw)
            ref = hs32[:32] + [hid32]
            hsb = []
            hidb, _ = LO.model_forward({k: v.to(BF) for k, v in sd.items()}, ids, am, vi, sig, hidden_states=hsb, **kw)
            theirs = [h.float() for h in hsb[:32] + [hidb]]
        cell = []
        for l in depths:
            e_o, e_t = rel_err(ours[l][valid], ref[l][valid]), rel_err(theirs[l][valid], ref[l][valid])
            cell.append(f"{l}:{e_o:.2e}/{e_t:.2e}")
            if l >= 28:
                ratios.append(e_o / e_t)
        rows.append(f"seed {seed} " + " ".join(cell))
        del m, out
        torch.cuda.empty_cache()
    mean = sum(ratios) / len(ratios)
    parity_report("[a20 depth 32, five other weight draws] hidden-state rel err ours/theirs(bf16 oracle) vs the fp32 oracle: "
                  + "; ".join(rows) + f"; ours/theirs at depth >= 28: mean {mean:.2f}, min {min(ratios):.2f}, max {max(ratios):.2f}")
    assert mean <= 1.25 and max(ratios) <= 1.6, (mean, ratios)
