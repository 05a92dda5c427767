"""The drop-in boundary under the reference's own recipe: gradient checkpointing,
`.logits`, optimizers that write through `.data`, the LibraTokenizer module on device tensors, and the
LibraTrainWrapper.forward call contract driven by a synthetic `samples` dict against the committed fixtures."""
import pytest
import torch

from helpers import (FakeImageTokenizer, FakeTextTokenizer, load_golden, rel_err, sub, torch_adamw_update,
                     word_level_tokenizer)

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _tiny(train=True):
    from libra_amd.libra import LibraConfig, LibraForCausalLM
    t, meta = load_golden("libra_tiny.safetensors")
    m = LibraForCausalLM(LibraConfig(**meta["cfg"]))
    m.load_state_dict(sub(t, "w."), strict=True)
    m = m.to(BF).cuda()
    m.train(train)
    kw = dict(input_ids=t["in.input_ids"].cuda(), attention_mask=t["in.attention_mask"].cuda(),
              vision_indices=t["in.vision_indices"].cuda(), contiguous_signal=t["in.signal"].to(BF).cuda(),
              labels=t["in.labels"].cuda())
    return m, kw, t, meta


def test_gradient_checkpointing_recompute_is_bit_identical():
    """libra_pretrain.yaml:120 `gradient_checkpointing: True` -> HF calls gradient_checkpointing_enable(); the engine keeps
    only each layer's input and re-runs the layer in backward: loss and every gradient equal the non-recompute run bit for bit."""
    m, kw, _, _ = _tiny()
    out = m(**kw)
    out.loss.backward()
    plain = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    saved_plain = out._engine_out["saved"]
    m.zero_grad(set_to_none=True)
    m.gradient_checkpointing_enable()
    assert m.model.gradient_checkpointing
    out2 = m(**kw)
    sv = out2._engine_out["saved"]
    assert sv["recompute"] and all(set(l) == {"x"} for l in sv["layers"])          # only the layer inputs are kept
    out2.loss.backward()
    assert torch.equal(out.loss, out2.loss)
    for n, p in m.named_parameters():
        if n in plain:
            assert torch.equal(p.grad, plain[n]), n
    m.eval()                                                                     # :787 `self.gradient_checkpointing and self.training`
    with torch.no_grad():
        assert m(**kw)._engine_out["saved"] is None


def test_inputs_embeds_runs_the_decoder_on_given_embeddings():
    """LibraModel.forward's `inputs_embeds` branch (modeling_libra.py:703-716, :748-754): no table lookup and no signal processing -
    the decoder runs on the given [B, S, H].  Fed with the embeddings the token path itself computes (hidden_states[0]), loss,
    logits and every decoder gradient are the token path's bit for bit; the gradient w.r.t. the embeddings, scattered by token id,
    IS the text embedding table's gradient of the token path; both / neither input raise as upstream."""
    m, kw, t, _ = _tiny()
    out = m(**kw, output_hidden_states=True)
    out.loss.backward()
    plain = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    logits = out.logits
    m.zero_grad(set_to_none=True)
    E = out.hidden_states[0].detach().clone().requires_grad_(True)
    kw2 = {k: v for k, v in kw.items() if k not in ("input_ids", "contiguous_signal")}
    out2 = m(inputs_embeds=E, **kw2)
    assert torch.equal(out2.loss, out.loss) and torch.equal(out2.logits, logits)
    out2.loss.backward()
    emb_stage = ("embed_tokens", "vision_embed_tokens", "vision_contiguous_signal_processor", "vision_signal_norm", "vision_position_embedding")
    checked = 0
    for n, p in m.named_parameters():
        if n in plain and not any(k in n for k in emb_stage):
            assert torch.equal(p.grad, plain[n]), n
            checked += 1
    assert checked > 20
    assert E.grad is not None and E.grad.shape == E.shape and bool(torch.isfinite(E.grad.float()).all())
    ids0 = kw["input_ids"][0].reshape(-1)
    lang = (kw["vision_indices"].reshape(-1) >= m.config.max_vision_token_length)
    acc = torch.zeros(m.model.embed_tokens.weight.shape, dtype=torch.float32, device="cuda")
    acc.index_add_(0, ids0[lang], E.grad.reshape(-1, E.shape[-1])[lang].float())
    assert torch.equal(acc.to(BF), plain["model.embed_tokens.weight"])
    m.eval()
    with torch.no_grad():
        ev = m(inputs_embeds=E.detach(), **{k: v for k, v in kw2.items() if k != "labels"})
    assert ev.loss is None and torch.equal(ev.logits, logits)
    with pytest.raises(ValueError, match="both"):
        m(inputs_embeds=E.detach(), **kw)
    with pytest.raises(ValueError, match="either"):
        m(**kw2)
    with pytest.raises(NotImplementedError):
        m(inputs_embeds=E.detach(), use_cache=True, **{k: v for k, v in kw2.items() if k != "labels"})


def test_logits_lazy_in_training_and_eager_in_eval():
    """`.logits` [Q,B,S,V+514] (modeling_libra.py:1180-1188): eager without autograd (Trainer.prediction_step reads it from
    items()), built on first access in a training step; same values either way and equal to the reference fixture's pattern."""
    m, kw, t, meta = _tiny()
    out = m(**kw)
    assert "logits" not in out.keys() and out._engine_out["saved"] is not None
    lg = out.logits
    assert lg is not None and tuple(lg.shape) == tuple(t["out.logits"].shape) and "logits" in out.keys()
    assert out["logits"] is lg and out.logits is lg
    out_b = m(**kw)
    assert out_b["logits"].shape == lg.shape                                     # dict-style first access works too
    m.eval()
    with torch.no_grad():
        ev = m(**kw)
    assert "logits" in ev.keys() and torch.equal(ev.logits, lg)
    assert torch.equal(torch.isfinite(lg.float().cpu()), torch.isfinite(t["out.logits"]))
    assert ev.to_tuple()[1] is ev.logits


def test_weights_updated_through_dot_data_are_seen():
    """optimizers that write through `.data` (DeepSpeed ZeRO, master-weight optimizers) bump neither
    `_version` nor `data_ptr`; the fused operand copies must follow anyway (trainable ones are refreshed every forward)."""
    from libra_amd.libra import LibraConfig, LibraForCausalLM, apply_freeze_policy
    m, kw, t, meta = _tiny()
    apply_freeze_policy(m, frozen_language=True)
    l0 = float(m(**kw).loss)
    g = torch.Generator().manual_seed(1)
    names = ["model.layers.0.self_attn.vision_q_proj.weight_A", "model.layers.1.mlp.vision_gate_proj.weight_A",
             "model.layers.0.self_attn.vision_k_bridge_on_language.weight_A",
             "model.layers.1.self_attn.vision_v_bridge_on_vision.weight_B"]
    params = dict(m.named_parameters())
    for n in names:
        p = params[n]
        v0, ptr0 = p._version, p.data_ptr()
        p.data.copy_((p.data.float() + 0.5 * torch.randn(p.shape, generator=g).cuda()).to(BF))
        assert p._version == v0 and p.data_ptr() == ptr0                       # the update is invisible to version counters
    l1 = float(m(**kw).loss)
    fresh = LibraForCausalLM(LibraConfig(**meta["cfg"])).to(BF).cuda()
    fresh.load_state_dict(m.state_dict(), strict=True)
    fresh.train()
    l_fresh = float(fresh(**kw).loss)
    assert l1 == l_fresh and l1 != l0, (l0, l1, l_fresh)
    # flat-buffer re-binding (p.data = view of a flat buffer), as ZeRO does
    p = params[names[0]]
    flat = torch.zeros(p.numel() + 64, dtype=BF, device="cuda")
    flat[64:].copy_((p.data.float() * 0.5).reshape(-1).to(BF))
    p.data = flat[64:].view(p.shape)
    fresh.load_state_dict(m.state_dict(), strict=True)
    assert float(m(**kw).loss) == float(fresh(**kw).loss)
    # a FROZEN weight written through .data needs the documented invalidate_packed()
    q = params["model.layers.0.self_attn.q_proj.weight"]
    q.data.mul_(0.5)
    m.invalidate_packed()
    fresh.load_state_dict(m.state_dict(), strict=True)
    assert float(m(**kw).loss) == float(fresh(**kw).loss)


def test_bad_ids_raise_instead_of_reading_out_of_bounds():
    """ids outside the embedding tables (nn.Embedding device-asserts upstream), wrong dtypes."""
    m, kw, t, meta = _tiny(train=False)
    V, Vv = meta["cfg"]["vocab_size"], meta["cfg"]["vision_vocab_size"]
    with torch.no_grad():
        bad = dict(kw); ids = kw["input_ids"].clone(); ids[1, 0, 2] = V + Vv; bad["input_ids"] = ids
        with pytest.raises(IndexError):
            m(**bad)
        bad = dict(kw); ids = kw["input_ids"].clone(); ids[1, 0, 2] = 5; bad["input_ids"] = ids       # text id in codebook 1
        with pytest.raises(IndexError):
            m(**bad)
        bad = dict(kw); bad["input_ids"] = kw["input_ids"].int()
        with pytest.raises(TypeError):
            m(**bad)
        bad = dict(kw); ids = kw["input_ids"].clone(); ids[:, 0, 8] = V + 3; bad["input_ids"] = ids   # flag mismatch
        with pytest.raises(AssertionError):
            m(**bad)


def test_adamw_kernel_vs_torch():
    """libra_adamw_step vs torch.optim.AdamW arithmetic on an fp32 master (several steps, odd length, weight decay on/off)."""
    from libra_amd import kernels as K
    g = torch.Generator().manual_seed(0)
    for n, wd in ((1 << 16, 0.01), (12345, 0.0), (7, 0.1)):
        master = torch.randn(n, generator=g).cuda()
        m, v = torch.zeros(n).cuda(), torch.zeros(n).cuda()
        param = torch.empty(n, dtype=BF).cuda()
        rm, rmm, rv, rp = master.clone(), m.clone(), v.clone(), param.clone()
        for step in range(1, 4):
            grad = (torch.randn(n, generator=g) * 0.1).to(BF).cuda()
            kw = dict(lr=1e-2, beta1=0.9, beta2=0.99, eps=1e-8, weight_decay=wd, bias_corr1=1 - 0.9 ** step,
                      bias_corr2=1 - 0.99 ** step)
            K.adamw_step(master, m, v, grad, param, **kw)
            torch_adamw_update(rm, rmm, rv, grad, rp, **kw)
        assert rel_err(master.cpu(), rm.cpu()) < 1e-5 and rel_err(m.cpu(), rmm.cpu()) < 1e-5 and rel_err(v.cpu(), rv.cpu()) < 1e-5
        assert float((param.float() - rp.float()).abs().max()) <= float(rp.float().abs().max()) * 2 ** -7
        assert torch.equal(param, master.to(BF))
    with pytest.raises(ValueError):
        K.adamw_step(torch.zeros(4), torch.zeros(4), torch.zeros(4), torch.zeros(4, dtype=BF), torch.zeros(4, dtype=BF),
                     lr=1.0, beta1=0.9, beta2=0.99, eps=1e-8, weight_decay=0.0, bias_corr1=0.1, bias_corr2=0.01)


def test_sumsq_and_clipped_adamw_kernel_vs_torch():
    """libra_sumsq_bf16 (deterministic, accumulate flag, ragged tail) and the device-side clipping coefficient of
    libra_adamw_step vs torch.nn.utils.clip_grad_norm_ + AdamW arithmetic."""
    from helpers import torch_sumsq
    from libra_amd import kernels as K
    g = torch.Generator().manual_seed(1)
    for n in (8, 12345, (1 << 20) + 24):
        x = (torch.randn(n, generator=g) * 0.3).to(BF).cuda()
        out = torch.full((1,), 5.0, device="cuda")
        K.sumsq(x, out)
        ref = float(x.double().pow(2).sum())
        assert abs(float(out) - ref) <= 1e-5 * ref
        first = float(out)
        K.sumsq(x, out, accumulate=True)
        assert abs(float(out) - 2 * ref) <= 2e-5 * ref
        out2 = torch.zeros(1, device="cuda")
        K.sumsq(x, out2)
        assert float(out2) == first                                   # bit-identical rerun
    n = 4099
    for scale, clips in ((1.0, True), (1e-3, False)):
        master = torch.randn(n, generator=g).cuda()
        m, v = torch.zeros(n).cuda(), torch.zeros(n).cuda()
        param = torch.empty(n, dtype=BF).cuda()
        rm, rmm, rv, rp = master.clone(), m.clone(), v.clone(), param.clone()
        grad = (torch.randn(n, generator=g) * scale).to(BF).cuda()
        nsq = torch.zeros(1, device="cuda")
        K.sumsq(grad, nsq)
        assert (float(nsq.sqrt()) > 1.0) == clips
        kw = dict(lr=1e-2, beta1=0.9, beta2=0.99, eps=1e-8, weight_decay=0.01, bias_corr1=0.1, bias_corr2=0.01)
        K.adamw_step(master, m, v, grad, param, grad_norm_sq=nsq, max_grad_norm=1.0, **kw)
        torch_adamw_update(rm, rmm, rv, grad, rp, grad_norm_sq=nsq.cpu(), max_grad_norm=1.0, **kw)
        assert rel_err(master.cpu(), rm.cpu()) < 1e-5 and rel_err(m.cpu(), rmm.cpu()) < 1e-5 and rel_err(v.cpu(), rv.cpu()) < 2e-5
    with pytest.raises(ValueError):                                    # a norm pointer without a positive max_grad_norm
        K.adamw_step(master, m, v, grad, param, grad_norm_sq=nsq, max_grad_norm=0.0, **kw)


def test_libra_tokenizer_module_on_device_matches_reference_fixture():
    """a11 on DEVICE tensors: LibraTokenizer.forward (nn.Module surface) with the fixture's text / image ids living on the GPU
    reproduces the reference's own LibraTokenizer.forward outputs bit for bit."""
    from libra_amd.libra import LibraTokenizer
    t, meta = load_golden("libra_tokenizer_assembly.safetensors")
    tok = LibraTokenizer(text_tokenizer=FakeTextTokenizer(t["in.text_ids"], t["in.attention_mask"], 96, meta["img_ph"],
                                                          meta["img_gen"], meta["max_length"]),
                         image_tokenizer=FakeImageTokenizer(t["in.image_ids"], t["in.encoder_feat"].to(BF), meta["boi"],
                                                            meta["L"], meta["Q"], device="cuda"))
    assert tok.device.type == "cuda" and tok.dtype == BF
    tok = tok.cuda()
    samples = {"language": list("abc"), "vision": [torch.zeros(3, 8, 8)] * 3, "contiguous_ignore_sign": meta["ignore"]}
    out = tok(samples, padding="longest", truncation=True, max_length=meta["max_length"])
    assert all(v.is_cuda for v in out.values())
    assert torch.equal(out["input_ids"].cpu(), t["out.input_ids"])
    assert torch.equal(out["attention_mask"].cpu(), t["out.attention_mask"])
    assert torch.equal(out["vision_indices"].cpu(), t["out.vision_indices"])
    assert torch.equal(out["coninous_signal"].cpu(), t["out.signal"].to(BF))


def test_train_wrapper_call_contract_vs_reference_fixture():
    """LibraTrainWrapper.forward(samples) (modeling_libra.py:1414-1433): synthetic `samples` dict -> tokenizer assembly ->
    get_labels -> LibraForCausalLM on device.  The stub tokenizers hand back the token ids / encoder features of the
    `libra_tiny` fixture, so the assembled input_ids / vision_indices / signal / labels must equal the tensors the reference
    model was run on, and the loss the reference's own loss."""
    from libra_amd.common.registry import registry
    from libra_amd.libra import LibraConfig, LibraForCausalLM, LibraTokenizer
    t, meta = load_golden("libra_tiny.safetensors")
    c = meta["cfg"]
    V, L, Q = c["vocab_size"], c["max_vision_token_length"], c["vision_codebook_num"]
    ids, am, vi, sig = t["in.input_ids"], t["in.attention_mask"], t["in.vision_indices"], t["in.signal"]
    PH, GEN = 1000, 1001                                              # placeholder ids outside every table
    text = ids[0].clone()
    text[vi < L] = PH
    spans_img = [(1, 7), (4, 10)]                                     # where make_golden_libra.py put the two images
    image_ids = torch.stack([torch.stack([ids[q, b, s:e] for b, (s, e) in enumerate(spans_img)]) for q in range(Q)])
    feat = torch.stack([sig[b, s + 1:e - 1] for b, (s, e) in enumerate(spans_img)])
    m = LibraForCausalLM(LibraConfig(**c))
    m.load_state_dict(sub(t, "w."), strict=True)
    m = m.to(BF).cuda()
    tok = LibraTokenizer(text_tokenizer=FakeTextTokenizer(text, am, V, PH, GEN, 64),
                         image_tokenizer=FakeImageTokenizer(image_ids, feat.to(BF), meta["boi"], L, Q, device="cuda")).cuda()
    cls = registry.get_model_class("libra_train_wrapper")
    emb_before = m.get_input_embeddings().weight.detach().clone()
    wrapper = cls({"pretrained": None, "model_kwargs": {"frozen_language": True}}, module=m, tokenizer=tok)
    # change_pad_token_to_eos (:1383-1388) ran; undo it so the loss is comparable with the fixture's untouched embedding
    assert torch.equal(m.get_input_embeddings().weight[0], emb_before[2])
    m.get_input_embeddings().weight.data.copy_(emb_before)
    assert all(("vision" in n) == p.requires_grad for n, p in m.named_parameters())
    samples = {"vision": [torch.zeros(3, 8, 8)] * 2, "language": ["x", "y"], "contiguous_ignore_sign": [False, False],
               "label_mask_position_map": [[tuple(s) for s in sp] for sp in meta["spans"]]}
    inputs = wrapper.tokenizer(samples, return_tensors="pt", padding="longest", max_length=64, truncation=True)
    assert torch.equal(inputs["input_ids"].cpu(), ids) and torch.equal(inputs["vision_indices"].cpu(), vi)
    assert torch.equal(inputs["coninous_signal"].cpu(), sig.to(BF))
    assert torch.equal(wrapper.get_labels(inputs, samples["label_mask_position_map"]).cpu(), t["in.labels"])
    wrapper.train()
    out = wrapper(samples)
    assert abs(float(out.loss) - float(t["out.loss"])) < 5e-2 * abs(float(t["out.loss"]))
    out.loss.backward()
    got = {n for n, p in m.named_parameters() if p.grad is not None}
    assert got == {n for n, p in m.named_parameters() if p.requires_grad and n != "vision_hidden_placeholder"}
    assert tuple(out.logits.shape) == tuple(t["out.logits"].shape)


def test_train_py_shaped_smoke_with_a_synthetic_collater():
    """what train.py does, end to end on device with REAL parts - a HF fast tokenizer, the CLIP ViT -> VQ
    image tokenizer on the HIP kernels, the routed decoder, gradient checkpointing on (the recipe), the reference's freeze
    policy and three fused-AdamW steps over the flat gradient buckets: the loss is finite and goes down."""
    from transformers import CLIPVisionConfig
    from libra_amd import decoder_engine as DE
    from libra_amd import dp
    from libra_amd.clip import CLIPVisionModel
    from libra_amd.libra import ImageTokenizer, LibraConfig, LibraForCausalLM, LibraTokenizer, LibraTrainWrapper
    from oracle import vit_oracle as VO, vq_oracle as QO
    torch.manual_seed(0)
    vcfg = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=56, patch_size=14)
    clip = CLIPVisionModel(CLIPVisionConfig(**vcfg))
    clip.load_state_dict({k: v for k, v in VO.random_vit_state_dict(hidden=128, inter=256, layers=3, patch=14, image=56).items()},
                         strict=False)
    clip = clip.to(BF).cuda()
    tt = word_level_tokenizer("a photo of cat dog on the grass some text follows here".split(), model_max_length=64)
    L = 18                                                            # 4x4 patches + BOI + EOI
    tcfg = {"params": {"ddconfig": {"encoder_name": "clip_tiny", "select_layer": [-2, -3]}, "embed_dim": 32,
                       "codebook_size": 512, "num_codebook": 2}, "max_vision_token_length": L}
    it = ImageTokenizer(tcfg, token_offset=tt.vocab_size, vision_model=clip)
    it.model.load_state_dict(QO.random_vq_state_dict(c_feat=256, embed_dim=32), strict=False)
    it = it.to(BF).cuda()
    tok = LibraTokenizer(text_tokenizer=tt, image_tokenizer=it).cuda()
    cfg = LibraConfig(vocab_size=tt.vocab_size, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                      max_position_embeddings=64, vision_vocab_size=514, max_vision_token_length=L, contiguous_signal_size=256,
                      image_feature_resolution=4)
    lm = LibraForCausalLM(cfg)
    with torch.no_grad():
        for n, p in lm.named_parameters():
            if "bridge" in n and n.endswith("weight_B"):
                p.normal_(0, 0.02)
    lm = lm.to(BF).cuda()
    model = LibraTrainWrapper({"pretrained": None, "model_kwargs": {"frozen_language": True}}, module=lm, tokenizer=tok)
    model.train()
    model.module.gradient_checkpointing_enable()
    ph = " ".join(["<img_ph>"] * L)
    g = torch.Generator().manual_seed(1)
    samples = {"vision": [torch.randn(3, 56, 56, generator=g) for _ in range(3)],
               "language": [f"a photo of cat {ph} some text follows", f"{ph} dog on the grass", f"the {ph} text here follows a cat"],
               "contiguous_ignore_sign": [False, True, False],
               "label_mask_position_map": [[(5 + L, 6 + L)], [(1 + L, 2 + L)], [(2 + L, 3 + L)]]}
    named = [(n, p) for n, p in lm.named_parameters() if p.requires_grad and n != "vision_hidden_placeholder"]
    st = dp.GradBuckets(named, bucket_bytes=1 << 16, group_fn=lambda n: DE.emit_group(n, 2))
    opt = dp.FlatAdamW(st, named, lr=2e-3, betas=(0.9, 0.99), weight_decay=0.01)
    losses = []
    for _ in range(4):
        out = model(samples)
        losses.append(float(out.loss))
        with st.capture():
            out.loss.backward()
        st.finish_into(named)
        opt.step()
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0], losses
    assert lm.model.gradient_checkpointing
