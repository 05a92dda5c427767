"""Full-size parity ("Full-size backward has no oracle check"): backward of a full-width decoder layer and of
ViT-L/14@336 against autograd through the fp32 oracle, VQ index flips at the BASELINE batch against the float64 oracle.
Every test appends its measured `ours / theirs` numbers to gpurun_out/parity_report.txt (committed as profiles/r02_parity.txt).

Tolerances (SURVEY §8d parity gates): activations / dx  max-norm rel <= max(2 x the reference's own op-by-op bf16 error,
2e-3); weight gradients rel <= max(2 x theirs, 1e-2) (bf16 operands, fp32 accumulation); integer outputs bit-exact up to
the margin rule stated in the test."""
import pytest
import torch

from helpers import parity_report, rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def test_full_width_decoder_layer_backward_vs_oracle_autograd():
    """(i) H=4096, I=11008, 32x128 heads, rank-8 bridges, B=2, S=256, non-contiguous padding: dx and EVERY weight gradient
    (dense text, low-rank vision A/B, all eight bridge matrices, four norm weights)."""
    from libra_amd import decoder_engine as DE
    from oracle import libra_oracle as LO
    H, heads, L = 4096, 32, 70
    B, S = 2, 256
    sd = LO.random_layer_state_dict(seed=5)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, S, H, generator=g).to(BF)
    ct = (torch.randn(B, S, H, generator=g) * 0.05).to(BF)
    vi = torch.full((B, S), L, dtype=torch.long)
    vi[0, 3:3 + L] = torch.arange(L); vi[1, 100:100 + L] = torch.arange(L)
    am = torch.ones(B, S, dtype=torch.long); am[1, 200:] = 0
    ct[1, 200:] = 0                                        # padded positions carry no loss
    d = DE.DecDims(hidden=H, inter=11008, layers=1, heads=heads, vocab=32000, vision_vocab=514, codebooks=2, max_vision_len=L,
                   signal=2048)
    dsd = {k: v.cuda() for k, v in sd.items()}
    dsd["model.embed_tokens.weight"] = torch.zeros(8, H, dtype=BF, device="cuda")
    pk = DE.pack(dsd, d)
    flag, li, vidx, lens, _ = DE.route(vi.cuda(), am.cuda(), d)
    cos, sin = DE.rope_tables(128, 2048, "cuda")
    sv = {}
    y = DE.layer_forward(dsd, pk[0], 0, d, x.view(B * S, H).cuda(), flag, li, vidx, lens, cos, sin, B, S, sv)
    grads = {}
    dx = DE.layer_backward(dsd, pk[0], 0, d, sv, ct.view(B * S, H).cuda().contiguous(), flag, li, vidx, lens, cos, sin, B, S,
                           grads, lambda n: True)
    f = vi < L
    cosr, sinr = LO.rope_tables(128, 2048)
    pos = torch.arange(S).unsqueeze(0).expand(B, S)

    def oracle(dtype):
        sdo = {k: v.to(dtype).requires_grad_(True) for k, v in sd.items()}
        xin = x.to(dtype).requires_grad_(True)
        yo = LO.decoder_layer(sdo, 0, xin, f, LO.additive_mask(am, S, dtype), pos, heads, 1e-6, cosr.to(dtype), sinr.to(dtype))
        (yo.float() * ct.float()).sum().backward()
        return yo.detach(), xin.grad, {k: v.grad for k, v in sdo.items()}
    y32, dx32, g32 = oracle(torch.float32)
    y16, dx16, g16 = oracle(BF)
    valid = am.bool()
    e_y, t_y = rel_err(y.view(B, S, H).float().cpu()[valid], y32[valid]), rel_err(y16.float()[valid], y32[valid])
    e_dx, t_dx = rel_err(dx.view(B, S, H).float().cpu()[valid], dx32[valid]), rel_err(dx16.float()[valid], dx32[valid])
    parity_report(f"[configs[2] layer, H=4096 S=256] forward y: ours {e_y:.3e} theirs(bf16 op-by-op) {t_y:.3e}; "
                  f"dx: ours {e_dx:.3e} theirs {t_dx:.3e}")
    assert e_y < max(2 * t_y, 2e-3) and e_dx < max(2 * t_dx, 2e-3), (e_y, t_y, e_dx, t_dx)
    worst = ("", 0.0, 0.0)
    assert set(g32) <= set(grads), set(g32) - set(grads)
    for name, ref in g32.items():
        ours, theirs = rel_err(grads[name].float().cpu().view(ref.shape), ref), rel_err(g16[name].float(), ref)
        if ours > worst[1]:
            worst = (name, ours, theirs)
        assert ours < max(2 * theirs, 1e-2), (name, ours, theirs)
    parity_report(f"[configs[2] layer, H=4096 S=256] {len(g32)} weight gradients incl. rank-8 bridges: worst ours {worst[1]:.3e} "
                  f"(theirs {worst[2]:.3e}) at {worst[0]}")


def test_vit_l_336_backward_vs_oracle_autograd():
    """(ii) ViT-L/14@336, B=1: pixel gradient and weight gradients (first / middle / last reached layer, embeddings,
    LayerNorms, biases) of a feature cotangent on cat(hs[-2], hs[-3])[:, 1:]."""
    from transformers import CLIPVisionConfig
    from libra_amd.clip import CLIPVisionModel
    from oracle import vit_oracle as VO
    cfg = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336,
               patch_size=14)
    sd = {k: v.to(BF) for k, v in VO.random_vit_state_dict(hidden=1024, inter=4096, layers=24, patch=14, image=336, seed=42).items()}
    m = CLIPVisionModel(CLIPVisionConfig(**cfg))
    m.load_state_dict(sd, strict=False)
    m = m.to(BF).cuda()
    m.requires_grad_(True)
    g = torch.Generator().manual_seed(42)
    x = torch.randn(1, 3, 336, 336, generator=g).to(BF)
    ct = torch.randn(1, 576, 2048, generator=g).to(BF)
    xg = x.cuda().requires_grad_(True)
    hs = m(xg, output_hidden_states=True).hidden_states
    (torch.cat([hs[-2], hs[-3]], -1)[:, 1:].float() * ct.cuda().float()).sum().backward()

    def oracle(dtype):
        sdo = {k: v.to(dtype).requires_grad_(True) for k, v in sd.items()}
        xin = x.to(dtype).requires_grad_(True)
        ref = VO.vit_hidden_states(sdo, xin, patch=14, heads=16, layers=24)
        (VO.feature_select(ref, [-2, -3], square=False).float() * ct.float()).sum().backward()
        return xin.grad, {k: v.grad for k, v in sdo.items() if v.grad is not None}
    dx32, g32 = oracle(torch.float32)
    dx16, g16 = oracle(BF)
    e, t = rel_err(xg.grad.float().cpu(), dx32), rel_err(dx16.float(), dx32)
    parity_report(f"[configs[1] ViT-L/14@336 B=1 backward] pixel grad: ours {e:.3e} theirs(bf16 op-by-op) {t:.3e}")
    assert e < max(2 * t, 1e-2), (e, t)
    P = "vision_model."
    pick = [P + "embeddings.patch_embedding.weight", P + "embeddings.position_embedding.weight", P + "embeddings.class_embedding",
            P + "pre_layrnorm.weight"]
    for i in (0, 11, 22):
        pre = f"{P}encoder.layers.{i}."
        pick += [pre + n for n in ("self_attn.q_proj.weight", "self_attn.q_proj.bias", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                                   "self_attn.out_proj.weight", "self_attn.out_proj.bias", "layer_norm1.weight", "layer_norm2.bias",
                                   "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")]
    named = dict(m.named_parameters())
    worst = ("", 0.0, 0.0)
    for name in pick:
        ref = g32[name]
        ours, theirs = rel_err(named[name].grad.float().cpu().view(ref.shape), ref), rel_err(g16[name].float(), ref)
        if ours > worst[1]:
            worst = (name, ours, theirs)
        assert ours < max(2 * theirs, 1e-2), (name, ours, theirs)
    # d(loss)/d(k_proj.bias) is identically zero (a constant added to every key shifts each softmax row uniformly): both sides
    # must leave only rounding noise there, far below the q bias gradient
    kb, qb = f"{P}encoder.layers.11.self_attn.k_proj.bias", f"{P}encoder.layers.11.self_attn.q_proj.bias"
    assert float(named[kb].grad.float().abs().max()) < 1e-2 * float(named[qb].grad.float().abs().max())
    # the last encoder layer is never reached by select_layer=[-2,-3]: no gradient, as in the reference's autograd
    assert named[f"{P}encoder.layers.23.mlp.fc1.weight"].grad is None and f"{P}encoder.layers.23.mlp.fc1.weight" not in g32
    parity_report(f"[configs[1] ViT-L/14@336 B=1 backward] {len(pick)} sampled weight gradients: worst ours {worst[1]:.3e} "
                  f"(theirs {worst[2]:.3e}) at {worst[0]}")


def test_vq_indices_at_baseline_batch_vs_fp64_oracle():
    """(iii) VQ encode at B=32, E=512 (BASELINE configs[1] batch): 18 432 tokens x 18 sign bits against the float64 oracle,
    stage by stage at the reference's two rounding points (h -> bf16 after quant_conv, x -> bf16 after project_in):
      A. h (kernel GEMM, fp32 accumulate, bf16 store) vs bf16(h64): a differing element must be the ADJACENT bf16 value and
         h64 must sit within fp32-accumulation noise (1e-5 relative to the row scale) of the rounding boundary;
      B. sign bits vs the float64 projection of the kernel's OWN bf16 h: a differing bit needs |x64| < 1e-5 (the margin rule of
         test_lfq_encode);
      and the end-to-end flip count vs the pure float64 chain with its largest margin is reported (and bounded)."""
    from transformers import CLIPVisionConfig
    from libra_amd.clip import CLIPVisionModel
    from libra_amd.libra import ImageTokenizer
    from oracle import vit_oracle as VO, vq_oracle as QO
    cfg = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336,
               patch_size=14)
    sdv = {k: v.to(BF) for k, v in VO.random_vit_state_dict(hidden=1024, inter=4096, layers=24, patch=14, image=336, seed=42).items()}
    m = CLIPVisionModel(CLIPVisionConfig(**cfg))
    m.load_state_dict(sdv, strict=False)
    m = m.to(BF).cuda().eval()
    E, Q = 512, 2
    tcfg = {"params": {"ddconfig": {"encoder_name": "clip_vit_l", "select_layer": [-2, -3]}, "embed_dim": E,
                       "codebook_size": 512, "num_codebook": Q}, "max_vision_token_length": 578}
    tok = ImageTokenizer(tcfg, token_offset=32000, vision_model=m)
    sd = QO.random_vq_state_dict(c_feat=2048, embed_dim=E, dtype=BF)
    tok.model.load_state_dict(sd, strict=False)
    tok = tok.to(BF).cuda()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(32, 3, 336, 336, generator=g).to(BF).cuda()
    feat, h2d, idx, ids, xpre, _ = tok.model.encode_flat(x, offset=32000, boi=32512, eoi=32513, want_ids=True, want_xpre=True,
                                                        want_quant=False)
    N = 32 * 576
    f64 = feat.view(N, 2048).double()                      # float64 on the device (plumbing: the oracle's arithmetic)
    W = sd["quant_conv.weight"].view(E, 2048).double().cuda()
    h64 = f64 @ W.t() + sd["quant_conv.bias"].double().cuda()
    hb = h64.float().to(BF)
    # ---- stage A
    diff = h2d != hb
    nA = int(diff.sum())
    if nA:
        # hb is the bf16 nearest to h64; the kernel rounds h64 + eps (eps = fp32 accumulation noise of a 2048-deep reduction,
        # bounded here by 1.5e-5 of the row scale): its result may be the neighbouring bf16 only when h64 lies within eps of the
        # rounding boundary, i.e. |h2d - h64| <= |hb - h64| + 2 eps
        eps = 1.5e-5 * h64.abs().amax(1, keepdim=True).expand_as(h64)[diff]
        excess = (h2d.double() - h64)[diff].abs() - (hb.double() - h64)[diff].abs()
        assert bool((excess <= 2 * eps).all()), f"h rounded away from a boundary: worst excess / eps = {float((excess / eps).max()):.2f}"
    # ---- stage B (on the kernel's own h)
    Wi, bi = sd["quantize.project_in.weight"].double().cuda(), sd["quantize.project_in.bias"].double().cuda()
    xk = h2d.double() @ Wi.t() + bi
    pw = (2 ** torch.arange(8, -1, -1, device="cuda")).long()
    bits_k = (xk.float().to(BF).float() > 0).view(N, Q, 9)
    got_bits = ((idx.unsqueeze(-1) >> torch.arange(8, -1, -1, device="cuda")) & 1).bool()
    flipB = got_bits != bits_k
    nB = int(flipB.sum())
    mB = float(xk.abs().view(N, Q, 9)[flipB].max()) if nB else 0.0
    assert mB < 1e-5 and nB <= 4, (nB, mB)
    # ---- end to end vs the pure float64 chain
    x64 = hb.double() @ Wi.t() + bi
    bits64 = (x64.float().to(BF).float() > 0).view(N, Q, 9)
    flip = got_bits != bits64
    nflip, nidx = int(flip.sum()), int(((bits64.long() * pw).sum(-1) != idx).sum())
    mflip = float(x64.abs().view(N, Q, 9)[flip].max()) if nflip else 0.0
    min_margin = float(x64.abs().min())
    # ---- the other side: the reference's OWN arithmetic - the oracle executed op by op in bf16 (bf16 Linear outputs,
    # a bf16 pre-sign value) on the same tower features - against the same float64 chain, and against ours
    with torch.no_grad():                                   # (on the host: the oracle builds its bit weights on the CPU)
        _, _, idx_b, _, _ = QO.vq_encode({k: v.cpu() for k, v in sd.items()},
                                         feat.view(32, 24, 24, 2048).permute(0, 3, 1, 2).contiguous().cpu(), num_codebooks=Q, codebook_dim=9)
    idx_b = idx_b.reshape(N, Q).to(idx.device)
    bits_b = ((idx_b.unsqueeze(-1) >> torch.arange(8, -1, -1, device="cuda")) & 1).bool()
    n_theirs = int((bits_b != bits64).sum())
    n_ours_theirs = int((bits_b != got_bits).sum())
    parity_report(f"[configs[1] VQ encode B=32 E=512, two-sided] sign-bit flips vs the float64 chain: ours {nflip}, the reference's bf16 "
                  f"arithmetic (oracle op by op in bf16) {n_theirs}; ours vs theirs {n_ours_theirs} of {N * Q * 9} bits "
                  f"({int((idx_b != idx).sum())} of {N * Q} indices)")
    # (bits where ours != theirs are bits where one of the two left the float64 chain: a count identity, not a tolerance)
    assert n_ours_theirs <= nflip + n_theirs, (n_ours_theirs, nflip, n_theirs)
    parity_report(f"[configs[1] VQ encode B=32 E=512] {N * Q * 9} sign bits: stage A (h rounding) {nA} of {h2d.numel()} elements "
                  f"differ from bf16(h64), all within fp32 accumulation noise of a rounding boundary; stage B (bits on the kernel's own h) {nB} flips, max margin "
                  f"{mB:.2e}; end-to-end vs float64 chain: {nflip} bit flips in {nidx} of {N * Q} indices, largest |x64| among "
                  f"flips {mflip:.2e}, min |x64| over all bits {min_margin:.2e}")
    # Gate = the margin rule, nothing looser (observed on MI355X: 1 flip at |x64| = 1.7e-5; the reference's own bf16 arithmetic: 4).
    # A bit may differ from the float64 chain only where the kernel's h is the ADJACENT bf16 of h64 (stage A, checked above) and
    # that one-ulp difference, pushed through project_in, covers the pre-sign value:  |x64| <= sum_c |W_in[bit, c]| |h2d - hb|[c]
    # (+ the stage-B margin 1e-5); and no more flips than the reference's own bf16 run makes, + 2.
    reach = ((h2d.double() - hb.double()).abs() @ Wi.abs().t()).view(N, Q, 9)
    if nflip:
        over = (x64.abs().view(N, Q, 9) - reach)[flip]
        assert float(over.max()) <= 1e-5, f"a sign bit flipped outside the reach of the h rounding: {float(over.max()):.3e}"
    assert nflip <= n_theirs + 2 and mflip < 1e-3, (nflip, n_theirs, mflip)
    assert torch.equal(ids[:, :, 1:-1].permute(1, 2, 0).reshape(N, Q) - 32000, idx)
