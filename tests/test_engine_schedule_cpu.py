"""CPU check of the decoder engine's LAUNCH SCHEDULES (no GPU, no kernels): the round-6 multi-problem schedule
(`decoder_engine.MULTI`, with and without `CHAIN`) must hand the same operands to the same GEMMs as the one-launch-per-GEMM
schedule - every activation, the layer's input gradient and every weight gradient of a decoder layer come out identical.

How: `libra_amd.kernels` is replaced by host stand-ins for the duration of the test.  GEMMs are real (fp32 product -> bf16, row
maps, residual, reduction-major operands); the other kernels are arbitrary but deterministic functions of ALL their inputs with
the right shapes (the comparison is schedule A vs schedule B through the SAME stand-ins, so they need to be consistent, not
faithful - the arithmetic of the real kernels is the business of the `-m gpu` parity tests).  The stand-in for `gemm_multi`
is ADVERSARIAL about ordering: every problem of a launch reads its inputs as they were BEFORE the launch (snapshot), except a
problem that declares `reads=` - that one runs after its producer.  A problem that silently depends on another problem of its
launch (the bug class a fused schedule can introduce) therefore reads stale data and the test fails."""
import types

import pytest
import torch

from helpers import load_golden, sub_params

BF = torch.bfloat16


class _Spec:
    def __init__(self, a, b, kw):
        self.a, self.b, self.kw = a, b, dict(kw)
        self.reads = self.kw.pop("reads", None)
        self.kw.pop("splitk", None)
        self.out = self.kw.get("out")


def _gemm(a, b, *, out=None, resid=None, a_t=False, b_t=False, a_rows=None, c_rows=None, k=None, tile=0, bias=None, **unused):
    assert not unused, unused
    A = a.float().t() if a_t else a.float()
    if a_rows is not None:
        A = A[a_rows.long()]
    Bm = b.float() if b_t else b.float().t()                     # [K, N]
    if k is not None:
        A, Bm = A[:, :k], Bm[:k]
    y = A @ Bm
    if bias is not None:
        y = y + bias.float()
    if out is None:
        out = torch.empty(y.shape, dtype=BF)
    if c_rows is not None:
        rows = c_rows.long()
        if resid is not None:
            y = y + resid.float()[rows]
        out[rows] = y.to(BF)
    else:
        if resid is not None:
            y = y + resid.float()
        out.copy_(y.to(BF))
    return out


def _fake_kernels(real_K, log):
    """A stand-in module for libra_amd.kernels (only what layer_forward / layer_backward touch)."""
    F = types.SimpleNamespace()
    F.round_up, F.alloc_rows, F.RowArena, F.BF16 = real_K.round_up, real_K.alloc_rows, real_K.RowArena, BF
    mix = lambda *ts: sum((t.float() * (0.5 + 0.25 * i)) for i, t in enumerate(ts))

    def gemm_nt(a, b, **kw):
        log.append(("gemm", 1))
        return _gemm(a, b, **kw)

    def gemm_nt_grouped(a_list, b_list, outs, **kw):
        log.append(("grouped", len(a_list)))
        for a, b, o in zip(a_list, b_list, outs):
            _gemm(a, b, out=o, **kw)
        return outs

    def gemm_spec(a, b, **kw):
        sp = _Spec(a, b, kw)
        if sp.out is None:                                         # same allocation rule as the real wrapper
            M = kw["a_rows"].numel() if kw.get("a_rows") is not None else (a.shape[1] if kw.get("a_t") else a.shape[0])
            N = b.shape[1] if kw.get("b_t") else b.shape[0]
            sp.out = sp.kw["out"] = torch.zeros((M, N), dtype=BF)
        return sp

    def gemm_multi(specs):
        specs = list(specs)
        log.append(("multi", len(specs)))
        for sp in specs:
            assert sp.reads is None or any(o is sp.reads for o in specs), "reads= a problem outside the launch"
            assert sp.reads is None or sp.reads.reads is None, "one level of producer -> consumer only"
        snap = {}

        def frozen(t):                                             # the tensor as it was before the launch
            if t is None:
                return None
            key = (t.data_ptr(), tuple(t.shape), tuple(t.stride()))
            if key not in snap:
                snap[key] = t.clone()
            return snap[key]
        free = [sp for sp in specs if sp.reads is None]
        for sp in free:                                            # snapshot every input of every independent problem first
            sp._in = (frozen(sp.a), frozen(sp.b), frozen(sp.kw.get("resid")))
        for sp in reversed(free):                                  # ... then write, in an order no caller should rely on
            a, b, r = sp._in
            kw = dict(sp.kw)
            if r is not None:
                kw["resid"] = r
            _gemm(a, b, **kw)
        for sp in specs:
            if sp.reads is not None:
                _gemm(sp.a, sp.b, **sp.kw)
        return [sp.out for sp in specs]

    def rmsnorm_routed(x, w_l, w_v, flag, eps, *, out=None, save_rstd=False):
        xf = x.float()
        rstd = torch.rsqrt(xf.pow(2).mean(-1) + eps)
        w = torch.where(flag.bool()[:, None], w_v.float()[None], w_l.float()[None]) if w_v is not None else w_l.float()[None]
        y = (xf * rstd[:, None] * w).to(BF)
        return (y, rstd) if save_rstd else y

    def rmsnorm_routed_bwd(dy, x, w_l, w_v, flag, rstd, *, dres=None, out=None):
        w = torch.where(flag.bool()[:, None], w_v.float()[None], w_l.float()[None])
        dx = dy.float() * w * rstd[:, None] - 0.01 * x.float() * rstd[:, None]
        if dres is not None:
            dx = dx + dres.float()
        return dx.to(BF)

    def rmsnorm_routed_wgrad(dy, x, rstd, flag, dw_l, dw_v, *, rows_sel=None):
        g = dy.float() * x.float() * rstd[:, None]
        fl = flag.bool()
        if dw_l is not None:
            dw_l += g[~fl].sum(0)
        if dw_v is not None:
            dw_v += g[fl].sum(0)

    def rope_bridge(qkv, tb, bk_l, bk_v, bv_l, bv_v, flag, cos, sin, S, H):
        Hd = qkv.shape[1] // 3
        kc = (mix(qkv[:, Hd:2 * Hd]) + tb.float()[:, :8] @ torch.where(flag.bool()[:, None, None], bk_v.float()[None], bk_l.float()[None]).mean(0).t()[:8] * 0 +
              tb.float().sum(-1, keepdim=True) * 0.125).to(BF)
        vc = (mix(qkv[:, 2 * Hd:]) + tb.float().sum(-1, keepdim=True) * 0.0625 + bv_l.float().mean() + bv_v.float().mean() + bk_l.float().mean() + bk_v.float().mean()).to(BF)
        return kc, vc

    def bridge_attn_fwd(q, ks, kc, vs, vc, flag, lens, B, S, H, scale, *, need_lse=True, kv_start=None, out_lo=None):
        o = (mix(q, ks, kc, vs, vc) * scale).to(BF)
        if out_lo is not None:
            out_lo.copy_((o.float() * 0.001).to(BF))
        return o, torch.zeros(B * H * S)

    def bridge_attn_bwd(q, ks, kc, vs, vc, o, do, flag, lens, lse, B, S, H, scale, *, out_lo=None):
        base = do.float() * scale
        return tuple((base * c + 0.01 * t.float()).to(BF) for c, t in ((1.0, q), (0.5, ks), (0.25, kc), (2.0, vs), (1.5, vc)))

    def rope_bridge_bwd(dq, dks, dkc, dvs, dvc, cos, sin, S, H, dqkv, dkb, *, bridge_b=None, flag=None, dtb=None, positions=None):
        Hd = dq.shape[1]
        dqkv[:, :Hd] = dq
        dqkv[:, Hd:2 * Hd] = (dks.float() + dkc.float()).to(BF)
        dqkv[:, 2 * Hd:] = (dvs.float() + dvc.float()).to(BF)
        dkb.copy_(dkc)
        if dtb is not None:
            dtb[:, :8] = (dkc.float()[:, :8] * 0.5).to(BF)
            dtb[:, 8:16] = (dvc.float()[:, :8] * 0.25).to(BF)

    def swiglu(g, u, *, out=None):
        y = (torch.nn.functional.silu(g.float()).to(BF).float() * u.float()).to(BF)
        if out is None:
            return y
        out.copy_(y)
        return out

    def swiglu_bwd(dy, g, u, dg, du):
        s = torch.sigmoid(g.float())
        dg.copy_((dy.float() * u.float() * s * (1 + g.float() * (1 - s))).to(BF))
        du.copy_((dy.float() * g.float() * s).to(BF))

    def copy_rows(src, rows_sel, n, out, col0=0):
        out[:n, col0:col0 + src.shape[1]] = src[rows_sel.long()[:n]]
        return out

    def rank_outer_wgrad(x, coef, flag, *, transpose_out, want_l=True, want_v=True):
        fl = flag.bool()
        res = []
        for sel, want in ((~fl, want_l), (fl, want_v)):
            g = coef.float()[sel].t() @ x.float()[sel] if want else None                   # [ncoef, C]
            res.append(None if g is None else (g.t() if transpose_out else g).contiguous().to(BF))
        return tuple(res)
    F.f32_to_bf16 = lambda t: t.to(BF)
    F.gemm_multi_ok = lambda dev: True
    for k, v in dict(gemm_nt=gemm_nt, gemm_nt_grouped=gemm_nt_grouped, gemm_spec=gemm_spec, gemm_multi=gemm_multi,
                     rmsnorm_routed=rmsnorm_routed, rmsnorm_routed_bwd=rmsnorm_routed_bwd, rmsnorm_routed_wgrad=rmsnorm_routed_wgrad,
                     rope_bridge=rope_bridge, bridge_attn_fwd=bridge_attn_fwd, bridge_attn_bwd=bridge_attn_bwd,
                     rope_bridge_bwd=rope_bridge_bwd, swiglu=swiglu, swiglu_bwd=swiglu_bwd, copy_rows=copy_rows,
                     rank_outer_wgrad=rank_outer_wgrad).items():
        setattr(F, k, v)
    return F


def _run_layer(DE, sd, d, mode, monkeypatch, full_finetune):
    from libra_amd import kernels as real_K
    log = []
    monkeypatch.setattr(DE, "K", _fake_kernels(real_K, log))
    monkeypatch.setattr(DE, "MULTI", mode != "single")
    monkeypatch.setattr(DE, "CHAIN", mode == "chain")
    monkeypatch.setattr(DE, "_ARENA", real_K.RowArena(), raising=False)
    B, S = 2, 24
    N = B * S
    g = torch.Generator().manual_seed(3)
    flag = torch.zeros(N, dtype=torch.uint8)
    flag[3:9] = 1; flag[S + 10:S + 16] = 1                           # one image per sequence, not at the same place
    lang_idx = (flag == 0).nonzero().flatten().to(torch.int32)
    vis_idx = (flag == 1).nonzero().flatten().to(torch.int32)
    x = (torch.randn(N, d.hidden, generator=g) * 0.5).to(BF)
    dx_out = (torch.randn(N, d.hidden, generator=g) * 0.1).to(BF)
    lens = torch.full((B,), S, dtype=torch.int32)
    hd = d.hidden // d.heads
    cos, sin = torch.ones(S, hd), torch.zeros(S, hd)
    pk = DE.PackedOperands(sd, d)[0]
    sv = {}
    x_out = DE.layer_forward(sd, pk, 0, d, x, flag, lang_idx, vis_idx, lens, cos, sin, B, S, sv)
    grads = {}
    want = (lambda n: True) if full_finetune else (lambda n: "vision" in n)
    dx = DE.layer_backward(sd, pk, 0, d, sv, dx_out, flag, lang_idx, vis_idx, lens, cos, sin, B, S, grads, want)
    acts = {k: v.clone() for k, v in sv.items() if isinstance(v, torch.Tensor)}
    return x_out, dx, {k: v.clone() for k, v in grads.items() if v is not None}, acts, log, grads


@pytest.mark.parametrize("full_finetune", [False, True])
def test_multi_problem_schedules_equal_the_one_launch_per_gemm_schedule(monkeypatch, full_finetune):
    from libra_amd import decoder_engine as DE
    t, meta = load_golden("libra_tiny.safetensors")
    c = meta["cfg"]
    sd = {k: v.to(BF) for k, v in sub_params(t, "w.").items()}
    d = DE.DecDims(hidden=c["hidden_size"], inter=c["intermediate_size"], layers=c["num_hidden_layers"],
                   heads=c["num_attention_heads"], vocab=c["vocab_size"], vision_vocab=c["vision_vocab_size"],
                   codebooks=c["vision_codebook_num"], max_vision_len=c["max_vision_token_length"],
                   signal=c["contiguous_signal_size"], rank=c["bridge_rank"], down_ratio=c["vision_down_ratio"])
    ref = _run_layer(DE, sd, d, "single", monkeypatch, full_finetune)
    assert not any(k == "multi" for k, _ in ref[4])
    for mode in ("multi", "chain"):
        got = _run_layer(DE, sd, d, mode, monkeypatch, full_finetune)
        launches = [n for k, n in got[4] if k == "multi"]
        assert launches and not any(k == "grouped" for k, _ in got[4]), (mode, got[4])
        # forward 4 + backward 6 launches; chained: the six first low-rank stages ride inside them (no stand-alone GEMM left but ... none)
        assert len(launches) == 10, (mode, launches)
        assert sum(1 for k, _ in got[4] if k == "gemm") == (6 if mode == "multi" else 0), (mode, got[4])
        assert torch.equal(got[0], ref[0]), f"{mode}: layer output differs"
        assert torch.equal(got[1], ref[1]), f"{mode}: input gradient differs"
        assert set(got[2]) == set(ref[2]), (mode, set(got[2]) ^ set(ref[2]))
        for n in ref[2]:
            assert torch.equal(got[2][n], ref[2][n]), f"{mode}: weight gradient {n} differs"
        for n in ref[3]:
            assert torch.equal(got[3][n], ref[3][n]), f"{mode}: saved activation {n} differs"
    n_grads = len(ref[2])
    assert n_grads >= (33 if full_finetune else 24), n_grads


def test_a_device_that_fails_the_self_check_keeps_the_one_launch_per_gemm_schedule(monkeypatch):
    """kernels.gemm_multi_ok is the engines' gate: False (the on-device acceptance check of the multi-problem launch failed) must leave
    MULTI = True engines on the old schedule - no multi launch is ever issued."""
    from libra_amd import decoder_engine as DE, kernels as real_K
    t, meta = load_golden("libra_tiny.safetensors")
    c = meta["cfg"]
    sd = {k: v.to(BF) for k, v in sub_params(t, "w.").items()}
    d = DE.DecDims(hidden=c["hidden_size"], inter=c["intermediate_size"], layers=c["num_hidden_layers"],
                   heads=c["num_attention_heads"], vocab=c["vocab_size"], vision_vocab=c["vision_vocab_size"],
                   codebooks=c["vision_codebook_num"], max_vision_len=c["max_vision_token_length"],
                   signal=c["contiguous_signal_size"], rank=c["bridge_rank"], down_ratio=c["vision_down_ratio"])
    orig = _fake_kernels

    def failing(real, log):
        F = orig(real, log)
        F.gemm_multi_ok = lambda dev: False
        return F
    monkeypatch.setitem(globals(), "_fake_kernels", failing)
    got = _run_layer(DE, sd, d, "multi", monkeypatch, False)
    assert not any(k == "multi" for k, _ in got[4]) and any(k == "grouped" for k, _ in got[4])


def test_multi_schedule_under_a_capturing_gradient_store_writes_every_slot_directly(monkeypatch):
    """Data parallel: while a `dp.GradBuckets` captures, the weight-gradient problems of the multi schedule take the bucket slots
    as their outputs - including the packed [gate_A; up_A] and [q_A; k_A; v_A | bridge] gradients, which go out as one problem
    per parameter instead of being re-packed by a copy.  The bucket contents must equal the gradients of the plain run, and every
    vision weight matrix must have been written in place (the view handed to the store IS the slot)."""
    from libra_amd import decoder_engine as DE, dp
    t, meta = load_golden("libra_tiny.safetensors")
    c = meta["cfg"]
    sd = {k: torch.nn.Parameter(v.to(BF), requires_grad="vision" in k) for k, v in sub_params(t, "w.").items()}
    d = DE.DecDims(hidden=c["hidden_size"], inter=c["intermediate_size"], layers=c["num_hidden_layers"],
                   heads=c["num_attention_heads"], vocab=c["vocab_size"], vision_vocab=c["vision_vocab_size"],
                   codebooks=c["vision_codebook_num"], max_vision_len=c["max_vision_token_length"],
                   signal=c["contiguous_signal_size"], rank=c["bridge_rank"], down_ratio=c["vision_down_ratio"])
    sdd = {k: v.detach() for k, v in sd.items()}
    ref = _run_layer(DE, sdd, d, "multi", monkeypatch, False)
    named = [(n, p) for n, p in sd.items() if p.requires_grad and n.startswith("model.layers.0.")]
    st = dp.GradBuckets(named, bucket_bytes=1 << 14)
    captured = {}
    real_add = st.add

    def spy(grads):
        for n, g in grads.items():
            if g is not None and n in st.where:
                captured[n] = g.data_ptr() == st.view(n).data_ptr()
        return real_add(grads)
    monkeypatch.setattr(st, "add", spy)
    with st.capture():
        got = _run_layer(DE, sdd, d, "multi", monkeypatch, False)
        dp.emit(got[5])                                              # the engine's own tensors (bucket views where it wrote in place)
    for n, _ in named:
        assert torch.equal(st.view(n), ref[2][n]), n
    mats = [n for n, p in named if p.ndim == 2 and "bridge" not in n]
    assert mats and all(captured[n] for n in mats), [n for n in mats if not captured[n]]


def _fake_vit_kernels(real_K, log):
    """Stand-ins for what vit_engine.forward / backward touch (same rules as above: GEMMs real, the rest arbitrary but deterministic)."""
    F = _fake_kernels(real_K, log)
    base_gemm = F.gemm_nt

    def gemm_full(a, b, *, quick_gelu=False, qgelu_grad_of=None, preact_out=None, **kw):
        out = kw.pop("out", None)
        y = _gemm(a, b, **kw)
        if preact_out is not None:
            preact_out.copy_(y)
        if quick_gelu:
            y = (y.float() * torch.sigmoid(1.702 * y.float())).to(BF)
        if qgelu_grad_of is not None:
            s_ = torch.sigmoid(1.702 * qgelu_grad_of.float())
            y = (y.float() * s_ * (1 + 1.702 * qgelu_grad_of.float() * (1 - s_))).to(BF)
        if out is not None:
            out.copy_(y)
            return out
        return y

    def gemm_nt(a, b, **kw):
        log.append(("gemm", 1))
        return gemm_full(a, b, **kw)

    def gemm_multi(specs):
        specs = list(specs)
        log.append(("multi", len(specs)))
        ins = [(sp.a.clone(), sp.b.clone()) for sp in specs]          # every problem reads its operands as they were before the launch
        for sp, (a, b) in reversed(list(zip(specs, ins))):
            gemm_full(a, b, **sp.kw)
        return [sp.out for sp in specs]
    del base_gemm
    F.gemm_nt, F.gemm_multi = gemm_nt, gemm_multi

    def patch_im2col(pixel, P, Kpad):
        B, C, H, W = pixel.shape
        cols = pixel.unfold(2, P, P).unfold(3, P, P).permute(0, 2, 3, 1, 4, 5).reshape(B * (H // P) * (W // P), C * P * P)
        buf = real_K.alloc_rows(cols.shape[0], Kpad, pixel.device)
        buf[:cols.shape[0]].zero_()
        buf[:cols.shape[0], :cols.shape[1]] = cols
        return buf[:cols.shape[0]]

    def vit_embed_ln(patches, cls, pos, gamma, beta, B, T, eps, *, save=True):
        D = patches.shape[1]
        emb = torch.cat([cls.float().expand(B, 1, D), patches.float().view(B, T - 1, D)], 1) + pos.float()[None, :T]
        emb = emb.reshape(B * T, D)
        mean, var = emb.mean(-1), emb.var(-1, unbiased=False)
        rstd = torch.rsqrt(var + eps)
        hs0 = ((emb - mean[:, None]) * rstd[:, None] * gamma.float() + beta.float()).to(BF)
        return emb.to(BF), hs0, mean, rstd

    def layernorm_fwd(x, gamma, beta, eps, *, save_stats=True, out=None):
        xf = x.float()
        mean, rstd = xf.mean(-1), torch.rsqrt(xf.var(-1, unbiased=False) + eps)
        y = ((xf - mean[:, None]) * rstd[:, None] * gamma.float() + beta.float()).to(BF)
        if out is not None:
            out.copy_(y); y = out
        return y, (mean if save_stats else None), (rstd if save_stats else None)

    def layernorm_bwd(dy, x, gamma, mean, rstd, *, dres=None, dgamma=None, dbeta=None, dxsum=None, out=None):
        xh = (x.float() - mean[:, None]) * rstd[:, None]
        dx = dy.float() * gamma.float() * rstd[:, None] - 0.01 * xh
        if dres is not None:
            dx = dx + dres.float()
        if dgamma is not None:
            dgamma += (dy.float() * xh).sum(0)
        if dbeta is not None:
            dbeta += dy.float().sum(0)
        dx = dx.to(BF)
        if dxsum is not None:
            dxsum += dx.float().sum(0)
        if out is not None:
            out.copy_(dx); dx = out
        return dx

    def vit_attn_fwd(qkv, B, T, H, scale, *, need_lse=True, out=None, out_lo=None):
        D = qkv.shape[1] // 3
        o = ((qkv[:, :D].float() + 0.5 * qkv[:, D:2 * D].float() + 0.25 * qkv[:, 2 * D:].float()) * scale).to(BF)
        if out is not None:
            out.copy_(o); o = out
        if out_lo is not None:
            out_lo.zero_()
        return o, torch.zeros(B * H * T)

    def vit_attn_bwd(qkv, o, do, lse, B, T, H, scale, *, out_dqkv=None, out_lo=None):
        D = o.shape[1]
        d = torch.cat([do.float() * scale, do.float() * 0.5 * scale + 0.01 * qkv[:, D:2 * D].float(), do.float() * 0.25 * scale], 1).to(BF)
        if out_dqkv is not None:
            out_dqkv.copy_(d); d = out_dqkv
        return d

    def colsum(x, out):
        out += x.float().sum(0)
        return out

    def add_(a, b):
        a.copy_((a.float() + b.float()).to(BF))
        return a

    def patch_col2im(dcols, B, C, H, W, P):
        return dcols[:, :C * P * P].reshape(B, H // P, W // P, C, P, P).permute(0, 3, 1, 4, 2, 5).reshape(B, C, H, W).contiguous()
    for k, v in dict(patch_im2col=patch_im2col, vit_embed_ln=vit_embed_ln, layernorm_fwd=layernorm_fwd, layernorm_bwd=layernorm_bwd,
                     vit_attn_fwd=vit_attn_fwd, vit_attn_bwd=vit_attn_bwd, colsum=colsum, add_=add_, patch_col2im=patch_col2im).items():
        setattr(F, k, v)
    return F


def test_vit_backward_multi_schedule_equals_the_one_launch_per_gemm_schedule(monkeypatch):
    """vit_engine.MULTI pairs each dgrad with the (K-sliced) weight gradient that reads the same dY: every gradient and the pixel
    gradient must equal the separate-launch schedule's through the same stand-in kernels."""
    from libra_amd import kernels as real_K, vit_engine as VE
    from oracle import vit_oracle as VO
    dims = VE.VitDims(hidden=128, inter=256, layers=3, heads=2, patch=14, image=56)
    sd = {k: v.to(BF) for k, v in VO.random_vit_state_dict(hidden=128, inter=256, layers=3, patch=14, image=56).items()}
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 56, 56, generator=g).to(BF)
    T = dims.tokens
    res = {}
    for mode in (False, True):
        log = []
        monkeypatch.setattr(VE, "K", _fake_vit_kernels(real_K, log))
        monkeypatch.setattr(VE, "MULTI", mode)
        packed = VE.pack_forward(sd, dims)
        hs, saved = VE.forward(sd, packed, x, dims, save=True)
        dhs = [None] * len(hs)
        dhs[-2] = (torch.randn(2 * T, 128, generator=torch.Generator().manual_seed(5)) * 0.1).to(BF)
        dhs[-3] = (torch.randn(2 * T, 128, generator=torch.Generator().manual_seed(6)) * 0.1).to(BF)
        dpix, grads = VE.backward(sd, packed, saved, dhs, dims)
        res[mode] = (dpix, grads, log)
    assert sum(1 for k, _ in res[True][2] if k == "multi") == 4 * 2 and not any(k == "multi" for k, _ in res[False][2])
    assert torch.equal(res[True][0], res[False][0])
    assert set(res[True][1]) == set(res[False][1]) and len(res[True][1]) >= 16 * 2
    for n in res[False][1]:
        assert torch.equal(res[True][1][n], res[False][1][n]), n
