"""Parity of the routed-decoder kernels against fp32 closed-form math on the same bf16 inputs
(tolerance: see test_kernels_gpu.py)."""
import pytest
import torch
import torch.nn.functional as F

from test_kernels_gpu import close, rnd

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def K():
    from libra_amd import kernels
    return kernels


def _flags(N, seed, mode):
    g = torch.Generator().manual_seed(seed)
    if mode == "random":
        return (torch.rand(N, generator=g) < 0.4).to(torch.uint8)
    f = torch.zeros(N, dtype=torch.uint8)
    if mode == "span":
        a = N // 5
        f[a:a + max(1, N // 3)] = 1
    elif mode == "allvis":
        f[:] = 1
    return f


def test_routed_gemm_gather_scatter(K):
    Ntok, Kd, N = 300, 128, 200
    x, w = rnd(Ntok, Kd, seed=1), rnd(N, Kd, seed=2)
    idx = torch.randperm(Ntok, generator=torch.Generator().manual_seed(3))[:137].sort().values.to(torch.int32).cuda()
    out = torch.zeros(Ntok, N, dtype=BF, device="cuda")
    res = rnd(Ntok, N, seed=4)
    K.gemm_nt(x, w, out=out, a_rows=idx, c_rows=idx, resid=res)
    ref = torch.zeros(Ntok, N)
    ii = idx.cpu().long()
    ref[ii] = (x.float().cpu()[ii] @ w.float().cpu().t()) + res.float().cpu()[ii]
    close(out, ref, what="routed gemm")
    untouched = torch.ones(Ntok, dtype=torch.bool); untouched[ii] = False
    assert float(out.cpu()[untouched].abs().max()) == 0.0


@pytest.mark.parametrize("rows,D", [(37, 256), (1000, 4096), (5, 6144)])
def test_rmsnorm_routed(K, rows, D):
    from oracle import libra_oracle as LO
    x = rnd(rows, D, seed=1)
    wl, wv = (rnd(D, seed=2) * 0.1 + 1).to(BF), (rnd(D, seed=3) * 0.1 + 1).to(BF)
    flag = _flags(rows, 5, "random").cuda()
    y = K.rmsnorm_routed(x, wl, wv, flag, 1e-6)
    # the reference rounds x*rstd to bf16 BEFORE the weight multiply (modeling_llama.py:132); the oracle run in bf16
    # reproduces exactly those rounding points, so the comparison is (nearly) bit-exact
    xb = x.cpu()
    ref = LO.routed(xb, flag.cpu().bool(), lambda t: LO.rms_norm(t, wl.cpu(), 1e-6), lambda t: LO.rms_norm(t, wv.cpu(), 1e-6))
    d = (y.float().cpu() - ref.float()).abs()
    assert float((d > 0).float().mean()) < 2e-3 and float(d.max()) <= float(ref.float().abs().max()) * 2 ** -7
    y2 = K.rmsnorm_routed(x, wl, None, None, 1e-6)
    d2 = (y2.float().cpu() - LO.rms_norm(xb, wl.cpu(), 1e-6).float()).abs()
    assert float((d2 > 0).float().mean()) < 2e-3
    # and against the fp32 (single-rounding) value it stays within two bf16 ulps
    close(y, LO.routed(xb.float(), flag.cpu().bool(), lambda t: LO.rms_norm(t, wl.float().cpu(), 1e-6),
                       lambda t: LO.rms_norm(t, wv.float().cpu(), 1e-6)), rel=8e-3, what="routed rmsnorm vs fp32")


def test_rmsnorm_routed_wgrad_full_and_row_subset(K):
    """Per-modality weight gradient dw_m = sum_{rows of m} dy * x * rstd: all rows, and with only one modality's rows visited
    (`rows_sel`, the frozen-language case) - same sums, nothing added to the other output."""
    rows, D = 1000, 4096
    dy, x = rnd(rows, D, seed=11), rnd(rows, D, seed=12)
    rstd = (torch.rand(rows, generator=torch.Generator().manual_seed(13)) + 0.5).cuda()
    flag = _flags(rows, 5, "random").cuda()
    t = dy.float() * x.float() * rstd[:, None]
    vis = flag.bool()
    ref_l, ref_v = t[~vis].sum(0), t[vis].sum(0)
    dl, dv = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    K.rmsnorm_routed_wgrad(dy, x, rstd, flag, dl, dv)
    close(dl, ref_l, rel=1e-4, what="dw_lang"); close(dv, ref_v, rel=1e-4, what="dw_vis")
    vis_idx = torch.nonzero(vis).squeeze(1).to(torch.int32)
    dl2, dv2 = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")             # (the kernel ADDS onto its outputs)
    K.rmsnorm_routed_wgrad(dy, x, rstd, flag, dl2, dv2, rows_sel=vis_idx)
    close(dv2, ref_v, rel=1e-4, what="dw_vis (vision rows only)")
    assert float(dl2.abs().max()) == 0.0
    lang_idx = torch.nonzero(~vis).squeeze(1).to(torch.int32)
    dl3, dv3 = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    K.rmsnorm_routed_wgrad(dy, x, rstd, flag, dl3, dv3, rows_sel=lang_idx)
    close(dl3, ref_l, rel=1e-4, what="dw_lang (text rows only)")
    assert float(dv3.abs().max()) == 0.0
    dl4, dv4 = torch.full((D,), 3.0, device="cuda"), torch.full((D,), 3.0, device="cuda")   # an empty selection adds nothing
    K.rmsnorm_routed_wgrad(dy, x, rstd, flag, dl4, dv4, rows_sel=vis_idx[:0].contiguous())
    assert float((dl4 - 3.0).abs().max()) == 0.0 and float((dv4 - 3.0).abs().max()) == 0.0


def _attn_ref(q, ks, kc, vs, vc, flag, lens, B, S, H, scale):
    d = 128
    def hd(t):
        return t.float().view(B, S, H, d).transpose(1, 2)
    q, ks, kc, vs, vc = map(hd, (q, ks, kc, vs, vc))
    m = flag.view(B, S).bool()
    cross = (m[:, :, None] != m[:, None, :]).unsqueeze(1)
    s = torch.where(cross, q @ kc.transpose(-1, -2), q @ ks.transpose(-1, -2)) * scale
    i = torch.arange(S)
    allowed = (i[None, :] <= i[:, None])[None, None] & (i[None, None, None, :] < lens.view(B, 1, 1, 1))
    s = s.masked_fill(~allowed, float("-inf"))
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p, nan=0.0)
    o = (p * (~cross)) @ vs + (p * cross) @ vc
    return o.transpose(1, 2).reshape(B * S, H * d), torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,S,H,mode", [(1, 16, 1, "span"), (2, 33, 2, "random"), (1, 128, 1, "none"), (2, 200, 2, "span"),
                                        (1, 640, 2, "random"), (1, 257, 1, "allvis"), (2, 1024, 2, "span"),
                                        # BASELINE sequence lengths (configs[2]: 2048, configs[4]: 4096), one or two heads
                                        (2, 2048, 1, "span"), (1, 4096, 1, "random"),
                                        # 640 items on <= 256 persistent workgroups: every workgroup walks 2-3 items (the rotation
                                        # schedule, LDS reuse across items) - the training shape's regime, at a size the reference finishes
                                        (2, 512, 160, "span")])
def test_bridge_attention_fwd(K, B, S, H, mode):
    N, D = B * S, H * 128
    q, ks, kc, vs, vc = [rnd(N, D, seed=10 + i) for i in range(5)]
    flag = _flags(N, 7, mode)
    lens = torch.full((B,), S, dtype=torch.int32)
    if B > 1:
        lens[1] = max(1, S - S // 4)
    o, lse = K.bridge_attn_fwd(q, ks, kc, vs, vc, flag.cuda(), lens.cuda(), B, S, H, 128 ** -0.5, need_lse=True)
    ro, rl = _attn_ref(q.cpu(), ks.cpu(), kc.cpu(), vs.cpu(), vc.cpu(), flag, lens.long(), B, S, H, 128 ** -0.5)
    # rows beyond a sequence's valid length are don't-care in the reference too (labels -100): compare valid rows
    valid = (torch.arange(S)[None, :] < lens[:, None].long()).reshape(N)
    close(o.cpu()[valid], ro[valid], rel=2e-3, what="bridge attn out")
    vl = valid.view(B, 1, S).expand(B, H, S)
    close(lse.cpu()[vl], rl[vl], rel=1e-4, what="bridge lse")
    assert torch.isfinite(o.float()).all()


@pytest.mark.parametrize("B,S,H,mode", [(1, 16, 1, "span"), (2, 33, 2, "random"), (1, 128, 1, "none"), (2, 200, 2, "span"),
                                        (1, 320, 1, "random"), (2, 512, 2, "span"),
                                        (2, 2048, 1, "span"), (1, 4096, 1, "random"),      # BASELINE sequence lengths
                                        (2, 512, 160, "span")])                            # several items per persistent workgroup (dQ pass)
def test_bridge_attention_bwd(K, B, S, H, mode):
    N, D = B * S, H * 128
    q, ks, kc, vs, vc = [rnd(N, D, seed=20 + i) for i in range(5)]
    do = rnd(N, D, seed=30)
    flag = _flags(N, 8, mode)
    lens = torch.full((B,), S, dtype=torch.int32)
    if B > 1:
        lens[1] = max(1, S - S // 4)
    sc = 128 ** -0.5
    o, lse = K.bridge_attn_fwd(q, ks, kc, vs, vc, flag.cuda(), lens.cuda(), B, S, H, sc, need_lse=True)
    grads = K.bridge_attn_bwd(q, ks, kc, vs, vc, o, do, flag.cuda(), lens.cuda(), lse, B, S, H, sc)
    ins = [t.float().cpu().requires_grad_(True) for t in (q, ks, kc, vs, vc)]
    ro, _ = _attn_ref(*ins, flag, lens.long(), B, S, H, sc)
    valid = (torch.arange(S)[None, :] < lens[:, None].long()).reshape(N)
    # gradients only flow from valid query rows (the reference's padded rows carry label -100)
    (ro * (do.float().cpu() * valid[:, None])).sum().backward()
    # NOTE: our kernel also propagates dO of padded QUERY rows; zero them for an apples-to-apples comparison
    do2 = (do.float() * valid.cuda()[:, None]).to(BF)
    grads = K.bridge_attn_bwd(q, ks, kc, vs, vc, o, do2, flag.cuda(), lens.cuda(), lse, B, S, H, sc)
    for g, r, n in zip(grads, ins, ("dq", "dk_same", "dk_cross", "dv_same", "dv_cross")):
        ref = r.grad
        if n == "dq":
            close(g.cpu()[valid], ref[valid], rel=6e-3, what=n)
        else:
            close(g, ref, rel=6e-3, what=n)
        assert torch.isfinite(g.float()).all(), n


@pytest.mark.parametrize("pad,S,mode", [(70, 300, "span"), (578, 700, "random"), (33, 200, "none"), (64, 256, "span"),
                                        (100, 2048, "span")])
def test_bridge_attention_left_padding_equals_unpadded(K, pad, S, mode):
    """a 32-query wave that straddles kv_start with kv_start >= 64 saw a fully masked first tile for its real
    rows while its pad rows kept their keys; the wave-uniform rescale then computed exp2(-inf - -inf) = NaN for the real rows.
    A batch of one image prompt and one text prompt is left-padded by 578.  The padded run must be finite and equal the same
    tokens run without padding (positions do not enter the kernel)."""
    H = 2
    Sr = S - pad
    D = H * 128
    real = [rnd(Sr, D, seed=80 + i) for i in range(5)]
    fl_r = _flags(Sr, 9, mode)
    padded = []
    for i, t in enumerate(real):
        junk = rnd(pad, D, seed=90 + i) * 3.0
        padded.append(torch.cat([junk, t], 0).contiguous())
    fl_p = torch.cat([_flags(pad, 10, "random"), fl_r])
    sc = 128 ** -0.5
    start = torch.tensor([pad], dtype=torch.int32).cuda()
    lens_p = torch.tensor([S], dtype=torch.int32).cuda()
    o_p, lse_p = K.bridge_attn_fwd(*padded, fl_p.cuda(), lens_p, 1, S, H, sc, need_lse=True, kv_start=start)
    o_r, lse_r = K.bridge_attn_fwd(*real, fl_r.cuda(), torch.tensor([Sr], dtype=torch.int32).cuda(), 1, Sr, H, sc, need_lse=True)
    assert torch.isfinite(o_p[pad:].float()).all() and torch.isfinite(lse_p[:, :, pad:]).all()
    ro, rl = _attn_ref(*[t.cpu() for t in real], fl_r, torch.tensor([Sr]), 1, Sr, H, sc)
    close(o_p.cpu()[pad:], ro, rel=2e-3, what="left-padded bridge attn out")
    close(lse_p.cpu()[:, :, pad:], rl, rel=1e-4, what="left-padded lse")
    # same arithmetic up to the tile phase of the keys: bf16-rounding level agreement with the unpadded launch
    assert float((o_p[pad:].float() - o_r.float()).abs().max()) <= float(o_r.float().abs().max()) * 2 ** -6


def test_bridge_attention_deferred_max_rescale_branch(K):
    """The deferred running-max rescale is a rare, data-dependent branch (cdna_hip_programming.md rule 26): force it by spiking
    one key against one query at a late tile (raw score far above the row's other scores), full-tensor fp32 reference."""
    B, S, H = 1, 1024, 1
    N, D = B * S, H * 128
    q, ks, kc, vs, vc = [rnd(N, D, seed=120 + i) for i in range(5)]
    flag = _flags(N, 3, "span")
    for (qi, kj, amp) in ((700, 650, 6.0), (1000, 130, 9.0), (301, 300, 12.0)):
        ks[kj] = (q[qi].float() * amp / 8).to(BF)              # q.k / sqrt(d) ~ amp * |q|^2 / (8 sqrt(128))
        kc[kj] = ks[kj]
    lens = torch.full((B,), S, dtype=torch.int32)
    o, lse = K.bridge_attn_fwd(q, ks, kc, vs, vc, flag.cuda(), lens.cuda(), B, S, H, 128 ** -0.5, need_lse=True)
    ro, rl = _attn_ref(q.cpu(), ks.cpu(), kc.cpu(), vs.cpu(), vc.cpu(), flag, lens.long(), B, S, H, 128 ** -0.5)
    close(o.cpu(), ro, rel=4e-3, what="bridge attn out (spiked keys)")
    close(lse.cpu(), rl, rel=1e-4, what="bridge lse (spiked keys)")


def test_bridge_attention_output_residual(K):
    """Same contract for the bridge kernels: identical bf16 output, O + out_lo ~ fp32 O, dq no worse (usually several times better)."""
    from helpers import rel_err
    B, S, H = 2, 512, 2
    N, D = B * S, H * 128
    q, ks, kc, vs, vc = [rnd(N, D, seed=60 + i) for i in range(5)]
    do = rnd(N, D, seed=70)
    flag = _flags(N, 8, "span")
    lens = torch.full((B,), S, dtype=torch.int32)
    sc = 128 ** -0.5
    o_lo = torch.empty(N, D, dtype=BF, device="cuda")
    o, lse = K.bridge_attn_fwd(q, ks, kc, vs, vc, flag.cuda(), lens.cuda(), B, S, H, sc, need_lse=True, out_lo=o_lo)
    o2, _ = K.bridge_attn_fwd(q, ks, kc, vs, vc, flag.cuda(), lens.cuda(), B, S, H, sc, need_lse=True)
    assert torch.equal(o, o2)
    ins = [t.float().cpu().requires_grad_(True) for t in (q, ks, kc, vs, vc)]
    ro, _ = _attn_ref(*ins, flag, lens.long(), B, S, H, sc)
    e_hi = float((o.float().cpu() - ro.detach()).abs().max())
    e_both = float(((o.float() + o_lo.float()).cpu() - ro.detach()).abs().max())
    (ro * do.float().cpu()).sum().backward()
    gp = K.bridge_attn_bwd(q, ks, kc, vs, vc, o, do, flag.cuda(), lens.cuda(), lse, B, S, H, sc)
    gl = K.bridge_attn_bwd(q, ks, kc, vs, vc, o, do, flag.cuda(), lens.cuda(), lse, B, S, H, sc, out_lo=o_lo)
    ep, el = rel_err(gp[0].float().cpu(), ins[0].grad), rel_err(gl[0].float().cpu(), ins[0].grad)
    print(f"bridge O error {e_hi:.2e} -> {e_both:.2e}; dq rel err {ep:.2e} -> {el:.2e}")
    assert e_both < 0.8 * e_hi and el <= ep * 1.05 + 1e-4
    for g, r, n in zip(gl, ins, ("dq", "dk_same", "dk_cross", "dv_same", "dv_cross")):
        close(g, r.grad, rel=6e-3, what=n)


def test_bridge_attention_bwd_deterministic(K):
    """No atomics anywhere: two runs are bit-identical (also a race screen for the LDS-DMA / barrier ordering)."""
    B, S, H = 2, 1100, 2
    N, D = B * S, H * 128
    q, ks, kc, vs, vc = [rnd(N, D, seed=40 + i) for i in range(5)]
    do = rnd(N, D, seed=50)
    flag = _flags(N, 11, "span").cuda()
    lens = torch.tensor([S, S - 300], dtype=torch.int32).cuda()
    sc = 128 ** -0.5
    runs = []
    for _ in range(3):
        o, lse = K.bridge_attn_fwd(q, ks, kc, vs, vc, flag, lens, B, S, H, sc, need_lse=True)
        runs.append((o, lse) + tuple(K.bridge_attn_bwd(q, ks, kc, vs, vc, o, do, flag, lens, lse, B, S, H, sc)))
    for r in runs[1:]:
        for x, y in zip(runs[0], r):
            assert torch.equal(x, y)


def test_rope_bridge(K):
    from oracle import libra_oracle as LO
    B, S, H = 2, 24, 2
    N, D = B * S, H * 128
    qkv = rnd(N, 3 * D, seed=1)
    tb = torch.zeros(N, 64, dtype=BF, device="cuda"); tb[:, :16] = rnd(N, 16, seed=2)
    bkl, bkv, bvl, bvv = [rnd(D, 8, seed=3 + i, scale=0.3) for i in range(4)]
    flag = _flags(N, 9, "span")
    cosf, sinf = LO.rope_tables(128, 64)
    cos, sin = cosf.to(BF).cuda(), sinf.to(BF).cuda()
    q0 = qkv.clone()
    kc, vc = K.rope_bridge(qkv, tb, bkl, bkv, bvl, bvv, flag.cuda(), cos, sin, S, H)
    # reference = the same op sequence in bf16 on the CPU (every op rounds, as in the reference's bf16 run)
    x = q0.cpu()
    q, k, v = x[:, :D], x[:, D:2 * D], x[:, 2 * D:]
    f = flag.bool()
    tk, tv = tb[:, :8].cpu(), tb[:, 8:16].cpu()
    kb = torch.where(f[:, None], tk @ bkv.cpu().t(), tk @ bkl.cpu().t())
    vb = torch.where(f[:, None], tv @ bvv.cpu().t(), tv @ bvl.cpu().t())
    pos = torch.arange(S).repeat(B)
    c, s_ = cos.cpu()[pos], sin.cpu()[pos]

    def rope(t):
        t = t.reshape(N, H, 128)
        return ((t * c[:, None]) + (LO.rotate_half(t) * s_[:, None])).reshape(N, D)

    def same(a, b, what):
        d = (a.float().cpu() - b.float()).abs()
        # identical rounding points; the only freedom left is fma-vs-mul+add inside one product
        assert float((d > 0).float().mean()) < 5e-3, (what, float((d > 0).float().mean()))
        assert float(d.max()) <= float(b.float().abs().max()) * 2 ** -6, what
    same(qkv[:, :D], rope(q), "q rope")
    same(qkv[:, D:2 * D], rope(k), "k_same")
    same(kc, rope(k + kb), "k_cross")
    same(vc, v + vb, "v_cross")
    assert torch.equal(qkv[:, 2 * D:], q0[:, 2 * D:])


def test_rope_bridge_bwd(K):
    """dq/dk/dv through the rope transpose, dkb, and the fused rank-8 gradients dtb = [B_k[m]^T dkb, B_v[m]^T dvb]."""
    from oracle import libra_oracle as LO
    B, S, H = 2, 37, 32          # H = 32: the kernel's full 512-thread token group (Libra-7B/11B), ragged token tail
    N, D = B * S, H * 128
    dq, dks, dkc, dvs, dvc = [rnd(N, D, seed=10 + i) for i in range(5)]
    bkl, bkv, bvl, bvv = [rnd(D, 8, seed=3 + i, scale=0.3) for i in range(4)]
    flag = _flags(N, 9, "span")
    cosf, sinf = LO.rope_tables(128, 64)
    cos, sin = cosf.to(BF).cuda(), sinf.to(BF).cuda()
    dqkvt = torch.full((N, 3 * D + 64), 7.0, dtype=BF, device="cuda")
    dqkv, dtb = dqkvt[:, :3 * D], dqkvt[:, 3 * D:]
    dkb = torch.empty(N, D, dtype=BF, device="cuda")
    K.rope_bridge_bwd(dq, dks, dkc, dvs, dvc, cos, sin, S, H, dqkv, dkb, bridge_b=tuple(t.t().contiguous() for t in (bkl, bkv, bvl, bvv)),
                      flag=flag.cuda(), dtb=dtb)
    pos = torch.arange(S).repeat(B)
    c, s_ = cos.float().cpu()[pos][:, None], sin.float().cpu()[pos][:, None]

    def rope_t(t):               # transpose of x -> x cos + rotate_half(x) sin
        t = t.float().cpu().reshape(N, H, 128)
        t1, t2 = t[..., :64], t[..., 64:]
        return torch.cat([t1 * c[..., :64] + t2 * s_[..., 64:], t2 * c[..., 64:] - t1 * s_[..., :64]], -1).reshape(N, D)
    close(dqkv[:, :D], rope_t(dq), rel=4e-3, what="dq")
    close(dqkv[:, D:2 * D], rope_t(dks.float() + dkc.float()), rel=4e-3, what="dk")
    close(dqkv[:, 2 * D:], dvs.float() + dvc.float(), rel=4e-3, what="dv")
    close(dkb, rope_t(dkc), rel=4e-3, what="dkb")
    f = flag.bool()[:, None]
    kbf, dvf = dkb.float().cpu(), dvc.float().cpu()
    tk = torch.where(f, kbf @ bkv.float().cpu(), kbf @ bkl.float().cpu())
    tv = torch.where(f, dvf @ bvv.float().cpu(), dvf @ bvl.float().cpu())
    close(dtb[:, 0:8], tk, rel=6e-3, what="dt_k")
    close(dtb[:, 8:16], tv, rel=6e-3, what="dt_v")
    assert torch.equal(dtb[:, 16:], torch.full((N, 48), 7.0, dtype=BF, device="cuda"))     # untouched columns
    # without the bridge operands the kernel leaves dtb alone
    dtb.fill_(3.0)
    K.rope_bridge_bwd(dq, dks, dkc, dvs, dvc, cos, sin, S, H, dqkv, dkb)
    assert torch.equal(dtb, torch.full((N, 64), 3.0, dtype=BF, device="cuda"))


@pytest.mark.parametrize("M,I,K_", [(8, 11008, 4096), (1, 1002, 576), (16, 2752, 1024), (5, 4096, 11008)])
def test_gemm_swiglu_skinny_is_the_gemm_then_swiglu_bit_for_bit(K, M, I, K_):
    """The generation step's fused gate | up GEMM + SwiGLU (libra_gemm_swiglu_skinny) against the two launches it replaces, with and
    without a row gather on A, plus a float reference; ragged I (not a multiple of the columns per workgroup)."""
    a = rnd(M + 7, K_, seed=51, scale=0.5)
    w = rnd(2 * I, K_, seed=52, scale=0.05)
    rows = torch.randperm(M + 7, generator=torch.Generator().manual_seed(7))[:M].to(torch.int32).cuda()
    ident = torch.arange(M, dtype=torch.int32).cuda()
    for ar in (ident, rows):          # (row-mapped calls, as the decode step makes them: a plain M = 8 gemm_nt may take the split-K MFMA route)
        out = K.gemm_swiglu_skinny(a, w, a_rows=ar)
        assert out.shape == (M, I)
        if I % 8 == 0:                # (libra_swiglu wants 16-byte rows; the ragged case is checked against the float reference below)
            gu = K.gemm_nt(a, w, a_rows=ar)
            ref = K.swiglu(gu[:, :I], gu[:, I:])
            assert torch.equal(out, ref), float((out.float() - ref.float()).abs().max())
    assert torch.equal(K.gemm_swiglu_skinny(a[:M], w), K.gemm_swiglu_skinny(a, w, a_rows=ident))
    af = a[:M].float()
    close(K.gemm_swiglu_skinny(a[:M], w), F.silu(af @ w[:I].float().t()) * (af @ w[I:].float().t()), rel=2e-2, what="swiglu skinny")


def test_swiglu_gather_ce(K):
    rows, I = 77, 512
    gu = rnd(rows, 2 * I, seed=1)
    y = K.swiglu(gu[:, :I], gu[:, I:])
    g, u = gu[:, :I].float(), gu[:, I:].float()
    close(y, F.silu(g) * u, rel=4e-3, what="swiglu")
    table = rnd(50, 64, seed=2)
    idx = torch.randint(100, 150, (30,), generator=torch.Generator().manual_seed(3)).cuda()
    sel = torch.arange(0, 30, 2, dtype=torch.int32).cuda()
    out = torch.zeros(15, 128, dtype=BF, device="cuda")
    K.gather_rows(table, idx, 100, sel, 15, out, 64)
    assert torch.equal(out[:, 64:], table[(idx[sel.long()] - 100)])
    K.copy_rows(table, sel, 15, out, 0)
    assert torch.equal(out[:, :64], table[sel.long()])
    V = 515
    z = torch.zeros(40, 520, dtype=BF, device="cuda"); z[:, :V] = rnd(40, V, seed=5) * 3
    tgt = torch.randint(0, V, (40,), generator=torch.Generator().manual_seed(6)) + 1000
    tgt[::7] = -100
    loss = K.ce_rows(z[:, :V], tgt.cuda(), 1000)
    ref = F.cross_entropy(z[:, :V].float().cpu(), (tgt - 1000).clamp_min(0), reduction="none")
    ref[tgt < 0] = 0
    close(loss, ref, rel=1e-4, what="ce rows")
    # backward: the upstream scalar as a device operand == the same scalar folded into the host coefficients (power of two: exact)
    zt, tt = z[:, :V], tgt.cuda()
    g_host = K.ce_rows_bwd(zt, tt, None, 1000, 0.25 * 4.0, 0.0, torch.empty(40, V, dtype=BF, device="cuda"))
    g_dev = K.ce_rows_bwd(zt, tt, None, 1000, 0.25, 0.0, torch.empty(40, V, dtype=BF, device="cuda"),
                          scale=torch.tensor([4.0], device="cuda"))
    assert torch.equal(g_host, g_dev)
    zl = z[:, :V].float().cpu().requires_grad_(True)
    F.cross_entropy(zl, (tgt - 1000).clamp_min(0), reduction="none")[tgt >= 0].sum().backward()
    close(g_dev, zl.grad, rel=4e-3, what="ce rows bwd")


def test_rope_bridge_with_explicit_positions(K):
    """rope_bridge_pos on a shuffled subset of tokens with their positions == the rows rope_bridge produces in sequence order
    (bit-exact: same kernel, the position operand replaces n % S)."""
    from oracle import libra_oracle as LO
    B, S, H = 2, 40, 2
    N, D = B * S, H * 128
    qkv = rnd(N, 3 * D, seed=1)
    tb = torch.zeros(N, 64, dtype=BF, device="cuda"); tb[:, :16] = rnd(N, 16, seed=2)
    w = [rnd(D, 8, seed=3 + i, scale=0.3) for i in range(4)]
    flag = _flags(N, 9, "random").cuda()
    cosf, sinf = LO.rope_tables(128, 64)
    cos, sin = cosf.to(BF).cuda(), sinf.to(BF).cuda()
    full = qkv.clone()
    kc, vc = K.rope_bridge(full, tb, *w, flag, cos, sin, S, H)
    pick = torch.randperm(N, generator=torch.Generator().manual_seed(4))[:23].cuda()
    sub = qkv[pick].contiguous()
    kc2, vc2 = K.rope_bridge_pos(sub, tb[pick].contiguous(), *w, flag[pick].contiguous(), cos, sin, (pick % S).to(torch.int32), H)
    assert torch.equal(sub, full[pick]) and torch.equal(kc2, kc[pick]) and torch.equal(vc2, vc[pick])


@pytest.mark.parametrize("B,H", [(8, 32), (3, 2)])
def test_rope_bridge_pos_with_cache_append(K, B, H):
    """A generation step's rope_bridge_pos(append=...) == rope_bridge_pos followed by kv_cache_append, bit for bit; the rest of the
    caches untouched; the same for 2d positions."""
    from oracle import libra_oracle as LO
    D, Lmax = H * 128, 13
    cosf, sinf = LO.rope_tables(128, 64)
    cos, sin = cosf.to(BF).cuda(), sinf.to(BF).cuda()
    w = [rnd(D, 8, seed=3 + i, scale=0.3) for i in range(4)]
    flag = _flags(B, 9, "random").cuda()
    tb = torch.zeros(B, 64, dtype=BF, device="cuda"); tb[:, :16] = rnd(B, 16, seed=2)
    slot = torch.tensor([5], dtype=torch.int64, device="cuda")
    for pos in (torch.arange(B, dtype=torch.int32).cuda() * 3 + 1,
                torch.stack([torch.arange(B) * 2 + 1, torch.arange(B) + 4], 1).to(torch.int32).cuda().contiguous()):
        qkv = rnd(B, 3 * D + 64, seed=1)[:, :3 * D]                 # a column slice of the wider activation buffer, as in the engine
        a, b = qkv.clone(), qkv.clone()
        caches_a = [rnd(B, Lmax, D, seed=20 + i) for i in range(4)]
        caches_b = [c.clone() for c in caches_a]
        kc, vc = K.rope_bridge_pos(a, tb, *w, flag, cos, sin, pos, H)
        K.kv_cache_append((a[:, D:2 * D], kc, a[:, 2 * D:], vc), caches_a, slot)
        kc2, vc2 = K.rope_bridge_pos(b, tb, *w, flag, cos, sin, pos, H, append=(caches_b, slot))
        assert torch.equal(a, b) and torch.equal(kc, kc2) and torch.equal(vc, vc2)
        for x, y in zip(caches_a, caches_b):
            assert torch.equal(x, y)
    with pytest.raises(ValueError):
        K.rope_bridge_pos(b, tb, *w, flag, cos, sin, pos, H, append=(caches_b, slot.to(torch.int32)))


@pytest.mark.parametrize("B,H,Lmax,mode", [(1, 1, 16, "span"), (3, 2, 77, "random"), (2, 4, 300, "allvis"), (8, 32, 2048, "span")])
def test_bridge_attention_decode(K, B, H, Lmax, mode):
    """One query token per sequence against the KV cache == the last row of the training-path closed form."""
    D = H * 128
    g = torch.Generator().manual_seed(5)
    caches = [rnd(B * Lmax, D, seed=30 + i).view(B, Lmax, D) for i in range(4)]          # K_same, K_cross, V_same, V_cross
    q = rnd(B, D, seed=40)
    kflag = _flags(B * Lmax, 11, mode).view(B, Lmax)
    lens = torch.randint(1, Lmax + 1, (B,), generator=g).to(torch.int32)
    lens[0] = Lmax
    if B > 1:
        lens[1] = 1                                           # a sequence whose cache holds only the new token
    qflag = (torch.rand(B, generator=g) < 0.5).to(torch.uint8)
    out = K.bridge_attn_decode(q, *caches, kflag.cuda(), qflag.cuda(), lens.cuda(), H, 128 ** -0.5)
    ks, kc, vs, vc = [c.float().cpu().view(B, Lmax, H, 128) for c in caches]
    qh = q.float().cpu().view(B, H, 128)
    cross = (kflag.bool() != qflag.bool()[:, None])[:, :, None, None]
    kk, vv = torch.where(cross, kc, ks), torch.where(cross, vc, vs)
    s = torch.einsum("bhd,blhd->bhl", qh, kk) * 128 ** -0.5
    s = s.masked_fill(torch.arange(Lmax)[None, None, :] >= lens[:, None, None].long(), float("-inf"))
    ref = torch.einsum("bhl,blhd->bhd", torch.softmax(s, -1), vv).reshape(B, D)
    close(out, ref, rel=2e-3, what="bridge decode attention")
    assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("N,C,nc,tr", [(300, 256, 8, True), (16384, 4096, 8, True), (1000, 512, 16, False), (16384, 4096, 16, False),
                                       (77, 264, 8, False)])
def test_rank_outer_wgrad(K, N, C, nc, tr):
    """libra_rank_outer_wgrad (the rank-8 bridge weight gradients as one pass over x) vs fp32 math on the same bf16 inputs:
    out_m[j][c] = sum over the tokens of modality m of coef[t][j] x[t][c]; column-sliced coef, both layouts, ragged sizes."""
    x = rnd(N, C, seed=1)
    wide = rnd(N, 64, seed=2)
    coef = wide[:, 8:8 + nc] if nc == 8 else wide[:, 0:16]
    flag = _flags(N, 4, "span" if N > 500 else "random").cuda()
    ol, ov = K.rank_outer_wgrad(x, coef, flag, transpose_out=tr)
    f = flag.bool().cpu()
    xf, cf = x.float().cpu(), coef.float().cpu()
    rl, rv = cf[~f].t() @ xf[~f], cf[f].t() @ xf[f]
    if tr:
        rl, rv = rl.t(), rv.t()
    close(ol, rl, rel=2e-3, what="rank outer wgrad (text)")
    close(ov, rv, rel=2e-3, what="rank outer wgrad (vision)")
    ol2, ov2 = K.rank_outer_wgrad(x, coef, flag, transpose_out=tr)
    assert torch.equal(ol, ol2) and torch.equal(ov, ov2)                       # deterministic
    only_l, none_v = K.rank_outer_wgrad(x, coef, flag, transpose_out=tr, want_v=False)
    assert none_v is None and torch.equal(only_l, ol)
    al, _ = K.rank_outer_wgrad(x, coef, None, transpose_out=tr, want_v=False)   # no flag: every token counts for out_l
    ra = cf.t() @ xf
    close(al, ra.t() if tr else ra, rel=2e-3, what="rank outer wgrad (no flag)")


def test_kv_cache_append(K):
    """libra_kv_cache_append: the four per-layer cache appends of a decode step in one launch == four index_copy_ calls."""
    B, W, Lmax = 3, 256, 11
    wide = rnd(B, 3 * W + 64, seed=1)
    rows = (wide[:, W:2 * W], rnd(B, W, seed=2), wide[:, 2 * W:3 * W], rnd(B, W, seed=3))
    caches = [rnd(B, Lmax, W, seed=10 + i) for i in range(4)]
    ref = [c.clone() for c in caches]
    slot = torch.tensor([7], dtype=torch.int64, device="cuda")
    K.kv_cache_append(rows, caches, slot)
    for c, r, x in zip(caches, ref, rows):
        r.index_copy_(1, slot, x.reshape(B, 1, W))
        assert torch.equal(c, r)
    with pytest.raises(ValueError):
        K.kv_cache_append(rows, caches, torch.tensor([7], dtype=torch.int32, device="cuda"))
