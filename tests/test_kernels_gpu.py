"""Per-kernel parity of the gfx950 C-ABI kernels against plain fp32 math on the same bf16 inputs.

Tolerance (stated once, used everywhere): an output element `o` (bf16) must satisfy
    |o - ref| <= 1e-3 * max|ref|  +  2^-8 * |ref|
i.e. the north-star 1e-3 max-norm bound plus one bf16 ulp for the final rounding of the element itself
(a bf16 result cannot be closer than half an ulp to an fp32 reference).  Integer outputs are bit-exact.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def close(out, ref, rel=1e-3, what=""):
    out = out.float().cpu().double()
    ref = ref.float().cpu().double()
    assert out.shape == ref.shape, (what, out.shape, ref.shape)
    assert torch.isfinite(out).all(), what
    tol = rel * ref.abs().max() + (2.0 ** -8) * ref.abs()
    bad = (out - ref).abs() > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} outside tolerance, max err "
                           f"{float((out - ref).abs().max()):.4g} vs max|ref| {float(ref.abs().max()):.4g}")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF).cuda()


@pytest.fixture(scope="module")
def K():
    from libra_amd import kernels
    return kernels


@pytest.mark.parametrize("M,N,K_", [(128, 128, 64), (256, 256, 128), (100, 72, 64), (1000, 1024, 1024),
                                    (577, 3072, 1024), (130, 136, 192), (18464, 1024, 1024), (64, 4096, 1024),
                                    (1024, 1024, 18496), (512, 768, 8192), (4096, 4096, 512), (300, 520, 320)])
def test_gemm_plain(K, M, N, K_):
    a, b = rnd(M, K_, seed=1), rnd(N, K_, seed=2)     # asymmetric random operands (transpose-detecting)
    out = K.gemm_nt(a, b)
    close(out, a.float() @ b.float().t(), what=f"gemm {M}x{N}x{K_}")


@pytest.mark.parametrize("M,N,K_", [(128, 128, 64), (300, 264, 192), (1024, 1024, 18496), (3072, 1024, 2048),
                                    (18464, 1024, 1024), (577, 4096, 1024), (256, 512, 4096)])
@pytest.mark.parametrize("a_t,b_t", [(False, True), (True, False), (True, True)])
def test_gemm_transposed_operands(K, M, N, K_, a_t, b_t):
    """Reduction-major operands read with the LDS transpose loads: no transposed copy is materialised."""
    if a_t and M % 8:
        M = M // 8 * 8
    a, b = rnd(M, K_, seed=11), rnd(N, K_, seed=12)
    aa = a.t().contiguous() if a_t else a
    bb = b.t().contiguous() if b_t else b
    out = K.gemm_nt(aa, bb, a_t=a_t, b_t=b_t)
    close(out, a.float() @ b.float().t(), what=f"gemm {M}x{N}x{K_} a_t={a_t} b_t={b_t}")


def test_gemm_identity_layout(K):
    # A = I (padded), B asymmetric: catches a row/col swap in the C fragment mapping
    n = 128
    a = torch.eye(n, dtype=BF, device="cuda")
    b = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 - 125).to(BF).cuda()
    out = K.gemm_nt(a, b)
    assert torch.equal(out, b.t().contiguous())


def test_gemm_epilogues(K):
    M, N, K_ = 300, 264, 128
    a, b = rnd(M, K_, seed=3), rnd(N, K_, seed=4, scale=0.2)
    bias, res, aux = rnd(N, seed=5), rnd(M, N, seed=6), rnd(M, N, seed=7)
    base = a.float() @ b.float().t()
    close(K.gemm_nt(a, b, bias=bias), base + bias.float(), what="bias")
    close(K.gemm_nt(a, b, bias=bias, resid=res), base + bias.float() + res.float(), what="bias+resid")
    sc = torch.ones(N); sc[:100] = 0.125
    close(K.gemm_nt(a, b, bias=bias, alpha=0.125, alpha_cols=100), (base + bias.float()) * sc.cuda(), what="alpha")
    pre = torch.empty(M, N, dtype=BF, device="cuda")
    out = K.gemm_nt(a, b, bias=bias, quick_gelu=True, preact_out=pre)
    close(pre, base + bias.float(), what="preact")
    p = pre.float()
    close(out, p * torch.sigmoid(1.702 * p), what="quick_gelu(bf16 preact)")
    x = aux.float()
    s = torch.sigmoid(1.702 * x)
    close(K.gemm_nt(a, b, qgelu_grad_of=aux), base * (s * (1 + 1.702 * x * (1 - s))), what="qgelu_grad")
    # strided output / operand views (column slices of wider buffers)
    wide = torch.zeros(M, N + 64, dtype=BF, device="cuda")
    K.gemm_nt(a, b, out=wide[:, 64:])
    close(wide[:, 64:], base, what="strided out")
    assert float(wide[:, :64].abs().max()) == 0.0


def test_gemm_row_split_dispatch(K):
    """Problems whose 256^2 tiles fill 1-2 waves plus a partial one are split by rows: whole waves on the 256^2 kernel,
    the rest on the 128^2 kernel (gemm_bf16.hip plan_rows256).  Every fused operand must follow the split: row gather /
    scatter maps, residual, aux, pre-activation store, transposed A."""
    M, N, K_ = 18464, 1024, 4096                   # 73 x 4 tiles = 1.14 waves -> rows [0, 16384) + [16384, 18464)
    a, b = rnd(M, K_, seed=21, scale=0.5), rnd(N, K_, seed=22, scale=0.1)
    base = a.float() @ b.float().t()
    close(K.gemm_nt(a, b), base, what="split plain")
    bias, res = rnd(N, seed=23), rnd(M, N, seed=24)
    close(K.gemm_nt(a, b, bias=bias, resid=res), base + bias.float() + res.float(), what="split bias+resid")
    aux = rnd(M, N, seed=25)
    x = aux.float(); sg = torch.sigmoid(1.702 * x)
    close(K.gemm_nt(a, b, qgelu_grad_of=aux), base * (sg * (1 + 1.702 * x * (1 - sg))), what="split qgelu_grad")
    pre = torch.empty(M, N, dtype=BF, device="cuda")
    out = K.gemm_nt(a, b, bias=bias, quick_gelu=True, preact_out=pre)
    close(pre, base + bias.float(), what="split preact")
    close(out, pre.float() * torch.sigmoid(1.702 * pre.float()), what="split quick_gelu")
    # transposed operands
    close(K.gemm_nt(a.t().contiguous(), b.t().contiguous(), a_t=True, b_t=True), base, what="split TT")
    # routed: gather A rows from a taller buffer, scatter C rows into a taller buffer with a residual read there
    g = torch.Generator().manual_seed(5)
    phys = M + 1000
    rows = torch.randperm(phys, generator=g)[:M].to(torch.int32).cuda()
    abig = rnd(phys, K_, seed=26, scale=0.5)
    rbig = rnd(phys, N, seed=27)
    cbig = torch.zeros(phys, N, dtype=BF, device="cuda")
    K.gemm_nt(abig, b, out=cbig, a_rows=rows, c_rows=rows, resid=rbig)
    ref = abig[rows.long()].float() @ b.float().t() + rbig[rows.long()].float()
    close(cbig[rows.long()], ref, what="split routed")
    untouched = torch.ones(phys, dtype=torch.bool, device="cuda"); untouched[rows.long()] = False
    assert float(cbig[untouched].abs().max()) == 0.0


TILES = {"128": 1, "256": 2, "W": 3}      # include/libra_hip.h LIBRA_GEMM_TILE_*


@pytest.mark.parametrize("tile", ["128", "256", "W"])
@pytest.mark.parametrize("M,N,K_,a_t,b_t", [(256, 128, 64, False, False), (300, 264, 192, False, False), (1000, 1024, 1024, False, False),
                                            (4624, 1024, 1024, False, False), (4624, 4096, 1024, False, True),
                                            (1024, 1024, 4672, True, True), (2752, 520, 1152, True, False), (130, 136, 192, True, True),
                                            (976, 4096, 2048, False, True), (577, 200, 128, False, False), (8, 72, 64, False, False),
                                            # 20 x 28 = 560 tiles of 256^2 on <= 256 persistent workgroups: 2-3 tiles per workgroup, ragged edges,
                                            # an odd K-tile count (the prefetched next tile lands in the other K-tile buffer)
                                            (5000, 7000, 192, False, False), (5000, 7000, 320, True, True)])
def test_gemm_every_tile_structure(K, M, N, K_, a_t, b_t, tile):
    """The three tile structures (128^2 x two per CU, 256^2 x one per CU, W = 256x128 x two per CU) share one contract; the
    caller-pinned entry point runs each of them on ragged, transposed and multi-K-tile problems against fp32 math.  The three
    also accumulate every output element in the same order (K tiles ascending, 16-deep MFMA steps ascending), so their results
    are bit-identical - which pins the W kernel's fragment / accumulator mapping to the two older structures'."""
    if a_t and M % 8:
        M = M // 8 * 8                                 # a reduction-major A needs a 16-byte aligned row start per k
    if b_t and N % 8:
        N = N // 8 * 8
    a, b = rnd(M, K_, seed=61), rnd(N, K_, seed=62)
    aa = a.t().contiguous() if a_t else a
    bb = b.t().contiguous() if b_t else b
    out = K.gemm_nt(aa, bb, a_t=a_t, b_t=b_t, tile=TILES[tile])
    close(out, a.float() @ b.float().t(), what=f"gemm[{tile}] {M}x{N}x{K_} a_t={a_t} b_t={b_t}")
    assert torch.equal(out, K.gemm_nt(aa, bb, a_t=a_t, b_t=b_t, tile=TILES["128"])), f"tile {tile} != tile 128 bitwise"


@pytest.mark.parametrize("tile", ["128", "256", "W"])
def test_gemm_every_tile_structure_fused_operands(K, tile):
    """Epilogue operands, row maps, strided views and grouped launches through each pinned structure (the W kernel shares the
    256^2 kernel's per-wave epilogue code: gemm_epilogue.hpp)."""
    t = TILES[tile]
    M, N, K_ = 1100, 776, 320                                  # 5 x 7 W tiles with ragged last row / column tiles
    a, b = rnd(M, K_, seed=3), rnd(N, K_, seed=4, scale=0.2)
    bias, res, aux = rnd(N, seed=5), rnd(M, N, seed=6), rnd(M, N, seed=7)
    base = a.float() @ b.float().t()
    close(K.gemm_nt(a, b, bias=bias, resid=res, tile=t), base + bias.float() + res.float(), what="bias+resid")
    sc = torch.ones(N); sc[:100] = 0.125
    close(K.gemm_nt(a, b, bias=bias, alpha=0.125, alpha_cols=100, tile=t), (base + bias.float()) * sc.cuda(), what="alpha")
    pre = torch.empty(M, N, dtype=BF, device="cuda")
    out = K.gemm_nt(a, b, bias=bias, quick_gelu=True, preact_out=pre, tile=t)
    close(pre, base + bias.float(), what="preact")
    close(out, pre.float() * torch.sigmoid(1.702 * pre.float()), what="quick_gelu(bf16 preact)")
    x = aux.float(); sg = torch.sigmoid(1.702 * x)
    close(K.gemm_nt(a, b, qgelu_grad_of=aux, tile=t), base * (sg * (1 + 1.702 * x * (1 - sg))), what="qgelu_grad")
    wide = torch.zeros(M, N + 64, dtype=BF, device="cuda")
    K.gemm_nt(a, b, out=wide[:, 64:], tile=t)
    close(wide[:, 64:], base, what="strided out")
    assert float(wide[:, :64].abs().max()) == 0.0
    # routed: gather A rows from a taller buffer, scatter C rows into a taller buffer with a residual read there
    g = torch.Generator().manual_seed(5)
    phys = M + 333
    rows = torch.randperm(phys, generator=g)[:M].to(torch.int32).cuda()
    abig, rbig = rnd(phys, K_, seed=26, scale=0.5), rnd(phys, N, seed=27)
    cbig = torch.zeros(phys, N, dtype=BF, device="cuda")
    K.gemm_nt(abig, b, out=cbig, a_rows=rows, c_rows=rows, resid=rbig, tile=t)
    close(cbig[rows.long()], abig[rows.long()].float() @ b.float().t() + rbig[rows.long()].float(), what="routed")
    untouched = torch.ones(phys, dtype=torch.bool, device="cuda"); untouched[rows.long()] = False
    assert float(cbig[untouched].abs().max()) == 0.0
    # grouped: three problems in one launch, reduction-major B, scatter map
    G = 3
    ag = rnd(M, G * K_, seed=31, scale=0.5)
    a_list = [ag[:, i * K_:(i + 1) * K_] for i in range(G)]
    b_list = [rnd(K_, N, seed=40 + i, scale=0.2) for i in range(G)]
    cg = torch.zeros(phys, G * N, dtype=BF, device="cuda")
    outs = [cg[:, i * N:(i + 1) * N] for i in range(G)]
    K.gemm_nt_grouped(a_list, b_list, outs, b_t=True, c_rows=rows, tile=t)
    for i in range(G):
        close(outs[i][rows.long()], a_list[i].float() @ b_list[i].float(), what=f"grouped group {i}")
    assert float(cg[untouched].abs().max()) == 0.0


def test_gemm_rejects_bad_shapes(K):
    a, b = rnd(64, 96), rnd(64, 96)
    with pytest.raises(ValueError):
        K.gemm_nt(a, b)                       # K % 64 != 0
    with pytest.raises(ValueError):
        K.gemm_nt(rnd(64, 128), rnd(64, 64))  # inner dims differ
    out = K.gemm_nt(rnd(0, 64), rnd(8, 64))   # empty problem is a no-op, like torch
    assert out.shape == (0, 8)


@pytest.mark.parametrize("rows,D", [(7, 128), (1000, 1024), (33, 2048), (18, 4096), (5, 64)])
def test_layernorm_fwd_bwd(K, rows, D):
    x, g, b, dy, dres = rnd(rows, D, seed=1), rnd(D, seed=2) * 0.1 + 1, rnd(D, seed=3), rnd(rows, D, seed=4), rnd(rows, D, seed=5)
    g = g.to(BF)
    y, mean, rstd = K.layernorm_fwd(x, g, b, 1e-5)
    xf = x.float().requires_grad_(True); gf = g.float().requires_grad_(True); bf = b.float().requires_grad_(True)
    ref = F.layer_norm(xf, (D,), gf, bf, 1e-5)
    close(y, ref.detach(), what="ln fwd")
    ref.backward(dy.float())
    dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
    dx = K.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, dgamma=dg, dbeta=db)
    close(dx, xf.grad + dres.float(), what="ln dx")
    close(K.f32_to_bf16(dg), gf.grad, rel=2e-3, what="ln dgamma")
    close(K.f32_to_bf16(db), bf.grad, rel=2e-3, what="ln dbeta")
    # fused column sum of the output (= the bias gradient of the Linear that dx is the dY of): exactly the sum of the
    # stored bf16 values, and the other outputs are unchanged by asking for it
    dg2 = torch.zeros(D, device="cuda"); db2 = torch.zeros(D, device="cuda"); ds = torch.full((D,), 3.0, device="cuda")
    dx2 = K.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, dgamma=dg2, dbeta=db2, dxsum=ds)
    assert torch.equal(dx2, dx) and torch.equal(dg2, dg) and torch.equal(db2, db)
    ref_sum = dx.float().sum(0) + 3.0
    assert float((ds - ref_sum).abs().max()) <= 1e-4 * max(1.0, float(ref_sum.abs().max()))


@pytest.mark.parametrize("rows,cols", [(577, 200), (18464, 1024), (100, 4096), (3, 8)])
def test_colsum(K, rows, cols):
    x = rnd(rows, cols, seed=rows)
    out = torch.ones(cols, device="cuda")
    K.colsum(x, out)
    close(out, 1.0 + x.float().sum(0), rel=1e-5, what="colsum")
    out2 = torch.ones(cols, device="cuda")
    K.colsum(x, out2)
    assert torch.equal(out, out2)          # deterministic (no atomics)


def test_patch_im2col_col2im(K):
    B, C, H, P = 2, 3, 56, 14
    pix = rnd(B, C, H, H, seed=11)
    cols = K.patch_im2col(pix, P, 640)
    g = H // P
    ref = pix.reshape(B, C, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, C * P * P)
    assert torch.equal(cols[:, :588], ref)
    assert float(cols[:, 588:].abs().max()) == 0.0
    back = K.patch_col2im(cols, B, C, H, H, P)
    assert torch.equal(back, pix)


def test_embed_ln(K):
    B, T, D = 3, 17, 128
    patches, cls, pos = rnd(B * (T - 1), D, seed=1), rnd(D, seed=2), rnd(T, D, seed=3)
    g, b = (rnd(D, seed=4) * 0.1 + 1).to(BF), rnd(D, seed=5)
    emb, hs0, mean, rstd = K.vit_embed_ln(patches, cls, pos, g, b, B, T, 1e-5)
    e = torch.cat([cls.float().expand(B, 1, D), patches.float().view(B, T - 1, D)], 1) + pos.float()
    e = e.to(BF)
    assert torch.equal(emb.view(B, T, D), e)
    close(hs0.view(B, T, D), F.layer_norm(e.float(), (D,), g.float(), b.float(), 1e-5), what="pre-LN")


def _attn_ref(qkv, B, T, H, scale):
    D = H * 64
    q, k, v = [t.float().view(B, T, H, 64).transpose(1, 2) for t in qkv.split(D, dim=1)]
    s = (q @ k.transpose(-1, -2)) * scale
    p = torch.softmax(s, -1)
    o = (p @ v).transpose(1, 2).reshape(B * T, D)
    return o, torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,T,H", [(2, 17, 2), (1, 64, 1), (2, 65, 2), (1, 128, 1), (2, 129, 3), (2, 577, 16)])
def test_attention_fwd(K, B, T, H):
    D = H * 64
    qkv = rnd(B * T, 3 * D, seed=T)
    Tp = K.round_up(T, 64)
    o, lse = K.vit_attn_fwd(qkv, B, T, H, 0.125)
    ro, rl = _attn_ref(qkv, B, T, H, 0.125)
    close(o, ro, rel=2e-3, what="attn out")        # P is fed to the MFMA in bf16 (as the reference's bmm does)
    close(lse, rl, rel=1e-4, what="lse")


def test_attention_fwd_spiky_rows(K):
    # one huge logit per row forces the online-softmax rescale path hard (max jumps by >> 8 at a late tile)
    B, T, H = 1, 200, 1
    qkv = rnd(B * T, 192, seed=3)
    qkv[:, 64:128][150] = qkv[:, 64:128][150] * 40
    Tp = 256
    o, lse = K.vit_attn_fwd(qkv, B, T, H, 0.125)
    ro, rl = _attn_ref(qkv, B, T, H, 0.125)
    close(o, ro, rel=2e-3, what="attn out spiky")
    close(lse, rl, rel=1e-4, what="lse spiky")


@pytest.mark.parametrize("B,T,H", [(2, 17, 2), (1, 64, 1), (2, 129, 2), (1, 577, 4)])
def test_attention_bwd(K, B, T, H):
    D = H * 64
    qkv = rnd(B * T, 3 * D, seed=T + 1)
    do = rnd(B * T, D, seed=T + 2)
    Tp = K.round_up(T, 64)
    o, lse = K.vit_attn_fwd(qkv, B, T, H, 0.125)
    dqkv = K.vit_attn_bwd(qkv, o, do, lse, B, T, H, 0.125)
    ref_in = qkv.float().requires_grad_(True)
    ro, _ = _attn_ref(ref_in, B, T, H, 0.125)
    ro.backward(do.float())
    for j, n in enumerate("qkv"):
        close(dqkv[:, j * D:(j + 1) * D], ref_in.grad[:, j * D:(j + 1) * D], rel=4e-3, what=f"d{n}")


def test_attention_output_residual_sharpens_the_backward(K):
    """out_lo = bf16(O - bf16(O)): O + out_lo reproduces the fp32 output to ~2^-15, and the backward's D = rowsum(dO O) taken from
    it brings dq / dk (difference-sensitive: dS = P (dP - D)) several times closer to the fp32 reference than the bf16 O alone."""
    B, T, H = 1, 577, 4
    D = H * 64
    qkv = rnd(B * T, 3 * D, seed=5)
    do = rnd(B * T, D, seed=6)
    o_lo = torch.empty(B * T, D, dtype=BF, device="cuda")
    o, lse = K.vit_attn_fwd(qkv, B, T, H, 0.125, out_lo=o_lo)
    o_plain, _ = K.vit_attn_fwd(qkv, B, T, H, 0.125)
    assert torch.equal(o, o_plain)
    ref_in = qkv.float().requires_grad_(True)
    ro, _ = _attn_ref(ref_in, B, T, H, 0.125)
    e_hi = float((o.float().cpu() - ro.detach().cpu()).abs().max())
    e_both = float(((o.float() + o_lo.float()).cpu() - ro.detach().cpu()).abs().max())
    ro.backward(do.float())
    g_plain = K.vit_attn_bwd(qkv, o, do, lse, B, T, H, 0.125)
    g_lo = K.vit_attn_bwd(qkv, o, do, lse, B, T, H, 0.125, out_lo=o_lo)
    from helpers import rel_err
    ref = ref_in.grad.cpu()
    ep, el = rel_err(g_plain[:, :D].float().cpu(), ref[:, :D]), rel_err(g_lo[:, :D].float().cpu(), ref[:, :D])
    print(f"O error {e_hi:.2e} -> {e_both:.2e} with the residual; dq rel err {ep:.2e} -> {el:.2e}")
    # (the kernel's fp32 O itself carries the bf16 rounding of P: the residual removes the OUTPUT rounding, ~40 % of the error here)
    assert e_both < 0.8 * e_hi and el <= ep * 1.05 + 1e-4
    close(g_lo[:, 2 * D:], ref_in.grad[:, 2 * D:], rel=4e-3, what="dv")


def test_feature_select(K):
    B, T, D = 2, 17, 128
    a, b = rnd(B * T, D, seed=1), rnd(B * T, D, seed=2)
    f = K.feature_select([a, b], B, T)
    ref = torch.cat([a.view(B, T, D), b.view(B, T, D)], -1)[:, 1:].reshape(B * (T - 1), 2 * D)
    assert torch.equal(f, ref)
    da, db = torch.empty_like(a), rnd(B * T, D, seed=3)
    db0 = db.clone()
    K.feature_select_bwd(f, [da, db], [False, True], B, T)
    assert float(da.view(B, T, D)[:, 0].abs().max()) == 0.0
    assert torch.equal(da.view(B, T, D)[:, 1:], a.view(B, T, D)[:, 1:])
    close(db.view(B, T, D)[:, 1:], (db0.float() + b.float()).view(B, T, D)[:, 1:], what="accumulate")


@pytest.mark.parametrize("E", [18, 32, 512])
def test_lfq_encode(K, E):
    from oracle import vq_oracle as QO
    B, hw, Q = 3, 16, 2
    h = rnd(B * hw, E, seed=E)
    sd = QO.random_vq_state_dict(c_feat=8, embed_dim=E, dtype=BF)
    dev = {k: v.cuda() for k, v in sd.items()}
    idx, ids, xpre, quant = K.lfq_encode(h, dev.get("quantize.project_in.weight"), dev.get("quantize.project_in.bias"),
                                         dev.get("quantize.project_out.weight"), dev.get("quantize.project_out.bias"),
                                         B=B, hw=hw, Q=Q, offset=32000, boi=32512, eoi=32513, want_ids=True,
                                         want_xpre=True, want_quant=True)
    # oracle in float64 with the same rounding point (x rounded to bf16 before the sign test)
    hd = h.cpu().double()
    if E != 18:
        x = hd @ sd["quantize.project_in.weight"].double().t() + sd["quantize.project_in.bias"].double()
    else:
        x = hd
    xb = x.float().to(BF)
    bits = (xb.float() > 0).view(B * hw, Q, 9)
    mask = (2 ** torch.arange(8, -1, -1)).long()
    ref_idx = (bits.long() * mask).sum(-1)
    margin = x.abs().view(B * hw, Q, 9)
    got = idx.cpu()
    mism = got != ref_idx
    if mism.any():
        # a disagreement is only admissible when the fp64 pre-sign value is within fp32 accumulation noise of 0
        rows = mism.nonzero()
        for r, q in rows.tolist():
            gb = ((got[r, q] >> torch.arange(8, -1, -1)) & 1).bool()
            diff = gb != bits[r, q]
            assert float(margin[r, q][diff].max()) < 1e-5, ("LFQ bit flipped with a non-negligible margin", r, q)
    assert int(mism.sum()) <= 1, f"{int(mism.sum())} index mismatches"
    assert torch.equal(ids[:, :, 0].cpu(), torch.full((Q, B), 32512))
    assert torch.equal(ids[:, :, -1].cpu(), torch.full((Q, B), 32513))
    assert torch.equal(ids[:, :, 1:-1].cpu(), got.view(B, hw, Q).permute(2, 0, 1) + 32000)
    sign = torch.where(((got.unsqueeze(-1) >> torch.arange(8, -1, -1)) & 1).bool(), 1.0, -1.0).view(B * hw, Q * 9)
    if E != 18:
        rq = sign.double() @ sd["quantize.project_out.weight"].double().t() + sd["quantize.project_out.bias"].double()
        close(quant, rq.float(), what="quant")
        close(xpre, x.float(), what="xpre")
    else:
        assert torch.equal(quant.float().cpu(), sign)


@pytest.mark.parametrize("M,N,K_,G,b_t", [(300, 264, 128, 3, False), (4624, 4096, 1024, 3, False), (4624, 1024, 4096, 3, True),
                                          (4624, 2816, 1408, 2, False), (130, 72, 64, 4, True)])
def test_gemm_grouped(K, M, N, K_, G, b_t):
    """Several same-shape GEMMs in one launch (blockIdx.z = group), incl. column-slice operands, the row-split dispatch
    and a scatter map."""
    abig = rnd(M, G * K_, seed=31, scale=0.5)
    a_list = [abig[:, g * K_:(g + 1) * K_] for g in range(G)]              # column slices of one buffer (shared row stride)
    b_list = [rnd(*((K_, N) if b_t else (N, K_)), seed=40 + g, scale=0.2) for g in range(G)]
    phys = M + 77
    gen = torch.Generator().manual_seed(9)
    rows = torch.randperm(phys, generator=gen)[:M].to(torch.int32).cuda()
    cbig = torch.zeros(phys, G * N, dtype=BF, device="cuda")
    outs = [cbig[:, g * N:(g + 1) * N] for g in range(G)]
    K.gemm_nt_grouped(a_list, b_list, outs, b_t=b_t, c_rows=rows)
    for g in range(G):
        ref = a_list[g].float() @ (b_list[g].float() if b_t else b_list[g].float().t())
        close(outs[g][rows.long()], ref, what=f"grouped gemm group {g}")
    untouched = torch.ones(phys, dtype=torch.bool, device="cuda"); untouched[rows.long()] = False
    assert float(cbig[untouched].abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K_", [(4096, 8, 11776), (64, 4096, 4672), (8, 8, 4096), (520, 16, 8192)])
def test_gemm_splitk_skinny(K, M, N, K_):
    """Skinny weight-gradient shapes (rank-8 bridges) go through the K-sliced 256^2 kernel + deterministic slab reduce."""
    a, b = rnd(K_, M, seed=51, scale=0.3), rnd(K_, N, seed=52, scale=0.3)           # both reduction-major, as wgrad operands are
    out = K.gemm_nt(a, b, a_t=True, b_t=True)
    close(out, a.float().t() @ b.float(), rel=2e-3, what=f"skinny split-K {M}x{N}x{K_}")
    out2 = K.gemm_nt(a, b, a_t=True, b_t=True)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("M,N,K_,b_t", [(976, 1024, 8192, False), (976, 4096, 11008, True), (1156, 1024, 11008, False), (200, 520, 4096, True)])
def test_gemm_splitk_routed_few_rows(K, M, N, K_, b_t):
    """Few output rows, long reduction (the text-stream projections of a 700-token pretraining step: 976 text rows; the vision
    projections of the instruction recipe: 1156 rows - M is not a multiple of 8): K-sliced launch + deterministic slab reduction with
    the row gather, the row scatter and the residual of the routed GEMM, against fp32 math and the unsplit pinned structure."""
    from libra_amd import _lib
    assert _lib.lib().libra_gemm_splitk_plan(M, N, K_) > 1, "the planner does not slice this shape: the test would not reach the path"
    g = torch.Generator().manual_seed(7)
    phys = M + 211
    rows = torch.randperm(phys, generator=g)[:M].to(torch.int32).cuda()
    abig, rbig = rnd(phys, K_, seed=71, scale=0.5), rnd(phys, N, seed=72)
    b = rnd(N, K_, seed=73, scale=0.1)
    bb = b.t().contiguous() if b_t else b
    ref = abig[rows.long()].float() @ b.float().t()
    for resid in (None, rbig):
        cbig = torch.zeros(phys, N, dtype=BF, device="cuda")
        K.gemm_nt(abig, bb, out=cbig, b_t=b_t, a_rows=rows, c_rows=rows, resid=resid)
        want = ref + (rbig[rows.long()].float() if resid is not None else 0.0)
        close(cbig[rows.long()], want, rel=2e-3, what=f"routed split-K {M}x{N}x{K_} resid={resid is not None}")
        untouched = torch.ones(phys, dtype=torch.bool, device="cuda"); untouched[rows.long()] = False
        assert float(cbig[untouched].abs().max()) == 0.0
        c2 = torch.zeros(phys, N, dtype=BF, device="cuda")
        K.gemm_nt(abig, bb, out=c2, b_t=b_t, a_rows=rows, c_rows=rows, resid=resid)
        assert torch.equal(cbig, c2)                                            # deterministic reduction
        c3 = torch.zeros(phys, N, dtype=BF, device="cuda")                   # the unsplit 128^2 structure on the same problem
        K.gemm_nt(abig, bb, out=c3, b_t=b_t, a_rows=rows, c_rows=rows, resid=resid, tile=1)
        close(cbig[rows.long()], c3[rows.long()].float(), rel=5e-3, what="split vs unsplit")      # two bf16 roundings: one ulp apart at most
    # plain problem whose M is not a multiple of 8 (was excluded from slicing by an over-strict plan)
    a = rnd(M, K_, seed=74, scale=0.5)
    close(K.gemm_nt(a, bb, b_t=b_t), a.float() @ b.float().t(), rel=2e-3, what="plain split-K, ragged M")


@pytest.mark.parametrize("M,N,K_", [(1, 8, 64), (3, 200, 256), (8, 4096, 4096), (8, 1024, 11008), (16, 520, 3136), (13, 64, 512)])
def test_gemm_skinny_rows(K, M, N, K_):
    """M <= 16 (the generation step) goes to the skinny HBM-bound kernel: plain, residual, row gather / scatter."""
    a, b = rnd(M, K_, seed=41, scale=0.5), rnd(N, K_, seed=42, scale=0.1)
    base = a.float() @ b.float().t()
    close(K.gemm_nt(a, b), base, what="skinny plain")
    res = rnd(M, N, seed=43)
    close(K.gemm_nt(a, b, resid=res), base + res.float(), what="skinny resid")
    # routed: A rows gathered from a taller buffer, C rows scattered into a taller buffer whose other rows stay untouched
    phys = M + 5
    rows = torch.randperm(phys, generator=torch.Generator().manual_seed(7))[:M].to(torch.int32).cuda()
    abig, rbig = rnd(phys, K_, seed=44, scale=0.5), rnd(phys, N, seed=45)
    cbig = torch.full((phys, N), 3.0, dtype=BF, device="cuda")
    K.gemm_nt(abig, b, out=cbig, a_rows=rows, c_rows=rows, resid=rbig)
    ref = abig[rows.long()].float() @ b.float().t() + rbig[rows.long()].float()
    close(cbig[rows.long()], ref, what="skinny routed")
    untouched = torch.ones(phys, dtype=torch.bool, device="cuda"); untouched[rows.long()] = False
    assert torch.equal(cbig[untouched], torch.full((phys - M, N), 3.0, dtype=BF, device="cuda"))
    # and it agrees with the tiled kernels on the same problem (M = 17 rows takes the MFMA path; compare the shared rows)
    if M == 16:
        a17 = torch.cat([a, rnd(1, K_, seed=46)], 0)
        close(K.gemm_nt(a, b), K.gemm_nt(a17, b)[:16].float(), what="skinny vs tiled")
