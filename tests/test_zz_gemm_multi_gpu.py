"""libra_gemm_bf16_multi: several independent GEMMs as one persistent launch over a common tile list (round 6).

The contract is "each problem computes what its own libra_gemm_bf16_nt_routed call would, bit for bit" - so the gate is
`torch.equal` against the single-problem launches (whose own parity against fp32 math is tests/test_kernels_gpu.py), plus the
1e-3 + 1 ulp bound against fp32 math directly, plus the properties of the tile queue: the 128-byte workspace is all zero again
after every launch, results do not change from launch to launch, and a launch with fewer tiles than compute units works.

(File name: the `zz` sorts these tests LAST.  The kernel was written in a round whose GPU pool was closed from the first call on, so
its first execution anywhere is this file; should it fault, it must not take the rest of `pytest -m gpu -x` down with it.)"""
import os

import pytest
import torch

from test_kernels_gpu import close, rnd

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def K():
    from libra_amd import kernels
    return kernels


def _ws_zero(K):
    torch.cuda.synchronize()
    return all(int(ws.abs().sum()) == 0 for ws in K._MULTI_WS.values())


def _problem(K, M, N, Kd, a_t, b_t, seed, *, routed=False, resid=False, bias=False):
    """-> (kwargs of gemm_nt / gemm_spec, fp32 reference of the rows written, out tensor to compare)"""
    a, b = rnd(M, Kd, seed=seed), rnd(N, Kd, seed=seed + 1)
    ref = a.float() @ b.float().t()
    kw = dict(a_t=a_t, b_t=b_t)
    aa = a.t().contiguous() if a_t else a
    bb = b.t().contiguous() if b_t else b
    out_rows = M
    if routed:                       # A gathered from a taller buffer, C scattered into a taller buffer (disjoint from everything else)
        phys = M + 37
        g = torch.Generator().manual_seed(seed + 2)
        perm = torch.randperm(phys, generator=g)[:M].to(torch.int32).cuda()
        big = rnd(phys, Kd, seed=seed + 3)
        big[perm.long()] = a
        aa = big
        kw.update(a_rows=perm, c_rows=perm)
        out_rows = phys
    if bias:
        bv = rnd(N, seed=seed + 4)
        kw.update(bias=bv)
        ref = ref + bv.float()
    r = None
    if resid:
        r = rnd(out_rows, N, seed=seed + 5)
        kw.update(resid=r)
    return aa, bb, kw, ref, out_rows, r


CASES = [
    # (M, N, K, a_t, b_t, routed, resid, bias)
    (1000, 1024, 1024, False, False, True, True, False),
    (520, 264, 192, False, True, False, False, False),
    (512, 768, 2048, True, True, False, False, False),
    (300, 520, 320, False, False, False, False, True),
    (256, 256, 64, True, False, False, False, False),
    (2304, 1024, 512, False, False, True, False, False),
    (72, 136, 128, False, True, False, True, False),
]


def _run_cases(K, cases, seed0=100):
    specs, single, refs = [], [], []
    for i, (M, N, Kd, a_t, b_t, routed, resid, bias) in enumerate(cases):
        aa, bb, kw, ref, out_rows, r = _problem(K, M, N, Kd, a_t, b_t, seed0 + 10 * i, routed=routed, resid=resid, bias=bias)
        o1 = torch.zeros((out_rows, N), dtype=BF, device="cuda")
        o2 = torch.zeros((out_rows, N), dtype=BF, device="cuda")
        specs.append(K.gemm_spec(aa, bb, out=o1, **kw))
        # the single launch of the SAME tile body (256 x 256) where the problem is big enough for it; smaller problems run the
        # 128 x 128 structure there (same accumulation order, its own epilogue code) and are compared through fp32 math only
        same_body = M >= 256 and N >= 256
        K.gemm_nt(aa, bb, out=o2, tile=K.GEMM_TILE_256 if same_body else K.GEMM_TILE_AUTO, **kw)
        single.append(o2 if same_body else None)
        rows = kw["c_rows"].long() if routed else slice(None)
        refs.append((ref + (r.float()[rows] if r is not None else 0.0), rows))
    outs = K.gemm_multi(specs)
    return outs, single, refs


def test_multi_equals_single_launches_bit_for_bit(K):
    outs, single, refs = _run_cases(K, CASES)
    for i, (o, s, (ref, rows)) in enumerate(zip(outs, single, refs)):
        assert s is None or torch.equal(o, s), f"problem {i} {CASES[i]}: multi launch differs from its own gemm_nt launch"
        close(o[rows], ref, what=f"problem {i} {CASES[i]} vs fp32")
    assert _ws_zero(K), "the tile-queue workspace must be all zero after the launch"


def test_multi_is_repeatable_and_leaves_the_queue_clear(K):
    first = None
    for _ in range(4):
        outs, _, _ = _run_cases(K, CASES)
        assert _ws_zero(K)
        if first is None:
            first = [o.clone() for o in outs]
        else:
            assert all(torch.equal(a, b) for a, b in zip(first, outs))


def test_multi_more_problems_than_one_launch_holds_and_tiny_launches(K):
    from libra_amd import _lib
    cases = [(256 + 8 * i, 264, 64 * (1 + i % 3), False, bool(i & 1), False, False, False) for i in range(_lib.GEMM_MULTI_MAX + 3)]
    outs, single, refs = _run_cases(K, cases, seed0=500)
    assert all(s is not None and torch.equal(o, s) for o, s in zip(outs, single))
    # one tile, one problem: fewer entries than compute units
    outs, single, refs = _run_cases(K, [(100, 72, 64, False, False, False, False, False)], seed0=900)
    close(outs[0], refs[0][0], what="one-tile launch vs fp32")
    assert _ws_zero(K)
    assert K.gemm_multi([]) == []


def test_multi_disjoint_row_scatter_into_one_tensor(K):
    """Two problems writing DISJOINT rows of the same output through their row maps - the decoder's text / vision routing
    (cal_language_vision, modeling_libra.py:111-147): text rows from a dense weight, vision rows from a low-rank expansion."""
    n, H, r = 1536, 512, 128
    g = torch.Generator().manual_seed(7)
    perm = torch.randperm(n, generator=g)
    lang, vis = perm[:1000].sort().values.to(torch.int32).cuda(), perm[1000:].sort().values.to(torch.int32).cuda()
    h, w, t, wb = rnd(n, H, seed=1), rnd(H, H, seed=2), rnd(vis.numel(), r, seed=3), rnd(H, r, seed=4)
    x = rnd(n, H, seed=5)
    out = torch.zeros((n, H), dtype=BF, device="cuda")
    K.gemm_multi([K.gemm_spec(h, w, out=out, a_rows=lang, c_rows=lang, resid=x),
                  K.gemm_spec(t, wb, out=out, c_rows=vis, resid=x)])
    ref = torch.zeros((n, H), dtype=BF, device="cuda")
    K.gemm_nt(h, w, out=ref, a_rows=lang, c_rows=lang, resid=x, tile=K.GEMM_TILE_256)
    K.gemm_nt(t, wb, out=ref, c_rows=vis, resid=x, tile=K.GEMM_TILE_256)
    assert torch.equal(out, ref)
    close(out[lang.long()], h[lang.long()].float() @ w.float().t() + x[lang.long()].float(), what="text rows")
    close(out[vis.long()], t.float() @ wb.float().t() + x[vis.long()].float(), what="vision rows")


def test_multi_decoder_stage_shapes_full_size(K):
    """The launch groups the decoder engine makes at the benchmark shape (11 760 text rows, 4 624 vision rows): text q|k|v + the
    three vision expansions; a dgrad with a weight gradient riding along.  Bit-equal to the separate launches."""
    nl, nv, H, r = 11760, 4624, 4096, 1024
    n = nl + nv
    g = torch.Generator().manual_seed(3)
    perm = torch.randperm(n, generator=g)
    lang, vis = perm[:nl].sort().values.to(torch.int32).cuda(), perm[nl:].sort().values.to(torch.int32).cuda()
    h = rnd(n, H, seed=1, scale=0.5)
    wq = rnd(3 * H + 64, H, seed=2, scale=0.02)
    t = rnd(nv, 3 * r, seed=3, scale=0.5)
    wb = [rnd(H, r, seed=4 + j, scale=0.02) for j in range(3)]
    outs = []
    for multi in (True, False):
        qkv = torch.zeros((n, 3 * H + 64), dtype=BF, device="cuda")
        # text rows: every column from the packed dense weight; vision rows: q / k / v columns from the three rank-r expansions
        probs = [dict(a=h, b=wq, out=qkv, a_rows=lang, c_rows=lang)] + \
                [dict(a=t[:, j * r:(j + 1) * r], b=wb[j], out=qkv[:, j * H:(j + 1) * H], c_rows=vis) for j in range(3)]
        if multi:
            K.gemm_multi([K.gemm_spec(p.pop("a"), p.pop("b"), **p) for p in probs])
        else:
            for p in probs:
                K.gemm_nt(p.pop("a"), p.pop("b"), tile=K.GEMM_TILE_256, **p)
        outs.append(qkv)
    assert torch.equal(outs[0], outs[1])
    # dgrad (bT) + weight gradient (aT bT) in one launch
    dy, w2, x2 = rnd(nv, H, seed=11, scale=0.5), rnd(H, r, seed=12, scale=0.02), K.alloc_rows(nv, r, "cuda")
    x2[:nv] = rnd(nv, r, seed=13, scale=0.5)
    dyp = K.alloc_rows(nv, H, "cuda")
    dyp[:nv] = dy
    res = []
    for multi in (True, False):
        dt = torch.zeros((nv, r), dtype=BF, device="cuda")
        dw = torch.zeros((H, r), dtype=BF, device="cuda")
        if multi:
            K.gemm_multi([K.gemm_spec(dy, w2, out=dt, b_t=True), K.gemm_spec(dyp, x2, out=dw, a_t=True, b_t=True)])
        else:
            K.gemm_nt(dy, w2, out=dt, b_t=True, tile=K.GEMM_TILE_256)
            K.gemm_nt(dyp, x2, out=dw, a_t=True, b_t=True, tile=K.GEMM_TILE_256)
        res.append((dt, dw))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    close(res[0][1], dy.float().t() @ x2[:nv].float(), what="weight gradient in a multi launch")
    assert _ws_zero(K)


def test_multi_k_sliced_problems(K):
    """splitk > 1: a weight-gradient-shaped problem (few tiles, long K) cut into K slices that ride in the same tile list as an
    unsplit problem; equal, bit for bit, to libra_gemm_bf16_nt_splitk with the same number of slices (same slices, same
    slice-order fp32 reduction), and within the kernel tolerance of fp32 math."""
    from libra_amd import _lib
    M, N, Kd, S = 512, 1024, 8192, 4
    a, b = rnd(M, Kd, seed=21, scale=0.5), rnd(N, Kd, seed=22, scale=0.5)
    at, bt = a.t().contiguous(), b.t().contiguous()                    # reduction-major, as a weight gradient's operands
    x, w = rnd(1000, 512, seed=23), rnd(768, 512, seed=24)
    o_split = torch.zeros((M, N), dtype=BF, device="cuda")
    o_plain = torch.zeros((1000, 768), dtype=BF, device="cuda")
    r = rnd(M, N, seed=25)
    o_res = torch.zeros((M, N), dtype=BF, device="cuda")
    K.gemm_multi([K.gemm_spec(at, bt, out=o_split, a_t=True, b_t=True, splitk=S), K.gemm_spec(x, w, out=o_plain),
                  K.gemm_spec(a, b, out=o_res, resid=r, splitk=3)])
    ref = torch.zeros((M, N), dtype=BF, device="cuda")
    nbytes = _lib.lib().libra_gemm_splitk_workspace_bytes(M, N, S)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
    rc = _lib.lib().libra_gemm_bf16_nt_splitk(at.data_ptr(), at.stride(0), bt.data_ptr(), bt.stride(0), ref.data_ptr(), ref.stride(0),
                                              M, N, Kd, S, K.GEMM_A_T | K.GEMM_B_T, ws.data_ptr(), nbytes, K._stream())
    assert rc == 0
    assert torch.equal(o_split, ref)
    close(o_split, a.float() @ b.float().t(), what="K-sliced problem vs fp32")
    close(o_res, a.float() @ b.float().t() + r.float(), what="K-sliced problem with a residual vs fp32")
    assert torch.equal(o_plain, K.gemm_nt(x, w))
    assert _ws_zero(K)
    with pytest.raises(ValueError):
        K.gemm_spec(a, b, bias=rnd(N, seed=1), splitk=2)               # only a residual may be fused into a K-sliced problem


def test_multi_producer_consumer_inside_one_launch(K):
    """`reads=`: the second stage of a low-rank pair y = (x A^T) B^T waits, on the device, for the first stage of the SAME launch
    (LibraLinear, modeling_libra.py:192-199), next to an independent dense GEMM - forward shapes (vision o_proj beside the text
    o_proj), then the dgrad chain with a weight gradient that reads the produced tensor as its reduction-major operand.
    Bit-equal to three separate launches; run several times (the wait is a race if it is wrong)."""
    nv, nl, H, r = 4624, 3000, 1024, 256
    n = nv + nl
    g = torch.Generator().manual_seed(9)
    perm = torch.randperm(n, generator=g)
    lang, vis = perm[:nl].sort().values.to(torch.int32).cuda(), perm[nl:].sort().values.to(torch.int32).cuda()
    o, x = rnd(n, H, seed=1, scale=0.5), rnd(n, H, seed=2)
    wo, wa, wb = rnd(H, H, seed=3, scale=0.03), rnd(r, H, seed=4, scale=0.03), rnd(H, r, seed=5, scale=0.06)
    ref = torch.zeros((n, H), dtype=BF, device="cuda")
    t_ref = K.gemm_nt(o, wa, a_rows=vis, tile=K.GEMM_TILE_256)
    K.gemm_nt(o, wo, out=ref, a_rows=lang, c_rows=lang, resid=x, tile=K.GEMM_TILE_256)
    K.gemm_nt(t_ref, wb, out=ref, c_rows=vis, resid=x, tile=K.GEMM_TILE_256)
    for _ in range(5):
        out = torch.zeros((n, H), dtype=BF, device="cuda")
        t = torch.zeros((nv, r), dtype=BF, device="cuda")
        first = K.gemm_spec(o, wa, out=t, a_rows=vis)
        K.gemm_multi([K.gemm_spec(t, wb, out=out, c_rows=vis, resid=x, reads=first),         # listed first on purpose: the launcher reorders
                      K.gemm_spec(o, wo, out=out, a_rows=lang, c_rows=lang, resid=x), first])
        assert torch.equal(t, t_ref) and torch.equal(out, ref)
    # backward-shaped: dt = dy W_B (bT), then dx = dt W_A (bT) and dW_A = dt^T h both read dt
    dy = rnd(nv, H, seed=6, scale=0.5)
    hv = K.alloc_rows(nv, H, "cuda")[:nv]; hv.copy_(rnd(nv, H, seed=7, scale=0.5))
    full = lambda tt: torch.as_strided(tt, (K.round_up(tt.shape[0], 64), tt.shape[1]), tt.stride(), tt.storage_offset())
    dt_ref = K.gemm_nt(dy, wb, b_t=True, tile=K.GEMM_TILE_256, out=K.alloc_rows(nv, r, "cuda")[:nv])
    dx_ref = K.gemm_nt(dt_ref, wa, b_t=True, tile=K.GEMM_TILE_256)
    dw_ref = K.gemm_nt(full(dt_ref), full(hv), a_t=True, b_t=True, tile=K.GEMM_TILE_256)
    for _ in range(5):
        dt = K.alloc_rows(nv, r, "cuda")[:nv]
        p0 = K.gemm_spec(dy, wb, b_t=True, out=dt)
        dx, dw, _ = K.gemm_multi([K.gemm_spec(dt, wa, b_t=True, reads=p0), K.gemm_spec(full(dt), full(hv), a_t=True, b_t=True, reads=p0), p0])
        assert torch.equal(dt, dt_ref) and torch.equal(dx, dx_ref) and torch.equal(dw, dw_ref)
    assert _ws_zero(K), "queue workspace not clear (word 10 = a device-side wait ran out)"
    with pytest.raises(ValueError):                                    # the producer must be part of the same launch
        K.gemm_multi([K.gemm_spec(t, wb, reads=K.gemm_spec(o, wa, a_rows=vis))])
    with pytest.raises(ValueError):                                    # ... and must not wait for anything itself (one level)
        a1 = K.gemm_spec(o, wa, out=t, a_rows=vis)
        a2 = K.gemm_spec(t, wb, reads=a1)
        K.gemm_multi([a1, a2, K.gemm_spec(a2.out, wa, reads=a2)])
    assert _ws_zero(K)


def test_multi_on_device_self_check_passes(K):
    """The one-time acceptance check the engines run before they switch to multi-problem launches (kernels.gemm_multi_ok)."""
    assert K._multi_selfcheck_inline("cuda") is True and _ws_zero(K)          # the check itself, in this process
    K._MULTI_CHECKED.clear()
    path = K._multi_verdict_file(torch.cuda.current_device())
    if path is not None and os.path.exists(path):
        os.remove(path)
    assert K.gemm_multi_ok("cuda") is True                                     # through the child process + the verdict file
    assert K._MULTI_CHECKED == {torch.cuda.current_device(): True}
    assert path is None or open(path).read() == "ok"


def test_multi_rejects_bad_problems(K):
    a, b = rnd(256, 64, seed=1), rnd(256, 64, seed=2)
    with pytest.raises(ValueError):
        K.gemm_spec(a, rnd(256, 128, seed=3))                      # inner dims differ
    sp = K.gemm_spec(a, b)
    sp.c.K = 96                                                    # not a multiple of 64: the C ABI refuses the whole launch
    with pytest.raises(ValueError):
        K.gemm_multi([K.gemm_spec(a, b), sp])
    assert _ws_zero(K)
