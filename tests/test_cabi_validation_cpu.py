"""Host-side behaviour of the C ABI that needs no GPU: every entry point validates its arguments BEFORE anything
is enqueued (bad shapes -> LIBRA_ERR_SHAPE, null / misaligned pointers -> LIBRA_ERR_ALIGN, empty problems -> LIBRA_OK),
and the pure planning / workspace functions answer for the BASELINE shapes.  The pointers handed in are never
dereferenced on the host; no kernel is launched by any call below."""
import ctypes as C

import pytest

OK, ERR_SHAPE, ERR_ALIGN = 0, -1, -2
P = C.c_void_p
FAKE = P(0x10000)            # 16-byte aligned, non-null, never dereferenced


@pytest.fixture(scope="module")
def L():
    from libra_amd import _lib
    return _lib.lib()


def _gemm(L, *, A=FAKE, lda=128, B=FAKE, ldb=128, Cp=FAKE, ldc=128, M=128, N=128, K=128, flags=0, bias=None):
    return L.libra_gemm_bf16_nt(A, lda, B, ldb, Cp, ldc, M, N, K, bias, None, 0, None, 0, None, 0, 1.0, 0, flags, None)


def test_gemm_argument_validation(L):
    assert _gemm(L, K=100) == ERR_SHAPE                   # reduction length must be a multiple of 64
    assert _gemm(L, lda=64) == ERR_SHAPE                  # leading dimension shorter than the row
    assert _gemm(L, A=None) != OK                         # null operand
    assert _gemm(L, A=P(0x10002)) == ERR_ALIGN            # 16-byte alignment of the LDS-DMA source
    assert _gemm(L, flags=1) != OK                        # LIBRA_GEMM_BIAS without a bias pointer
    assert _gemm(L, M=0) == OK                            # empty problem: nothing to do, nothing launched


def test_row_kernels_argument_validation(L):
    assert L.libra_layernorm_fwd(FAKE, FAKE, FAKE, FAKE, None, None, 4, 1001, 1e-5, None) == ERR_SHAPE      # D % 8
    assert L.libra_layernorm_fwd(FAKE, FAKE, FAKE, FAKE, None, None, 0, 1024, 1e-5, None) == OK
    assert L.libra_vit_attn_fwd(FAKE, 3072, FAKE, 1024, None, None, 1, 577, 0, 0.125, None) == ERR_SHAPE          # no heads
    # rope_bridge_bwd: the 512-thread token group covers H <= 32 heads (Libra-7B / 11B); more is refused, not mis-computed
    args = [FAKE] * 5 + [8192, FAKE, FAKE, 4096, FAKE, 3 * 8192, FAKE, 8192] + [None] * 6 + [0, 10, 10]
    assert L.libra_rope_bridge_bwd(*args, 64, None, 1, None) == ERR_SHAPE
    # with the bridge-gradient output requested, its operands are mandatory
    args2 = [FAKE] * 5 + [4096, FAKE, FAKE, 4096, FAKE, 3 * 4096, FAKE, 4096, None, None, None, None, None, FAKE, 64, 10, 10]
    assert L.libra_rope_bridge_bwd(*args2, 32, None, 1, None) == ERR_ALIGN


def test_rmsnorm_wgrad_row_selection_validation(L):
    # dy, lddy, x, ldx, rstd, flag, dw_lang, dw_vis, workspace, workspace_bytes, rows, D, rows_sel, n_sel, stream
    ws = L.libra_rmsnorm_wgrad_workspace_bytes(100, 4096)
    base = [FAKE, 4096, FAKE, 4096, FAKE, FAKE, FAKE, FAKE, FAKE, ws, 100, 4096]
    assert L.libra_rmsnorm_routed_wgrad(*base, FAKE, 101, None) == ERR_SHAPE       # more selected rows than rows
    assert L.libra_rmsnorm_routed_wgrad(*base, FAKE, -1, None) == ERR_SHAPE
    assert L.libra_rmsnorm_routed_wgrad(*base, FAKE, 0, None) == OK                 # an empty selection adds nothing, launches nothing
    small = list(base); small[9] = ws - 1
    assert L.libra_rmsnorm_routed_wgrad(*small, None, 0, None) == ERR_ALIGN         # workspace too small
    odd = list(base); odd[11] = 4100
    assert L.libra_rmsnorm_routed_wgrad(*odd, None, 0, None) == ERR_SHAPE           # D % 8


def test_generation_entry_points_argument_validation(L):
    # rope with explicit positions: the position operand is mandatory
    args = [FAKE, 3 * 256, FAKE, 64, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, 64, FAKE, FAKE, 256, 4]
    assert L.libra_rope_bridge_pos(*args, None, 1, 2, None) == ERR_ALIGN
    # decode attention: cache row stride shorter than H*128, batch stride shorter than a row, null cache
    ok = dict(q=FAKE, ldq=256, ks=FAKE, kc=FAKE, vs=FAKE, vc=FAKE, ldc=256, bs=256 * 64, kf=FAKE, fs=64, qf=FAKE, kl=FAKE, out=FAKE,
              ldo=256, B=2, H=2)

    def call(**kw):
        a = dict(ok, **kw)
        return L.libra_bridge_attn_decode(a["q"], a["ldq"], a["ks"], a["kc"], a["vs"], a["vc"], a["ldc"], a["bs"], a["kf"], a["fs"],
                                          a["qf"], a["kl"], a.get("kstart"), a["out"], a["ldo"], a["B"], a["H"], 128 ** -0.5,
                                          a.get("ws"), a.get("wsb", 0), None)
    assert call(ldc=128) == ERR_SHAPE
    assert call(ws=FAKE, wsb=16) == ERR_ALIGN                          # a workspace smaller than the key-split partial states
    assert L.libra_bridge_attn_decode_workspace_bytes(2, 2) == 2 * 2 * 4 * 132 * 4
    assert call(bs=64) == ERR_SHAPE
    assert call(kc=None) == ERR_ALIGN
    assert call(ks=P(0x10008)) == ERR_ALIGN
    assert call(B=0) == OK
    # fused gate | up + SwiGLU of a generation step: at most 16 rows, K a multiple of 64, K-contiguous operands
    base = dict(A=FAKE, lda=4096, W=FAKE, ldw=4096, Y=FAKE, ldy=11008, M=8, I=11008, K=4096, ar=None, ap=8, st=None)
    sw = lambda **kw: L.libra_gemm_swiglu_skinny(*[dict(base, **kw)[k] for k in base])
    assert sw(M=17) == ERR_SHAPE and sw(K=4000) == ERR_SHAPE and sw(lda=2048) == ERR_SHAPE and sw(ldy=100) == ERR_SHAPE
    assert sw(W=None) == ERR_SHAPE and sw(A=P(0x10008)) == ERR_ALIGN and sw(M=0) == OK and sw(ar=FAKE, ap=0) == ERR_SHAPE


def test_splitk_plan_for_the_baseline_shapes(L):
    # weight gradients of the ViT step (reduction over 32 x 577 tokens, padded to 18496): few tiles, long K -> sliced
    for M, N in [(1024, 4096), (4096, 1024), (3072, 1024), (1024, 1024)]:
        s = L.libra_gemm_splitk_plan(M, N, 18496)
        assert 2 <= s <= 64, (M, N, s)
        assert L.libra_gemm_splitk_workspace_bytes(M, N, s) == s * M * N * 4
    # forward / dgrad GEMMs of the same step fill the chip on their own
    for M, N, K in [(18464, 4096, 1024), (18464, 1024, 4096), (18464, 3072, 1024), (16384, 11008, 4096)]:
        assert L.libra_gemm_splitk_plan(M, N, K) == 1, (M, N, K)
    # skinny weight gradients of the rank-8 bridges (M = 8 or 16 output rows, thousands of tokens)
    assert L.libra_gemm_splitk_plan(16, 4096, 16384) > 1
    assert L.libra_gemm_splitk_plan(8, 8, 64) == 1        # too small to be worth slicing


def test_workspace_sizes_are_positive_and_scale(L):
    a = L.libra_layernorm_bwd_workspace_bytes(18464, 1024)
    b = L.libra_layernorm_bwd_workspace_bytes(18464, 2048)
    assert 0 < a < b
    assert L.libra_colsum_workspace_bytes(18464, 4096) > 0
    assert L.libra_rmsnorm_wgrad_workspace_bytes(16384, 4096) > 0
    assert L.libra_rmsnorm_wgrad_workspace_bytes(0, 4096) == 0


def test_error_codes_map_to_the_reference_exception_types():
    from libra_amd import _lib
    _lib.check(OK, "x")
    with pytest.raises(ValueError):            # shape mismatches are ValueError upstream (modeling_libra.py:374-403)
        _lib.check(ERR_SHAPE, "x")
    with pytest.raises(ValueError):
        _lib.check(ERR_ALIGN, "x")
    with pytest.raises(_lib.LibraHipError):
        _lib.check(-3, "x")


def test_gemm_multi_argument_validation(L):
    """libra_gemm_bf16_multi validates EVERY problem before anything is enqueued (and reads the ctypes records at the offsets
    the header's struct defines: a field that landed in the wrong place would not produce these verdicts)."""
    from libra_amd import _lib
    fake = 0x10000

    def prob(**kw):
        d = dict(A=fake, lda=128, B=fake, ldb=128, C=fake, ldc=128, M=256, N=128, K=128, bias=None, resid=None, ldr=0, aux=None,
                 ldaux=0, preact=None, ldpre=0, alpha=1.0, flags=0, alpha_cols=0, a_rows=None, a_phys_rows=256, c_rows=None,
                 splitk=1, slab=None, wait_on=-1)
        d.update(kw)
        return _lib.GemmProblem(**d)

    def call(ps, ws=FAKE):
        arr = (_lib.GemmProblem * len(ps))(*ps)
        return L.libra_gemm_bf16_multi(arr, len(ps), ws, None)
    assert call([prob(M=0), prob(N=0)]) == OK                              # only empty problems: nothing to do, nothing launched
    assert call([]) == OK
    assert call([prob()], ws=None) == ERR_ALIGN                            # the queue workspace is mandatory
    assert call([prob()] * (_lib.GEMM_MULTI_MAX + 1)) == ERR_SHAPE
    assert call([prob(), prob(K=100)]) == ERR_SHAPE                        # one bad problem refuses the whole launch
    assert call([prob(lda=64)]) == ERR_SHAPE
    assert call([prob(A=0x10002)]) == ERR_ALIGN
    assert call([prob(flags=4)]) == ERR_ALIGN                              # LIBRA_GEMM_RESIDUAL without a residual
    assert call([prob(flags=32, M=100, lda=128)]) == ERR_SHAPE             # A_T: M % 8
    assert call([prob(a_rows=fake, flags=32)]) == ERR_SHAPE                # a row gather needs a K-contiguous A
    assert call([prob(splitk=2)]) == ERR_ALIGN                             # K slices without a slab
    assert call([prob(splitk=4, slab=fake)]) == ERR_SHAPE                  # more slices than K tiles (K = 128 -> 2)
    assert call([prob(splitk=2, slab=fake, flags=1, bias=fake)]) == ERR_SHAPE      # only a residual may ride on a K-sliced problem
    assert call([prob(wait_on=1)]) == ERR_SHAPE                            # the producer must be a problem of this call ...
    assert call([prob(wait_on=0)]) == ERR_SHAPE                            # ... other than itself ...
    assert call([prob(), prob(wait_on=0), prob(wait_on=1)]) == ERR_SHAPE   # ... that waits for nothing itself ...
    assert call([prob(splitk=2, slab=fake), prob(wait_on=0)]) == ERR_SHAPE # ... and is not K-sliced (its C is complete only after the reduction)
