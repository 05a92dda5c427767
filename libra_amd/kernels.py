"""Thin, typed host wrappers over the C-ABI kernels (include/libra_hip.h).

PyTorch is used only for device memory (the caching allocator), streams and dtype bookkeeping;
every arithmetic step below is a HIP kernel launch on the *current* stream (re-read on every call:
autograd runs backward on its own thread).
"""
from __future__ import annotations

from typing import Optional, Sequence

import ctypes as C
import warnings
import weakref

import torch

from . import _lib

GEMM_BIAS, GEMM_QUICK_GELU, GEMM_RESIDUAL, GEMM_MUL_QGELU_GRAD, GEMM_STORE_PREACT, GEMM_A_T, GEMM_B_T = 1, 2, 4, 8, 16, 32, 64
GEMM_TILE_AUTO, GEMM_TILE_128, GEMM_TILE_256, GEMM_TILE_W = 0, 1, 2, 3      # include/libra_hip.h LIBRA_GEMM_TILE_*
BF16 = torch.bfloat16


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk2d(t: torch.Tensor, name: str, dtype=BF16):
    if t.dim() != 2 or t.stride(1) != 1 or t.dtype != dtype or not t.is_cuda:
        raise ValueError(f"{name}: expected a 2-D row-major cuda {dtype} tensor, got {tuple(t.shape)} "
                         f"strides {t.stride()} {t.dtype} {t.device}")


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def alloc_rows(rows: int, cols: int, device) -> torch.Tensor:
    """[round_up(rows,64), cols] bf16 with the pad rows zeroed: activations / gradients that later serve as
    reduction-major (token-major) GEMM operands for wgrad need a reduction length that is a multiple of 64.
    Use ``buf[:rows]`` as the kernel output."""
    rp = round_up(rows, 64)
    buf = torch.empty((rp, cols), dtype=BF16, device=device)
    if rp != rows:
        buf[rows:].zero_()
    return buf


class _ArenaLease:
    """Ownership token of a RowArena's per-layer tags: it lives in the `saved` state of ONE forward."""
    __slots__ = ("__weakref__",)


class RowArena:
    """Reusable zero-padded row buffers for one model: `rows(tag, n, c)` behaves like `alloc_rows(n, c)[:n]` but hands out the
    SAME storage for the same tag every time, so the pad rows are zeroed once instead of once per call (the decoder step made
    ~800 five-microsecond fill launches for them, and the caching allocator round trips on top).  Only for buffers whose lifetime
    is bounded by the caller's own schedule: per-layer temporaries of a backward (one tag serves all layers, the stream orders
    the reuse) and per-layer saved activations (one tag per layer, reused by the next step).

    The per-layer tags belong to at most one saved forward at a time.  Ownership is a LEASE held by that forward's saved state and
    tracked through a weak reference: it ends when the backward releases it OR when the saved state is dropped without a backward
    (an evaluation forward outside `no_grad`, `float(model(**kw).loss)`, an exception between forward and backward) - a bool
    flag cleared only by backward() stayed set for good in those cases and silently sent every later step to fresh allocations."""

    def __init__(self):
        self.bufs = {}
        self._lease = None         # weakref to the _ArenaLease of the saved forward that owns the per-layer tags, or None
        self._warned = False

    @property
    def busy(self) -> bool:
        return self._lease is not None and self._lease() is not None

    def lease(self) -> "_ArenaLease":
        tok = _ArenaLease()
        self._lease = weakref.ref(tok)
        return tok

    def release(self, tok) -> None:
        if tok is not None and self._lease is not None and self._lease() is tok:
            self._lease = None

    def warn_busy(self) -> None:
        """A training forward found the arena leased to an earlier forward that is still alive (two graphs in flight): it falls
        back to fresh buffers - correct, but the step then holds a second set of saved activations."""
        if not self._warned:
            self._warned = True
            warnings.warn("libra_amd: a second grad-enabled forward started while the previous one still holds its saved "
                          "activations; it uses freshly allocated buffers (activation memory roughly doubles). Run evaluation "
                          "forwards under torch.no_grad() or drop the earlier output first.", RuntimeWarning, stacklevel=3)

    def rows(self, tag: str, rows: int, cols: int, device) -> torch.Tensor:
        rp = round_up(rows, 64)
        key = (tag, cols)
        ent = self.bufs.get(key)
        if ent is None or ent[0].shape[0] < rp or ent[0].device != torch.device(device):
            buf = torch.zeros((rp, cols), dtype=BF16, device=device)
            self.bufs[key] = [buf, rows]
            return buf[:rows]
        buf, hi = ent
        if hi > rows:
            buf[rows:hi].zero_()                         # a shorter use after a longer one: its tail is pad again
        ent[1] = rows
        return buf[:rows]

    def nbytes(self) -> int:
        return sum(b.numel() * b.element_size() for b, _ in self.bufs.values())


class DeviceErrors:
    """Sticky device-side error word of kernels with bounded in-kernel waits (`err_word` of libra_bridge_attn_bwd).  The library
    never synchronises, so the word is polled WITHOUT a stall: `poll_async` queues a 4-byte copy into pinned memory behind the
    work just launched, `check` (called at the start of the next forward, or with wait=True where the caller synchronises anyway)
    raises LibraHipError once the copy has completed and shows a non-zero word."""

    def __init__(self):
        self._st = {}

    def _state(self, device):
        dev = torch.device(device)
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        st = self._st.get(key)
        if st is None:
            d = torch.device("cuda", key)
            st = dict(word=torch.zeros(1, dtype=torch.int32, device=d), host=torch.zeros(1, dtype=torch.int32).pin_memory(),
                      event=torch.cuda.Event(), pending=False)
            self._st[key] = st
        return st

    def word(self, device) -> torch.Tensor:
        return self._state(device)["word"]

    def poll_async(self, device) -> None:
        st = self._state(device)
        st["host"].copy_(st["word"], non_blocking=True)
        st["event"].record()
        st["pending"] = True

    def check(self, device, wait: bool = False, pending_only: bool = False) -> None:
        """wait=False: raise only if an earlier poll has already landed.  wait=True: poll now and wait for it (a host
        synchronisation).  pending_only: wait only when a backward has queued a poll since the last check - what an optimizer
        calls before it applies that backward's gradients (dp.FlatAdamW.step)."""
        st = self._state(device)
        if wait and pending_only and not st["pending"]:
            return
        if wait:
            if not (pending_only and st["pending"]):
                self.poll_async(device)
            st["event"].synchronize()
        elif not st["pending"] or not st["event"].query():
            return
        st["pending"] = False
        bits = int(st["host"][0])
        if bits:
            st["word"].zero_()
            st["host"].zero_()
            raise _lib.LibraHipError(f"device-side error word = {bits:#x}: a bounded in-kernel wait of the bridge-attention backward "
                                     "(dK/dV P hand-over) ran out; the gradients of that step are invalid")


errors = DeviceErrors()


def gemm_nt(a: torch.Tensor, b: torch.Tensor, *, out: Optional[torch.Tensor] = None,
            bias: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, quick_gelu: bool = False,
            qgelu_grad_of: Optional[torch.Tensor] = None, preact_out: Optional[torch.Tensor] = None,
            alpha: float = 1.0, alpha_cols: int = 0, k: Optional[int] = None, a_t: bool = False,
            b_t: bool = False, a_rows: Optional[torch.Tensor] = None, c_rows: Optional[torch.Tensor] = None,
            tile: int = 0) -> torch.Tensor:
    """out[M,N] = epi(A @ B^T) with A = a [M,K] (or a^T when a_t: a is [K,M], reduction-major) and
    B = b [N,K] (or b^T when b_t: b is [K,N]).  `k` limits the contraction to the first k reduction steps.
    The reduction length must be a multiple of 64 (reduction-major operands: allocate with alloc_rows).
    `tile` (GEMM_TILE_128 / _256 / _W; 0 = the library's cost model) pins the tile structure and disables the split-K path:
    for the parity tests (every structure on every shape) and for tools/gemm_sweep.py."""
    _chk2d(a, "a"); _chk2d(b, "b")
    (Ka, M) = a.shape if a_t else a.shape[::-1]
    a_phys = a.shape[0]
    if a_rows is not None:                       # routed gather: logical rows = len(a_rows)
        if a_t or a_rows.dtype != torch.int32:
            raise ValueError("gemm_nt: a_rows needs a K-contiguous A and an int32 index tensor")
        M = a_rows.numel()
    (Kb, N) = b.shape if b_t else b.shape[::-1]
    K = k if k is not None else Ka
    if K > Ka or K > Kb or (k is None and Ka != Kb):
        raise ValueError(f"gemm_nt: inner dims differ: a {tuple(a.shape)} b {tuple(b.shape)} k={k} a_t={a_t} b_t={b_t}")
    if c_rows is not None:
        if out is None or c_rows.dtype != torch.int32 or c_rows.numel() != M or out.shape[1] != N:
            raise ValueError("gemm_nt: c_rows (int32 [M]) scatters into a caller-provided out [rows, N]")
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=a.device)
    _chk2d(out, "out")
    if c_rows is None and out.shape != (M, N):
        raise ValueError(f"gemm_nt: out is {tuple(out.shape)}, expected {(M, N)}")
    plain = (bias is None and resid is None and not quick_gelu and qgelu_grad_of is None and preact_out is None
             and alpha_cols == 0 and a_rows is None and c_rows is None)
    # K-sliced launch + deterministic fp32 slab reduction where the unsplit problem would fill a fraction of the chip: plain
    # problems (weight gradients), and routed ones with at most a residual (few-row text / vision projections)
    # (M <= 16 routed rows are the generation step: the library's weight-streaming skinny kernel, not K slices)
    routed_ok = (not plain and M > 16 and bias is None and not quick_gelu and qgelu_grad_of is None and preact_out is None and alpha_cols == 0
                 and not a_t and (resid is None or (resid.dim() == 2 and resid.stride(1) == 1 and resid.dtype == BF16
                                                    and resid.is_cuda and resid.shape == out.shape)))
    if (plain or routed_ok) and tile == 0 and M > 0 and N > 0 and not (a_t and M % 8):
        splits = _lib.lib().libra_gemm_splitk_plan(M, N, K)
        if splits > 1:
            nbytes = _lib.lib().libra_gemm_splitk_workspace_bytes(M, N, splits)
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=a.device)
            if plain:
                rc = _lib.lib().libra_gemm_bf16_nt_splitk(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(),
                                                          out.stride(0), M, N, K, splits,
                                                          (GEMM_A_T if a_t else 0) | (GEMM_B_T if b_t else 0), ws.data_ptr(),
                                                          nbytes, _stream())
            else:
                rc = _lib.lib().libra_gemm_bf16_nt_splitk_routed(
                    a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), M, N, K, splits,
                    (GEMM_B_T if b_t else 0) | (GEMM_RESIDUAL if resid is not None else 0), _ptr(resid),
                    resid.stride(0) if resid is not None else 0, _ptr(a_rows), a_phys, _ptr(c_rows), ws.data_ptr(), nbytes, _stream())
            _lib.check(rc, f"gemm_nt_splitk M={M} N={N} K={K} S={splits}")
            return out
    flags = (GEMM_A_T if a_t else 0) | (GEMM_B_T if b_t else 0)
    if bias is not None:
        if bias.numel() != N or bias.dtype != BF16:
            raise ValueError("gemm_nt: bias must be bf16 [N]")
        flags |= GEMM_BIAS
    ldr = ldaux = ldpre = 0
    if resid is not None:
        _chk2d(resid, "resid")
        if resid.shape != out.shape:
            raise ValueError("gemm_nt: resid shape")
        flags |= GEMM_RESIDUAL; ldr = resid.stride(0)
    if qgelu_grad_of is not None:
        _chk2d(qgelu_grad_of, "qgelu_grad_of")
        if qgelu_grad_of.shape != out.shape:
            raise ValueError("gemm_nt: qgelu_grad_of shape")
        flags |= GEMM_MUL_QGELU_GRAD; ldaux = qgelu_grad_of.stride(0)
    if preact_out is not None:
        _chk2d(preact_out, "preact_out")
        if preact_out.shape != out.shape:
            raise ValueError("gemm_nt: preact_out shape")
        flags |= GEMM_STORE_PREACT; ldpre = preact_out.stride(0)
    if quick_gelu:
        flags |= GEMM_QUICK_GELU
    rc = _lib.lib().libra_gemm_bf16_nt_tile(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(),
                                            out.stride(0), M, N, K, _ptr(bias), _ptr(resid), ldr,
                                            _ptr(qgelu_grad_of), ldaux, _ptr(preact_out), ldpre, float(alpha),
                                            int(alpha_cols), flags, _ptr(a_rows), a_phys, _ptr(c_rows), int(tile), _stream())
    _lib.check(rc, f"gemm_nt M={M} N={N} K={K}")
    return out


def gemm_swiglu_skinny(a: torch.Tensor, w_gate_up: torch.Tensor, *, a_rows: Optional[torch.Tensor] = None,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """silu(a @ W_gate^T) * (a @ W_up^T) for M <= 16 rows in one launch (libra_gemm_swiglu_skinny): w_gate_up [2I, K] = gate rows
    then up rows.  Same bits as gemm_nt + swiglu."""
    _chk2d(a, "a"); _chk2d(w_gate_up, "w_gate_up")
    M = a_rows.numel() if a_rows is not None else a.shape[0]
    I, Kd = w_gate_up.shape[0] // 2, w_gate_up.shape[1]
    if a.shape[1] != Kd or w_gate_up.shape[0] != 2 * I or M > 16:
        raise ValueError(f"gemm_swiglu_skinny: a {tuple(a.shape)} w {tuple(w_gate_up.shape)} rows {M} (<= 16)")
    if a_rows is not None and a_rows.dtype != torch.int32:
        raise ValueError("gemm_swiglu_skinny: a_rows must be int32")
    if out is None:
        out = torch.empty((M, I), dtype=BF16, device=a.device)
    _chk2d(out, "out")
    if out.shape != (M, I):
        raise ValueError("gemm_swiglu_skinny: out shape")
    rc = _lib.lib().libra_gemm_swiglu_skinny(a.data_ptr(), a.stride(0), w_gate_up.data_ptr(), w_gate_up.stride(0), out.data_ptr(),
                                             out.stride(0), M, I, Kd, _ptr(a_rows), a.shape[0], _stream())
    _lib.check(rc, f"gemm_swiglu_skinny M={M} I={I} K={Kd}")
    return out


def gemm_nt_grouped(a_list: Sequence[torch.Tensor], b_list: Sequence[torch.Tensor], outs: Sequence[torch.Tensor], *,
                    a_t: bool = False, b_t: bool = False, a_rows: Optional[torch.Tensor] = None,
                    c_rows: Optional[torch.Tensor] = None, tile: int = 0) -> Sequence[torch.Tensor]:
    """outs[g] = A_g @ B_g^T for up to 4 problems of identical shape / strides / row maps in ONE launch (the groups
    share whole waves of workgroups).  Same operand conventions as gemm_nt; no fused epilogue operands."""
    G = len(a_list)
    if not (1 <= G <= 4) or len(b_list) != G or len(outs) != G:
        raise ValueError("gemm_nt_grouped: 1..4 groups, one a / b / out each")
    for t in list(a_list) + list(b_list) + list(outs):
        _chk2d(t, "gemm_nt_grouped operand")
    a0, b0, o0 = a_list[0], b_list[0], outs[0]
    for a, b, o in zip(a_list, b_list, outs):
        if a.shape != a0.shape or b.shape != b0.shape or o.shape != o0.shape or a.stride(0) != a0.stride(0) \
                or b.stride(0) != b0.stride(0) or o.stride(0) != o0.stride(0):
            raise ValueError("gemm_nt_grouped: all groups must share shapes and row strides")
    (Ka, M) = a0.shape if a_t else a0.shape[::-1]
    a_phys = a0.shape[0]
    if a_rows is not None:
        if a_t:
            raise ValueError("gemm_nt_grouped: a_rows needs a K-contiguous A")
        M = a_rows.numel()
    (Kb, N) = b0.shape if b_t else b0.shape[::-1]
    if Ka != Kb or Ka % 64:
        raise ValueError(f"gemm_nt_grouped: inner dims {Ka} vs {Kb} (must match, multiple of 64)")
    if c_rows is None and o0.shape != (M, N):
        raise ValueError("gemm_nt_grouped: out shape")
    if c_rows is not None and (c_rows.numel() != M or o0.shape[1] != N):
        raise ValueError("gemm_nt_grouped: c_rows / out shape")
    arr = lambda ts: (C.c_void_p * G)(*[t.data_ptr() for t in ts])
    flags = (GEMM_A_T if a_t else 0) | (GEMM_B_T if b_t else 0)
    prof = LaunchProfile.active
    if prof is not None:
        ev = prof.bracket("gemm", (2.0 * M * N * Ka * G, 2.0 * G * (M * Ka + N * Ka + M * N),
                                   f"{G}x[{M}x{N}x{Ka}{' aT' if a_t else ''}{' bT' if b_t else ''}]"))
        ev[0].record()
    rc = _lib.lib().libra_gemm_bf16_nt_grouped(arr(a_list), a0.stride(0), arr(b_list), b0.stride(0), arr(outs), o0.stride(0),
                                               G, M, N, Ka, 1.0, 0, flags, _ptr(a_rows), a_phys, _ptr(c_rows), int(tile), _stream())
    if prof is not None:
        ev[1].record()
    _lib.check(rc, f"gemm_nt_grouped G={G} M={M} N={N} K={Ka}")
    return outs


class GemmSpec:
    """One problem of a multi-problem launch (gemm_multi): the arguments of a gemm_nt call, validated, as the C ABI's
    `libra_gemm_problem` - plus the tensors it points into (kept alive until the launch is queued) and its work for the profile."""
    __slots__ = ("c", "out", "keep", "work", "reads")


def gemm_spec(a: torch.Tensor, b: torch.Tensor, *, out: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
              resid: Optional[torch.Tensor] = None, quick_gelu: bool = False, qgelu_grad_of: Optional[torch.Tensor] = None,
              preact_out: Optional[torch.Tensor] = None, alpha: float = 1.0, alpha_cols: int = 0, k: Optional[int] = None,
              a_t: bool = False, b_t: bool = False, a_rows: Optional[torch.Tensor] = None,
              c_rows: Optional[torch.Tensor] = None, splitk: int = 1, reads: Optional["GemmSpec"] = None) -> GemmSpec:
    """The operands of `gemm_nt(a, b, ...)` (same conventions, same checks) as one problem of a `gemm_multi` launch.
    splitk > 1: cut the reduction into that many slices, each a tile-list entry of its own, fp32 partial slabs reduced in slice
    order by a second kernel (weight gradients: few tiles, very long K); at most a residual may be fused, N % 8 == 0.
    reads: another problem OF THE SAME gemm_multi CALL whose output this problem reads (as a, b or resid) - its tiles then wait on
    the device until the producer's are stored (libra_gemm_problem.wait_on); the producer must be unsplit and read nothing itself."""
    _chk2d(a, "a"); _chk2d(b, "b")
    (Ka, M) = a.shape if a_t else a.shape[::-1]
    a_phys = a.shape[0]
    if a_rows is not None:
        if a_t or a_rows.dtype != torch.int32:
            raise ValueError("gemm_spec: a_rows needs a K-contiguous A and an int32 index tensor")
        M = a_rows.numel()
    (Kb, N) = b.shape if b_t else b.shape[::-1]
    K = k if k is not None else Ka
    if K > Ka or K > Kb or (k is None and Ka != Kb):
        raise ValueError(f"gemm_spec: inner dims differ: a {tuple(a.shape)} b {tuple(b.shape)} k={k} a_t={a_t} b_t={b_t}")
    if c_rows is not None:
        if out is None or c_rows.dtype != torch.int32 or c_rows.numel() != M or out.shape[1] != N:
            raise ValueError("gemm_spec: c_rows (int32 [M]) scatters into a caller-provided out [rows, N]")
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=a.device)
    _chk2d(out, "out")
    if c_rows is None and out.shape != (M, N):
        raise ValueError(f"gemm_spec: out is {tuple(out.shape)}, expected {(M, N)}")
    flags = (GEMM_A_T if a_t else 0) | (GEMM_B_T if b_t else 0)
    if bias is not None:
        if bias.numel() != N or bias.dtype != BF16:
            raise ValueError("gemm_spec: bias must be bf16 [N]")
        flags |= GEMM_BIAS
    ldr = ldaux = ldpre = 0
    for t, nm in ((resid, "resid"), (qgelu_grad_of, "qgelu_grad_of"), (preact_out, "preact_out")):
        if t is not None:
            _chk2d(t, nm)
            if t.shape != out.shape:
                raise ValueError(f"gemm_spec: {nm} shape")
    if resid is not None:
        flags |= GEMM_RESIDUAL; ldr = resid.stride(0)
    if qgelu_grad_of is not None:
        flags |= GEMM_MUL_QGELU_GRAD; ldaux = qgelu_grad_of.stride(0)
    if preact_out is not None:
        flags |= GEMM_STORE_PREACT; ldpre = preact_out.stride(0)
    if quick_gelu:
        flags |= GEMM_QUICK_GELU
    slab = None
    splitk = max(1, min(int(splitk), K // 64))
    if splitk > 1:
        if (flags & ~(GEMM_A_T | GEMM_B_T | GEMM_RESIDUAL)) or alpha_cols or N % 8:
            raise ValueError("gemm_spec: a K-sliced problem may only fuse a residual (and needs N % 8 == 0)")
        slab = torch.empty(splitk * M * N, dtype=torch.float32, device=a.device)
    sp = GemmSpec()
    sp.c = _lib.GemmProblem(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), M, N, K,
                            _ptr(bias), _ptr(resid), ldr, _ptr(qgelu_grad_of), ldaux, _ptr(preact_out), ldpre, float(alpha), flags,
                            int(alpha_cols), _ptr(a_rows), a_phys, _ptr(c_rows), splitk, _ptr(slab), -1)
    sp.reads = reads
    sp.out = out
    sp.keep = (a, b, out, bias, resid, qgelu_grad_of, preact_out, a_rows, c_rows, slab)
    sp.work = (2.0 * M * N * K, 2.0 * (M * K + N * K + M * N), f"{M}x{N}x{K}{' aT' if a_t else ''}{' bT' if b_t else ''}")
    return sp


_MULTI_WS = {}


def _multi_ws(device) -> torch.Tensor:
    """The 128-byte tile-queue workspace of libra_gemm_bf16_multi: one per (device, stream), zeroed ONCE (the kernel leaves it zero)."""
    dev = torch.device(device)
    if torch.cuda.is_current_stream_capturing():
        # inside a hipGraph capture the launch belongs to the GRAPH, which may be replayed on any stream next to other graphs: it gets
        # a workspace of its own from the graph's private pool (a 128-byte fill node per launch; nothing cached outside the graph)
        return torch.zeros(32, dtype=torch.int32, device=dev)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream())
    ws = _MULTI_WS.get(key)
    if ws is None:
        ws = _MULTI_WS[key] = torch.zeros(32, dtype=torch.int32, device=dev)
    return ws


def gemm_multi(specs: Sequence[GemmSpec]) -> list:
    """The problems `specs` (gemm_spec(...), each what one gemm_nt call would compute) as ONE persistent launch over a common tile
    list, longest reduction first (libra_gemm_bf16_multi).  The problems must be independent: none reads what another one writes
    (two may write disjoint rows of the same tensor through their row maps).  More than 12 problems go out as several launches.
    -> [out of each problem]."""
    specs = list(specs)
    if not specs:
        return []
    dev = specs[0].out.device
    n_max = _lib.GEMM_MULTI_MAX
    for i0 in range(0, len(specs), n_max):
        chunk = specs[i0:i0 + n_max]
        for sp in chunk:                                 # producer -> consumer edges, by position inside this launch
            sp.c.wait_on = -1
            if sp.reads is not None:
                where = [j for j, other in enumerate(chunk) if other is sp.reads]
                if not where:
                    raise ValueError("gemm_multi: a problem `reads` a problem that is not part of the same launch")
                sp.c.wait_on = where[0]
        arr = (_lib.GemmProblem * len(chunk))(*[sp.c for sp in chunk])
        prof = LaunchProfile.active
        if prof is not None:
            tag = "multi{" + " + ".join(_tag_counts([sp.work[2] for sp in chunk])) + "}"
            ev = prof.bracket("gemm", (sum(sp.work[0] for sp in chunk), sum(sp.work[1] for sp in chunk), tag))
            ev[0].record()
        rc = _lib.lib().libra_gemm_bf16_multi(arr, len(chunk), _multi_ws(dev).data_ptr(), _stream())
        if prof is not None:
            ev[1].record()
        _lib.check(rc, f"gemm_multi [{', '.join(sp.work[2] for sp in chunk)}]")
    return [sp.out for sp in specs]


_MULTI_CHECKED = {}


def _multi_selfcheck_inline(device) -> bool:
    """The acceptance check itself, in THIS process: four small problems - routed with a residual, reduction-major B, both operands
    reduction-major, K-sliced - through `gemm_multi` and through single launches of the same tile body must agree bit for bit
    (the K-sliced one: within the kernel tolerance of fp32 math) and the queue workspace must come back zero."""
    dev = torch.device(device)
    g = torch.Generator().manual_seed(1234)
    rn = lambda *shape: (torch.randn(*shape, generator=g) * 0.5).to(BF16).to(dev)
    rows = torch.randperm(340, generator=g)[:300].to(torch.int32).to(dev)
    a0, b0, r0 = rn(340, 192), rn(264, 192), rn(340, 264)
    a1, b1 = rn(520, 256), rn(256, 264)
    a2, b2 = rn(128, 256), rn(128, 264)
    a3, b3 = rn(256, 1024), rn(264, 1024)
    cases = [dict(a=a0, b=b0, a_rows=rows, c_rows=rows, resid=r0), dict(a=a1, b=b1, b_t=True), dict(a=a2, b=b2, a_t=True, b_t=True),
             dict(a=a3, b=b3)]
    outs_m = [torch.zeros((340, 264), dtype=BF16, device=dev), None, None, None]
    outs_s = [torch.zeros((340, 264), dtype=BF16, device=dev), None, None, None]
    specs = []
    for i, c in enumerate(cases):
        kw = {k: v for k, v in c.items() if k not in ("a", "b")}
        specs.append(gemm_spec(c["a"], c["b"], out=outs_m[i], splitk=4 if i == 3 else 1, **kw))
    res_m = gemm_multi(specs)
    for i, c in enumerate(cases[:3]):
        kw = {k: v for k, v in c.items() if k not in ("a", "b")}
        outs_s[i] = gemm_nt(c["a"], c["b"], out=outs_s[i], tile=GEMM_TILE_256, **kw)      # (every case is >= 256 x 256: the same tile body)
    ref3 = a3.float() @ b3.float().t()
    ok = all(torch.equal(res_m[i], outs_s[i]) for i in range(3))
    err = (res_m[3].float() - ref3).abs()
    ok = ok and bool((err <= 1e-3 * ref3.abs().max() + 2.0 ** -8 * ref3.abs()).all())
    return ok and int(_multi_ws(dev).abs().sum()) == 0


def _multi_verdict_file(key: int):
    """Where the verdict of the acceptance check is remembered across processes: keyed by this build of the library and the device
    model (as MIOpen's find-db or an autotuner's cache: a property of (kernel binary, chip), not of a process)."""
    import hashlib
    import os
    try:
        st = os.stat(_lib.LIB_PATH)
        tag = f"{st.st_size}-{int(st.st_mtime)}-{_lib.ABI_VERSION}-{torch.cuda.get_device_name(key)}"
        d = os.path.join(os.path.expanduser("~"), ".cache", "libra_amd")
        return os.path.join(d, "gemm_multi_ok_" + hashlib.sha256(tag.encode()).hexdigest()[:16])
    except Exception:
        return None


def gemm_multi_ok(device) -> bool:
    """May the engines use multi-problem launches on this device?  One-time acceptance check of `libra_gemm_bf16_multi` (cached per
    process and, through a one-line file under ~/.cache/libra_amd, per library build and device model).  The engines ask before
    they switch their launch schedule (decoder_engine.MULTI / vit_engine.MULTI): a schedule switch is a pure speed choice and the
    kernel's hand-out of tiles leans on properties of the running system, so a device where the check fails keeps the
    one-launch-per-GEMM schedule, with a warning, instead of training on wrong numbers.  The check (`_multi_selfcheck_inline`) runs
    in a CHILD process the first time: a kernel that faulted would take the caller's HIP context - i.e. the training job - with it;
    a child that crashes is just a failed check.  Inside a graph capture the answer is the cached one (or True: a capture is always
    preceded by eager warm-up steps)."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key in _MULTI_CHECKED:
        return _MULTI_CHECKED[key]
    if torch.cuda.is_current_stream_capturing():
        return True
    import os
    import subprocess
    import sys
    path = _multi_verdict_file(key)
    ok = None
    if path is not None and os.path.exists(path):
        try:
            ok = open(path).read().strip() == "ok"
        except OSError:
            ok = None
    if ok is None:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        code = (f"import sys; sys.path.insert(0, {root!r}); import torch; from libra_amd import kernels as K; "
                f"torch.cuda.set_device({key}); sys.exit(0 if K._multi_selfcheck_inline('cuda:{key}') else 3)")
        remember = True
        try:
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
            ok = r.returncode == 0
            why = f"child exit code {r.returncode}: {r.stderr.strip()[-300:]}"
        except Exception as e:                                       # (timeout, no interpreter, ...): not a verdict about the kernel
            ok, why, remember = False, repr(e), False
        if not ok:
            warnings.warn("libra_amd: the multi-problem GEMM launch FAILED its on-device acceptance check on this device "
                          f"({why}); the engines keep the one-launch-per-GEMM schedule (slower, same results). Please report this.",
                          RuntimeWarning)
        if path is not None and remember:
            try:
                os.makedirs(os.path.dirname(path), exist_ok=True)
                with open(path, "w") as f:
                    f.write("ok" if ok else "failed")
            except OSError:
                pass
    _MULTI_CHECKED[key] = ok
    return ok


def _tag_counts(tags):
    out, seen = [], {}
    for t in tags:
        seen[t] = seen.get(t, 0) + 1
    for t in dict.fromkeys(tags):
        out.append(t if seen[t] == 1 else f"{seen[t]}x[{t}]")
    return out


def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, *, save_stats: bool = True,
                  out: Optional[torch.Tensor] = None):
    _chk2d(x, "x")
    if not x.is_contiguous():
        raise ValueError("layernorm_fwd: x must be contiguous")
    rows, D = x.shape
    y = torch.empty_like(x) if out is None else out
    mean = rstd = None
    if save_stats:
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    rc = _lib.lib().libra_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _ptr(mean),
                                        _ptr(rstd), rows, D, float(eps), _stream())
    _lib.check(rc, f"layernorm_fwd rows={rows} D={D}")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, *, dres=None, dgamma=None, dbeta=None, dxsum=None, out=None):
    """dx = LN'(dy) [+ dres]; dgamma/dbeta (and dxsum += column sum of dx) are fp32 [D] accumulators (added to), or None."""
    _chk2d(dy, "dy"); _chk2d(x, "x")
    rows, D = x.shape
    if dy.shape != x.shape or not (dy.is_contiguous() and x.is_contiguous()):
        raise ValueError("layernorm_bwd: dy/x must be contiguous and equal-shaped")
    dx = torch.empty_like(x) if out is None else out
    ws, ws_bytes = None, 0
    if dgamma is not None:
        ws_bytes = _lib.lib().libra_layernorm_bwd_workspace_bytes(rows, D)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
    rc = _lib.lib().libra_layernorm_bwd(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                        rstd.data_ptr(), _ptr(dres), dx.data_ptr(), _ptr(dgamma), _ptr(dbeta),
                                        _ptr(dxsum), _ptr(ws), ws_bytes, rows, D, _stream())
    _lib.check(rc, f"layernorm_bwd rows={rows} D={D}")
    return dx


def colsum(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out (fp32 [cols]) += column sums of x (bf16 [rows, cols])."""
    _chk2d(x, "x")
    rows, cols = x.shape
    nbytes = _lib.lib().libra_colsum_workspace_bytes(rows, cols)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x.device)
    rc = _lib.lib().libra_colsum_bf16(x.data_ptr(), x.stride(0), rows, cols, out.data_ptr(), ws.data_ptr(), nbytes,
                                      _stream())
    _lib.check(rc, "colsum")
    return out


def transpose_tokens(x: torch.Tensor, B: int, T: int, T_pad: int, *, out: Optional[torch.Tensor] = None):
    """x [B*T, C] (a column slice is fine) -> out [C, B*T_pad] with token (b,t) at column b*T_pad+t, zero padded."""
    _chk2d(x, "x")
    cols = x.shape[1]
    if x.shape[0] != B * T:
        raise ValueError("transpose_tokens: rows != B*T")
    if out is None:
        out = torch.empty((cols, B * T_pad), dtype=BF16, device=x.device)
    rc = _lib.lib().libra_transpose_bf16(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), T, cols, T_pad,
                                         None, B, T * x.stride(0), T_pad, _stream())
    _lib.check(rc, "transpose_tokens")
    return out


def patch_im2col(pixel: torch.Tensor, P: int, Kpad: int) -> torch.Tensor:
    if pixel.dim() != 4 or pixel.dtype != BF16 or not pixel.is_contiguous():
        raise ValueError("patch_im2col: pixel must be contiguous bf16 [B,C,H,W]")
    B, Cc, H, W = pixel.shape
    n = B * (H // P) * (W // P)
    cols = alloc_rows(n, Kpad, pixel.device)[:n]
    rc = _lib.lib().libra_patch_im2col(pixel.data_ptr(), cols.data_ptr(), B, Cc, H, W, P, Kpad, _stream())
    _lib.check(rc, "patch_im2col")
    return cols


def patch_col2im(dcols: torch.Tensor, B: int, Cc: int, H: int, W: int, P: int) -> torch.Tensor:
    _chk2d(dcols, "dcols")
    dpix = torch.empty((B, Cc, H, W), dtype=BF16, device=dcols.device)
    rc = _lib.lib().libra_patch_col2im(dcols.data_ptr(), dpix.data_ptr(), B, Cc, H, W, P, dcols.stride(0), _stream())
    _lib.check(rc, "patch_col2im")
    return dpix


def vit_embed_ln(patches, cls, pos, gamma, beta, B: int, T: int, eps: float, *, save: bool = True):
    _chk2d(patches, "patches")
    D = patches.shape[1]
    dev = patches.device
    emb = torch.empty((B * T, D), dtype=BF16, device=dev) if save else None
    hs0 = torch.empty((B * T, D), dtype=BF16, device=dev)
    mean = torch.empty(B * T, dtype=torch.float32, device=dev) if save else None
    rstd = torch.empty(B * T, dtype=torch.float32, device=dev) if save else None
    rc = _lib.lib().libra_vit_embed_ln(patches.data_ptr(), cls.data_ptr(), pos.data_ptr(), gamma.data_ptr(),
                                       beta.data_ptr(), _ptr(emb), hs0.data_ptr(), _ptr(mean), _ptr(rstd), B, T, D,
                                       float(eps), _stream())
    _lib.check(rc, "vit_embed_ln")
    return emb, hs0, mean, rstd


def _chk_lo(out_lo, out, what):
    if out_lo is not None and (out_lo.shape != out.shape or out_lo.stride() != out.stride() or out_lo.dtype != BF16):
        raise ValueError(f"{what}: out_lo must have the shape / strides / dtype of out")


def vit_attn_fwd(qkv: torch.Tensor, B: int, T: int, H: int, scale: float, *, need_lse: bool = True,
                 out: Optional[torch.Tensor] = None, out_lo: Optional[torch.Tensor] = None):
    """out_lo (same layout as out): receives the rounding residual of the output, for the backward's D term."""
    _chk2d(qkv, "qkv")
    if out is None:
        out = torch.empty((B * T, H * 64), dtype=BF16, device=qkv.device)
    _chk_lo(out_lo, out, "vit_attn_fwd")
    lse = torch.empty((B, H, T), dtype=torch.float32, device=qkv.device) if need_lse else None
    rc = _lib.lib().libra_vit_attn_fwd(qkv.data_ptr(), qkv.stride(0), out.data_ptr(), out.stride(0), _ptr(lse), _ptr(out_lo), B, T,
                                       H, float(scale), _stream())
    _lib.check(rc, "vit_attn_fwd")
    return out, lse


def vit_attn_bwd(qkv, out, dout, lse, B: int, T: int, H: int, scale: float, *,
                 out_dqkv: Optional[torch.Tensor] = None, out_lo: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk2d(qkv, "qkv"); _chk2d(out, "out"); _chk2d(dout, "dout")
    dev = qkv.device
    delta = torch.empty((B, H, T), dtype=torch.float32, device=dev)
    if dout.stride(0) != out.stride(0):
        raise ValueError("vit_attn_bwd: out/dout leading dims differ")
    _chk_lo(out_lo, out, "vit_attn_bwd")
    rc = _lib.lib().libra_vit_attn_delta(out.data_ptr(), _ptr(out_lo), dout.data_ptr(), out.stride(0), delta.data_ptr(), B, T, H,
                                         _stream())
    _lib.check(rc, "vit_attn_delta")
    dqkv = torch.empty_like(qkv) if out_dqkv is None else out_dqkv
    rc = _lib.lib().libra_vit_attn_bwd(qkv.data_ptr(), qkv.stride(0), dout.data_ptr(), dout.stride(0), lse.data_ptr(),
                                       delta.data_ptr(), dqkv.data_ptr(), dqkv.stride(0), B, T, H, float(scale), _stream())
    _lib.check(rc, "vit_attn_bwd")
    return dqkv


def feature_select(hs: Sequence[torch.Tensor], B: int, T: int) -> torch.Tensor:
    """hs: list of [B*T, D] bf16 contiguous -> [B*(T-1), len(hs)*D] (CLS dropped, channel concat)."""
    n = len(hs)
    D = hs[0].shape[-1]
    for h in hs:
        if h.dtype != BF16 or not h.is_contiguous() or h.numel() != B * T * D:
            raise ValueError("feature_select: hidden states must be contiguous bf16 [B*T, D]")
    feat = torch.empty((B * (T - 1), n * D), dtype=BF16, device=hs[0].device)
    arr = (C.c_void_p * n)(*[h.data_ptr() for h in hs])
    rc = _lib.lib().libra_feature_select(arr, n, feat.data_ptr(), B, T, D, _stream())
    _lib.check(rc, "feature_select")
    return feat


def feature_select_bwd(dfeat: torch.Tensor, dhs: Sequence[torch.Tensor], accumulate: Sequence[bool], B: int, T: int):
    n = len(dhs)
    D = dhs[0].shape[-1]
    if dfeat.dtype != BF16 or not dfeat.is_contiguous() or dfeat.numel() != B * (T - 1) * n * D:
        raise ValueError("feature_select_bwd: dfeat must be contiguous bf16 [B*(T-1), n*D]")
    arr = (C.c_void_p * n)(*[h.data_ptr() for h in dhs])
    acc = (C.c_int * n)(*[int(a) for a in accumulate])
    rc = _lib.lib().libra_feature_select_bwd(dfeat.data_ptr(), arr, acc, n, B, T, D, _stream())
    _lib.check(rc, "feature_select_bwd")


def lfq_encode(h: torch.Tensor, w_in, b_in, w_out, b_out, *, B: int, hw: int, Q: int, offset: int, boi: int,
               eoi: int, want_ids: bool = True, want_xpre: bool = False, want_quant: bool = False):
    _chk2d(h, "h")
    if h.shape[0] != B * hw:
        raise ValueError("lfq_encode: h must be [B*hw, E]")
    E = h.shape[1]
    dev = h.device
    indices = torch.empty((B * hw, Q), dtype=torch.int64, device=dev)
    ids = torch.empty((Q, B, hw + 2), dtype=torch.int64, device=dev) if want_ids else None
    xpre = torch.empty((B * hw, Q * 9), dtype=BF16, device=dev) if want_xpre else None
    quant = torch.empty((B * hw, E), dtype=BF16, device=dev) if want_quant else None
    rc = _lib.lib().libra_lfq_encode(h.data_ptr(), h.stride(0), _ptr(w_in), _ptr(b_in), _ptr(w_out), _ptr(b_out),
                                     indices.data_ptr(), _ptr(ids), _ptr(xpre), _ptr(quant), B, hw, E, Q, offset, boi,
                                     eoi, _stream())
    _lib.check(rc, "lfq_encode")
    return indices, ids, xpre, quant


def f32_to_bf16(x: torch.Tensor) -> torch.Tensor:
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    rc = _lib.lib().libra_f32_to_bf16(x.data_ptr(), out.data_ptr(), x.numel(), _stream())
    _lib.check(rc, "f32_to_bf16")
    return out


def add_(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a += b (bf16, contiguous)."""
    if a.shape != b.shape or not (a.is_contiguous() and b.is_contiguous()):
        raise ValueError("add_: contiguous equal shapes required")
    rc = _lib.lib().libra_add_bf16(a.data_ptr(), b.data_ptr(), a.data_ptr(), a.numel(), _stream())
    _lib.check(rc, "add_bf16")
    return a


# ---- VQ image decoder rows ------------------------------------------------------------------
def lfq_codes(indices: torch.Tensor, nbits: int, ldc: int) -> torch.Tensor:
    """indices int64 [M, Q] -> codes bf16 [M, ldc]: +-1 per bit (MSB first), zero beyond Q * nbits."""
    if indices.dtype != torch.int64 or indices.dim() != 2 or not indices.is_contiguous() or not indices.is_cuda:
        raise ValueError("lfq_codes: indices must be a contiguous cuda int64 [M, Q] tensor")
    M, Q = indices.shape
    out = torch.empty((M, ldc), dtype=BF16, device=indices.device)
    rc = _lib.lib().libra_lfq_codes(indices.data_ptr(), out.data_ptr(), M, Q, nbits, ldc, _stream())
    _lib.check(rc, "lfq_codes")
    return out


def groupnorm_affine(x: torch.Tensor, gamma, beta, B: int, HW: int, G: int, eps: float):
    """x [B*HW, C] bf16 contiguous -> (scale, shift) fp32 [B, C]: GroupNorm(G) folded into y = x * scale + shift."""
    _chk2d(x, "x")
    C = x.shape[1]
    if x.shape[0] != B * HW or not x.is_contiguous():
        raise ValueError("groupnorm_affine: x must be contiguous [B*HW, C]")
    scale = torch.empty((B, C), dtype=torch.float32, device=x.device)
    shift = torch.empty((B, C), dtype=torch.float32, device=x.device)
    nbytes = _lib.lib().libra_groupnorm_workspace_bytes(B, HW, C)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x.device)
    rc = _lib.lib().libra_groupnorm_affine(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), scale.data_ptr(), shift.data_ptr(), B, HW,
                                           C, G, float(eps), ws.data_ptr(), nbytes, _stream())
    _lib.check(rc, "groupnorm_affine")
    return scale, shift


def conv_gather(x: torch.Tensor, B: int, Hs: int, Ws: int, H: int, W: int, ksize: int, ldo: int, *, scale=None, shift=None,
                swish: bool = False, inv_scale: float = 1.0, pad_rows: int = 0) -> torch.Tensor:
    """x [B*Hs*Ws, C] (NHWC) -> the GEMM operand [B*H*W (+ zeroed pad rows), ldo] of a ksize x ksize conv on the nearest-upsampled
    [H, W] image, with the preceding GroupNorm affine / swish applied on the fly."""
    _chk2d(x, "x")
    C = x.shape[1]
    if x.shape[0] != B * Hs * Ws or not x.is_contiguous():
        raise ValueError("conv_gather: x must be contiguous [B*Hs*Ws, C]")
    M = B * H * W
    out = torch.empty((M + pad_rows, ldo), dtype=BF16, device=x.device)
    if pad_rows:
        out[M:].zero_()
    rc = _lib.lib().libra_conv_gather(x.data_ptr(), out.data_ptr(), ldo, _ptr(scale), _ptr(shift), int(swish), B, Hs, Ws, C, H, W,
                                      ksize, float(inv_scale), float(inv_scale), _stream())
    _lib.check(rc, "conv_gather")
    return out[:M] if pad_rows else out


def softmax_rows_(x: torch.Tensor, cols: int, scale: float) -> torch.Tensor:
    """in place: x[r, :cols] = softmax(bf16(x[r, :cols] * scale)), x[r, cols:] = 0."""
    _chk2d(x, "x")
    rc = _lib.lib().libra_softmax_rows(x.data_ptr(), x.shape[0], cols, x.stride(0), float(scale), _stream())
    _lib.check(rc, "softmax_rows")
    return x


# ---- image input pipeline -----------------------------------------------------------------------
def resample_h_u8(img: torch.Tensor, pad_y: int, pad_x: int, bg, bounds: torch.Tensor, coeffs: torch.Tensor, rows: int, row0: int):
    """img uint8 [H, W, 3] cuda -> uint8 [rows, out_w, 3]: Pillow's horizontal 8-bit pass over canvas rows [row0, row0 + rows)."""
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or not img.is_contiguous() or not img.is_cuda:
        raise ValueError("resample_h_u8: a contiguous cuda uint8 [H, W, 3] image is expected")
    out_w, ksize = coeffs.shape
    out = torch.empty((rows, out_w, 3), dtype=torch.uint8, device=img.device)
    rc = _lib.lib().libra_resample_h_u8(img.data_ptr(), img.shape[0], img.shape[1], pad_y, pad_x, int(bg[0]), int(bg[1]), int(bg[2]),
                                        bounds.data_ptr(), coeffs.data_ptr(), ksize, out.data_ptr(), rows, out_w, row0, _stream())
    _lib.check(rc, "resample_h_u8")
    return out


def resample_v_u8_norm(tmp: torch.Tensor, row0: int, bounds: torch.Tensor, coeffs: torch.Tensor, top: int, left: int, crop: int,
                       lut: torch.Tensor, out: torch.Tensor, patch: int = 0, kpad: int = 0):
    """Pillow's vertical pass + center crop + normalisation LUT + layout (NCHW [3,crop,crop] or im2col rows), into `out`."""
    rc = _lib.lib().libra_resample_v_u8_norm(tmp.data_ptr(), tmp.shape[1], row0, bounds.data_ptr(), coeffs.data_ptr(), coeffs.shape[1],
                                             top, left, crop, lut.data_ptr(), out.data_ptr(), patch, kpad, _stream())
    _lib.check(rc, "resample_v_u8_norm")
    return out


def adamw_step(master, m, v, grad, param, *, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float,
               bias_corr1: float, bias_corr2: float, grad_scale: float = 1.0, grad_norm_sq: Optional[torch.Tensor] = None,
               max_grad_norm: float = 0.0):
    """One fused AdamW update of a flat range: master / m / v fp32 [n], grad / param bf16 [n] (param = bf16(master)).
    grad_norm_sq (fp32 device scalar) + max_grad_norm: global-norm clipping, the coefficient is formed on the device."""
    n = master.numel()
    for t, dt, nm in ((master, torch.float32, "master"), (m, torch.float32, "m"), (v, torch.float32, "v"),
                      (grad, BF16, "grad"), (param, BF16, "param")):
        if t.dtype != dt or t.numel() != n or not t.is_contiguous() or not t.is_cuda:
            raise ValueError(f"adamw_step: {nm} must be a contiguous cuda {dt} tensor of {n} elements "
                             f"(got {t.dtype} {tuple(t.shape)} {t.device}); there is no CPU optimizer path")
    rc = _lib.lib().libra_adamw_step(master.data_ptr(), m.data_ptr(), v.data_ptr(), grad.data_ptr(), param.data_ptr(), n,
                                     float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                     float(bias_corr1), float(bias_corr2), float(grad_scale), _ptr(grad_norm_sq),
                                     float(max_grad_norm), _stream())
    _lib.check(rc, "adamw_step")


def sumsq(x: torch.Tensor, out: torch.Tensor, *, accumulate: bool = False) -> torch.Tensor:
    """out (fp32 device scalar) (+)= sum of squares of the contiguous bf16 tensor x; deterministic."""
    if x.dtype != BF16 or not x.is_contiguous() or not x.is_cuda or out.dtype != torch.float32 or out.numel() != 1:
        raise ValueError("sumsq: x must be a contiguous cuda bf16 tensor, out an fp32 scalar tensor")
    n = x.numel()
    nbytes = _lib.lib().libra_sumsq_workspace_bytes(n)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x.device)
    rc = _lib.lib().libra_sumsq_bf16(x.data_ptr(), n, out.data_ptr(), int(accumulate), ws.data_ptr(), nbytes, _stream())
    _lib.check(rc, "sumsq")
    return out


# ---- optional per-launch timing (bench.py's roofline leg) ---------------------------------------------
class LaunchProfile:
    """Context manager: while active, every gemm_nt / vit_attn_* call is bracketed by events on the current
    stream (the stream the kernel is launched on) and (kind, work, ms) is available after .finish()."""
    active = None

    def __init__(self):
        self.records = []

    def __enter__(self):
        LaunchProfile.active = self
        return self

    def __exit__(self, *a):
        LaunchProfile.active = None

    def bracket(self, kind, work):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.records.append([kind, work, s, e])
        return s, e

    def finish(self):
        torch.cuda.synchronize()
        return [(k, w, s.elapsed_time(e)) for k, w, s, e in self.records]


def _profiled(kind, work_fn):
    def deco(fn):
        def wrapper(*a, **kw):
            prof = LaunchProfile.active
            if prof is None:
                return fn(*a, **kw)
            s, e = prof.bracket(kind, work_fn(*a, **kw))
            s.record()
            r = fn(*a, **kw)
            e.record()
            return r
        wrapper.__name__ = fn.__name__
        wrapper.__doc__ = fn.__doc__
        return wrapper
    return deco


def _gemm_flops(a, b, **kw):
    a_t, b_t = kw.get("a_t", False), kw.get("b_t", False)
    k = kw.get("k") or (a.shape[0] if a_t else a.shape[1])
    m = kw["a_rows"].numel() if kw.get("a_rows") is not None else (a.shape[1] if a_t else a.shape[0])
    n = b.shape[1] if b_t else b.shape[0]
    return (2.0 * m * n * k, 2.0 * (m * k + n * k + m * n),          # (FLOP, algorithmic operand + result bytes, shape tag)
            f"{m}x{n}x{k}{' aT' if a_t else ''}{' bT' if b_t else ''}")


gemm_nt = _profiled("gemm", _gemm_flops)(gemm_nt)


# ---- routed decoder rows -------------------------------------------------------------------------------
def rmsnorm_routed(x, w_lang, w_vis, flag, eps: float, *, out=None, save_rstd: bool = False):
    _chk2d(x, "x")
    rows, D = x.shape
    y = torch.empty((rows, D), dtype=BF16, device=x.device) if out is None else out
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_rstd else None
    rc = _lib.lib().libra_rmsnorm_routed_fwd(x.data_ptr(), x.stride(0), w_lang.data_ptr(), _ptr(w_vis), _ptr(flag),
                                             y.data_ptr(), y.stride(0), _ptr(rstd), rows, D, float(eps), _stream())
    _lib.check(rc, "rmsnorm_routed")
    return (y, rstd) if save_rstd else y


def rope_bridge(qkv, tb, bk_l, bk_v, bv_l, bv_v, flag, cos, sin, S: int, H: int):
    _chk2d(qkv, "qkv"); _chk2d(tb, "tb")
    N = qkv.shape[0]
    kc = torch.empty((N, H * 128), dtype=BF16, device=qkv.device)
    vc = torch.empty((N, H * 128), dtype=BF16, device=qkv.device)
    rc = _lib.lib().libra_rope_bridge(qkv.data_ptr(), qkv.stride(0), tb.data_ptr(), tb.stride(0), bk_l.data_ptr(),
                                      bk_v.data_ptr(), bv_l.data_ptr(), bv_v.data_ptr(), flag.data_ptr(), cos.data_ptr(),
                                      sin.data_ptr(), cos.shape[0], kc.data_ptr(), vc.data_ptr(), kc.stride(0), N, S, H,
                                      _stream())
    _lib.check(rc, "rope_bridge")
    return kc, vc


def rope_bridge_pos(qkv, tb, bk_l, bk_v, bv_l, bv_v, flag, cos, sin, positions, H: int, *, append=None):
    """rope_bridge with explicit int32 positions: [N] (cached decode step, left-padded prompts) or [N, 2] (use_2d_rope: even heads
    rotate by column 0, odd heads by column 1).  `append` = (caches, slot): a generation step (row n = sequence n) - the four K / V
    rows of every new token are also stored at `slot` (cuda int64 [1]) of the layer's four caches [B, Lmax, H*128] in this launch."""
    _chk2d(qkv, "qkv"); _chk2d(tb, "tb")
    N = qkv.shape[0]
    if positions.dtype != torch.int32 or positions.numel() not in (N, 2 * N) or not positions.is_contiguous():
        raise ValueError("rope_bridge_pos: positions must be contiguous int32 [N] or [N, 2]")
    pstride = positions.numel() // N
    kc = torch.empty((N, H * 128), dtype=BF16, device=qkv.device)
    vc = torch.empty((N, H * 128), dtype=BF16, device=qkv.device)
    args = (qkv.data_ptr(), qkv.stride(0), tb.data_ptr(), tb.stride(0), bk_l.data_ptr(), bk_v.data_ptr(), bv_l.data_ptr(),
            bv_v.data_ptr(), flag.data_ptr(), cos.data_ptr(), sin.data_ptr(), cos.shape[0], kc.data_ptr(), vc.data_ptr(), kc.stride(0),
            N, positions.data_ptr(), pstride, H)
    if append is None:
        rc = _lib.lib().libra_rope_bridge_pos(*args, _stream())
    else:
        caches, slot = append
        c0 = caches[0]
        for c in caches:
            if c.dim() != 3 or c.shape[0] != N or c.shape[2] != H * 128 or c.stride(2) != 1 or c.dtype != BF16 or c.stride() != c0.stride():
                raise ValueError("rope_bridge_pos: caches must be four [B, Lmax, H*128] bf16 buffers with identical strides, B = rows")
        if slot.dtype != torch.int64 or slot.numel() != 1 or not slot.is_cuda:
            raise ValueError("rope_bridge_pos: slot must be a cuda int64 tensor with one element")
        rc = _lib.lib().libra_rope_bridge_pos_append(*args, caches[0].data_ptr(), caches[1].data_ptr(), caches[2].data_ptr(),
                                                     caches[3].data_ptr(), c0.stride(1), c0.stride(0), slot.data_ptr(), _stream())
    _lib.check(rc, "rope_bridge_pos")
    return kc, vc


def bridge_attn_decode(q, k_same, k_cross, v_same, v_cross, key_flag, query_flag, kv_len, H: int, scale: float, *,
                       kv_start=None):
    """One new query token per sequence against the KV cache: q [B, H*128]; caches [B, Lmax, H*128] (same strides);
    key_flag [B, Lmax] u8, query_flag [B] u8, kv_len [B] int32 (valid cached tokens incl. the new one, <= Lmax: the caller's
    invariant - checking it here would be a host synchronisation inside a capturable step) -> [B, H*128]."""
    _chk2d(q, "q")
    B = q.shape[0]
    for t in (k_same, k_cross, v_same, v_cross):
        if t.dim() != 3 or t.shape[0] != B or t.shape[2] != H * 128 or t.stride(2) != 1 or t.dtype != BF16 or \
                t.stride() != k_same.stride():
            raise ValueError("bridge_attn_decode: caches must be [B, Lmax, H*128] bf16 with identical strides")
    if key_flag.dtype != torch.uint8 or query_flag.dtype != torch.uint8 or kv_len.dtype != torch.int32:
        raise ValueError("bridge_attn_decode: flags are uint8, kv_len is int32")
    if kv_start is not None and (kv_start.dtype != torch.int32 or kv_start.numel() != B):
        raise ValueError("bridge_attn_decode: kv_start must be int32 [B]")
    out = torch.empty((B, H * 128), dtype=BF16, device=q.device)
    nbytes = _lib.lib().libra_bridge_attn_decode_workspace_bytes(B, H)
    ws = torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device=q.device)      # partial states of the key splits
    rc = _lib.lib().libra_bridge_attn_decode(q.data_ptr(), q.stride(0), k_same.data_ptr(), k_cross.data_ptr(), v_same.data_ptr(),
                                             v_cross.data_ptr(), k_same.stride(1), k_same.stride(0), key_flag.data_ptr(),
                                             key_flag.stride(0), query_flag.data_ptr(), kv_len.data_ptr(), _ptr(kv_start),
                                             out.data_ptr(), out.stride(0), B, H, float(scale), ws.data_ptr(), nbytes, _stream())
    _lib.check(rc, "bridge_attn_decode")
    return out


def kv_cache_append(rows, caches, slot: torch.Tensor):
    """caches[x][b, slot] = rows[x][b] for the four K/V buffers of a layer (one launch).  rows: four [B, W] bf16 tensors (column
    slices allowed), caches: four [B, Lmax, W] buffers with identical strides, slot: int64 device tensor [1]."""
    B, W = rows[0].shape
    c0 = caches[0]
    for r in rows:
        _chk2d(r, "kv_cache_append rows")
    for c in caches:
        if c.dim() != 3 or c.shape[0] != B or c.shape[2] != W or c.stride(2) != 1 or c.dtype != BF16 or c.stride() != c0.stride():
            raise ValueError("kv_cache_append: caches must be [B, Lmax, W] bf16 with identical strides")
    if slot.dtype != torch.int64 or slot.numel() != 1 or not slot.is_cuda:
        raise ValueError("kv_cache_append: slot must be a cuda int64 tensor with one element")
    rc = _lib.lib().libra_kv_cache_append(rows[0].data_ptr(), rows[0].stride(0), rows[1].data_ptr(), rows[1].stride(0),
                                          rows[2].data_ptr(), rows[2].stride(0), rows[3].data_ptr(), rows[3].stride(0),
                                          caches[0].data_ptr(), caches[1].data_ptr(), caches[2].data_ptr(), caches[3].data_ptr(),
                                          c0.stride(1), c0.stride(0), slot.data_ptr(), B, W, _stream())
    _lib.check(rc, "kv_cache_append")


def bridge_attn_fwd(q, k_same, k_cross, v_same, v_cross, flag, kv_len, B: int, S: int, H: int, scale: float, *,
                    need_lse: bool = False, kv_start=None, out_lo: Optional[torch.Tensor] = None):
    """kv_len [B] int32 = end of the valid keys (right padding); kv_start [B] int32 = first valid key (left padding)."""
    for t, n in ((q, "q"), (k_same, "k_same"), (k_cross, "k_cross"), (v_same, "v_same"), (v_cross, "v_cross")):
        _chk2d(t, n)
    out = torch.empty((B * S, H * 128), dtype=BF16, device=q.device)
    _chk_lo(out_lo, out, "bridge_attn_fwd")
    lse = torch.empty((B, H, S), dtype=torch.float32, device=q.device) if need_lse else None
    rc = _lib.lib().libra_bridge_attn_fwd(q.data_ptr(), q.stride(0), k_same.data_ptr(), k_same.stride(0), k_cross.data_ptr(),
                                          k_cross.stride(0), v_same.data_ptr(), v_same.stride(0), v_cross.data_ptr(),
                                          v_cross.stride(0), flag.data_ptr(),
                                          _ptr(kv_len), _ptr(kv_start), out.data_ptr(), out.stride(0), _ptr(lse), _ptr(out_lo),
                                          B, S, H, float(scale), _stream())
    _lib.check(rc, "bridge_attn_fwd")
    return out, lse


def swiglu(gate, up, *, out=None):
    _chk2d(gate, "gate"); _chk2d(up, "up")
    rows, I = gate.shape
    if gate.stride(0) != up.stride(0):
        raise ValueError("swiglu: gate/up must share a row stride")
    y = torch.empty((rows, I), dtype=BF16, device=gate.device) if out is None else out
    rc = _lib.lib().libra_swiglu(gate.data_ptr(), up.data_ptr(), gate.stride(0), y.data_ptr(), y.stride(0), rows, I, _stream())
    _lib.check(rc, "swiglu")
    return y


def gather_rows(table, idx, sub: int, rows_sel, n: int, out, col0: int = 0):
    rc = _lib.lib().libra_gather_rows(table.data_ptr(), table.shape[1], idx.data_ptr(), sub, _ptr(rows_sel), n,
                                      out.data_ptr(), out.stride(0), col0, _stream())
    _lib.check(rc, "gather_rows")
    return out


def copy_rows(src, rows_sel, n: int, out, col0: int = 0):
    rc = _lib.lib().libra_copy_rows(src.data_ptr(), src.stride(0), src.shape[1], _ptr(rows_sel), n, out.data_ptr(),
                                    out.stride(0), col0, _stream())
    _lib.check(rc, "copy_rows")
    return out


def ce_rows(logits, target, target_sub: int = 0):
    _chk2d(logits, "logits")
    rows, V = logits.shape
    loss = torch.empty(rows, dtype=torch.float32, device=logits.device)
    rc = _lib.lib().libra_ce_rows(logits.data_ptr(), logits.stride(0), V, target.data_ptr(), target_sub, loss.data_ptr(),
                                  rows, _stream())
    _lib.check(rc, "ce_rows")
    return loss


def bridge_attn_bwd(q, k_same, k_cross, v_same, v_cross, out, dout, flag, kv_len, lse, B: int, S: int, H: int, scale: float, *,
                    out_lo: Optional[torch.Tensor] = None):
    """-> dq, dk_same, dk_cross, dv_same, dv_cross  (each [B*S, H*128] bf16)."""
    for t, n in ((q, "q"), (k_same, "k_same"), (k_cross, "k_cross"), (v_same, "v_same"), (v_cross, "v_cross"), (out, "out"),
                 (dout, "dout")):
        _chk2d(t, n)
    _chk_lo(out_lo, out, "bridge_attn_bwd")
    dev = q.device
    N, HD = B * S, H * 128
    g = [torch.empty((N, HD), dtype=BF16, device=dev) for _ in range(5)]
    delta = torch.empty((B, H, S), dtype=torch.float32, device=dev)
    rc = _lib.lib().libra_bridge_attn_bwd(q.data_ptr(), q.stride(0), k_same.data_ptr(), k_same.stride(0), k_cross.data_ptr(),
                                          k_cross.stride(0), v_same.data_ptr(), v_same.stride(0), v_cross.data_ptr(),
                                          v_cross.stride(0), out.data_ptr(), _ptr(out_lo), out.stride(0), dout.data_ptr(), dout.stride(0),
                                          flag.data_ptr(), _ptr(kv_len), lse.data_ptr(), delta.data_ptr(), g[0].data_ptr(),
                                          g[0].stride(0), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), g[4].data_ptr(),
                                          HD, B, S, H, float(scale), errors.word(dev).data_ptr(), _stream())
    _lib.check(rc, "bridge_attn_bwd")
    return tuple(g)


# ---- routed decoder, backward rows -----------------------------------------------------------------------
def ce_rows_bwd(logits, t0, t1, sub: int, c0: float, c1: float, out, scale=None):
    """scale (fp32 device scalar, optional): multiplies both coefficients inside the kernel (no host read of it)."""
    _chk2d(logits, "logits"); _chk2d(out, "out")
    rows, V = logits.shape
    if scale is not None and (scale.dtype != torch.float32 or scale.numel() != 1 or scale.device != logits.device):
        raise ValueError("ce_rows_bwd: scale must be a one-element fp32 tensor on the logits' device")
    rc = _lib.lib().libra_ce_rows_bwd(logits.data_ptr(), logits.stride(0), V, _ptr(t0), _ptr(t1), sub, float(c0), float(c1),
                                      _ptr(scale), out.data_ptr(), out.stride(0), rows, _stream())
    _lib.check(rc, "ce_rows_bwd")
    return out


def rmsnorm_routed_bwd(dy, x, w_lang, w_vis, flag, rstd, *, dres=None, out=None):
    _chk2d(dy, "dy"); _chk2d(x, "x")
    rows, D = x.shape
    dx = torch.empty((rows, D), dtype=BF16, device=x.device) if out is None else out
    rc = _lib.lib().libra_rmsnorm_routed_bwd(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), w_lang.data_ptr(),
                                             _ptr(w_vis), _ptr(flag), rstd.data_ptr(), _ptr(dres),
                                             dres.stride(0) if dres is not None else 0, dx.data_ptr(), dx.stride(0), rows, D,
                                             _stream())
    _lib.check(rc, "rmsnorm_routed_bwd")
    return dx


def rmsnorm_routed_wgrad(dy, x, rstd, flag, dw_lang, dw_vis, *, rows_sel=None):
    """dw_lang / dw_vis (fp32 [D], either may be None) += per-modality sums of dy * x * rstd.
    rows_sel (int32 [n], optional): only these rows are visited (e.g. the vision rows when only the vision weight is wanted)."""
    rows, D = x.shape
    if rows_sel is not None and (rows_sel.dtype != torch.int32 or not rows_sel.is_contiguous()):
        raise ValueError("rmsnorm_routed_wgrad: rows_sel must be a contiguous int32 index tensor")
    if rows_sel is not None and rows_sel.numel() == 0:
        return                                               # an empty selection adds nothing (an empty tensor has no pointer to pass)
    nbytes = _lib.lib().libra_rmsnorm_wgrad_workspace_bytes(rows, D)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x.device)
    rc = _lib.lib().libra_rmsnorm_routed_wgrad(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), rstd.data_ptr(),
                                               _ptr(flag), _ptr(dw_lang), _ptr(dw_vis), ws.data_ptr(), nbytes, rows, D,
                                               _ptr(rows_sel), rows_sel.numel() if rows_sel is not None else 0, _stream())
    _lib.check(rc, "rmsnorm_routed_wgrad")


def rank_outer_wgrad(x, coef, flag, *, transpose_out: bool, want_l: bool = True, want_v: bool = True):
    """(out_l, out_v): out_m[j][c] = sum over the tokens of modality m of coef[t][j] * x[t][c]  (bf16, fp32 accumulation).
    x [N, C], coef [N, 8 or 16] (a column slice is fine), flag uint8 [N] or None.  transpose_out: out_m is [C, ncoef] (weight_B
    layout) instead of [ncoef, C] (weight_A layout).  The rank-8 bridge weight gradients in one pass over x."""
    _chk2d(x, "x"); _chk2d(coef, "coef")
    N, Cc = x.shape
    nc = coef.shape[1]
    if coef.shape[0] != N or nc not in (8, 16):
        raise ValueError(f"rank_outer_wgrad: coef must be [N, 8] or [N, 16], got {tuple(coef.shape)} for x {tuple(x.shape)}")
    shape = (Cc, nc) if transpose_out else (nc, Cc)
    out_l = torch.empty(shape, dtype=BF16, device=x.device) if want_l else None
    out_v = torch.empty(shape, dtype=BF16, device=x.device) if want_v else None
    nbytes = _lib.lib().libra_rank_outer_wgrad_workspace_bytes(max(N, 1), Cc, nc)
    ws = torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device=x.device)
    rc = _lib.lib().libra_rank_outer_wgrad(x.data_ptr(), x.stride(0), coef.data_ptr(), coef.stride(0), nc, _ptr(flag), _ptr(out_l),
                                           _ptr(out_v), nc if transpose_out else Cc, int(transpose_out), N, Cc, ws.data_ptr(), nbytes,
                                           _stream())
    _lib.check(rc, "rank_outer_wgrad")
    return out_l, out_v


def swiglu_bwd(dy, gate, up, dgate, dup):
    rows, I = gate.shape
    rc = _lib.lib().libra_swiglu_bwd(dy.data_ptr(), dy.stride(0), gate.data_ptr(), up.data_ptr(), gate.stride(0),
                                     dgate.data_ptr(), dup.data_ptr(), dgate.stride(0), rows, I, _stream())
    _lib.check(rc, "swiglu_bwd")


def rope_bridge_bwd(dq, dks, dkc, dvs, dvc, cos, sin, S: int, H: int, dqkv, dkb, *, bridge_b=None, flag=None, dtb=None,
                    positions=None):
    """bridge_b = (bk_l, bk_v, bv_l, bv_v) as weight_B^T [8, H*128] + flag + dtb [N, >=16]: also writes the rank-8 bridge
    activation gradients."""
    N = dq.shape[0]
    if dtb is not None:
        _chk2d(dtb, "dtb")
        bk_l, bk_v, bv_l, bv_v = bridge_b
        for t in bridge_b:
            if tuple(t.shape) != (8, H * 128) or not t.is_contiguous():
                raise ValueError(f"rope_bridge_bwd: bridge operands are weight_B^T [8, {H * 128}] contiguous, got {tuple(t.shape)}")
        extra = (bk_l.data_ptr(), bk_v.data_ptr(), bv_l.data_ptr(), bv_v.data_ptr(), flag.data_ptr(), dtb.data_ptr(), dtb.stride(0))
    else:
        extra = (None, None, None, None, None, None, 0)
    pstride = 1
    if positions is not None:
        if positions.dtype != torch.int32 or positions.numel() not in (N, 2 * N) or not positions.is_contiguous():
            raise ValueError("rope_bridge_bwd: positions must be contiguous int32 [N] or [N, 2]")
        pstride = positions.numel() // N
    rc = _lib.lib().libra_rope_bridge_bwd(dq.data_ptr(), dks.data_ptr(), dkc.data_ptr(), dvs.data_ptr(), dvc.data_ptr(),
                                          dq.stride(0), cos.data_ptr(), sin.data_ptr(), cos.shape[0], dqkv.data_ptr(),
                                          dqkv.stride(0), dkb.data_ptr(), dkb.stride(0), *extra, N, S, H, _ptr(positions), pstride,
                                          _stream())
    _lib.check(rc, "rope_bridge_bwd")


# ---- compute-unit budget (include/libra_hip.h, "compute-unit budget") --------------------------------------------------------
def cu_count() -> int:
    return int(_lib.lib().libra_get_cu_count())


def set_cu_budget(cus: int) -> int:
    """Persistent kernels launch at most `cus` workgroups (0 = one per physical CU).  Returns the previous budget."""
    prev = _lib.lib().libra_set_cu_budget(int(cus))
    if prev < 0:
        _lib.check(prev, "set_cu_budget")
    return int(prev)


class ReservedCUStream:
    """A compute stream that leaves `reserve` CUs to other streams (RCCL's reduction kernels under backward) + the matching budget
    for the persistent kernels.  `with rs: step()` runs the step on it; close() restores the budget and destroys the stream."""

    def __init__(self, reserve: int):
        import ctypes
        h, cus = ctypes.c_void_p(), ctypes.c_int32()
        _lib.check(_lib.lib().libra_stream_create_cu_reserved(int(reserve), ctypes.byref(h), ctypes.byref(cus)), "stream_create_cu_reserved")
        self.handle, self.cus, self.reserve = h.value, int(cus.value), int(reserve)
        self.stream = torch.cuda.ExternalStream(self.handle)
        # persistent kernels: one workgroup per CU only where every shader engine still has its full CU count - the hardware deals
        # workgroups to an XCC's four SEs round-robin, so the grid is sized for the SE-symmetric part (multiples of 32 reserved)
        self.persistent_cus = cu_count() - 32 * ((int(reserve) + 31) // 32)
        self._prev = set_cu_budget(max(self.persistent_cus, 32))
        self._ctx = None

    def __enter__(self):
        self.stream.wait_stream(torch.cuda.current_stream())
        self._ctx = torch.cuda.stream(self.stream)
        self._ctx.__enter__()
        return self

    def __exit__(self, *a):
        r = self._ctx.__exit__(*a)
        torch.cuda.current_stream().wait_stream(self.stream)
        return r

    def close(self):
        if self.handle:
            self.stream.synchronize()
            set_cu_budget(self._prev)
            _lib.check(_lib.lib().libra_stream_destroy(self.handle), "stream_destroy")
            self.handle = None


# ---- per-launch timing of the attention and row kernels (bench.py's `roofline.by_kernel`) ----------------------------------
# work = (FLOP, algorithmic HBM bytes, tag): attention FLOPs are the causal-minimal count bench.py's roofline uses
# (2 products x 2 S(S+1)/2 x 128 per head forward, 2.5 x that backward); row kernels count each operand / result row once.
def _attn_work(mult, tag):
    def work(q, k_same, k_cross, v_same, v_cross, *rest, **kw):
        B, S, H = [x for x in rest if isinstance(x, int)][:3]
        n = q.shape[0] * H * 128 * 2
        return (mult * B * H * 4.0 * (S * (S + 1) / 2) * 128, (5 if mult == 1.0 else 13) * n, tag)
    return work


def _rows_work(n_mats, tag, arg=0):
    def work(*a, **kw):
        t = a[arg]
        return (0.0, float(n_mats) * t.shape[0] * t.shape[1] * 2, tag)
    return work


bridge_attn_fwd = _profiled("attn_fwd", _attn_work(1.0, "bridge_attn_fwd"))(bridge_attn_fwd)
bridge_attn_bwd = _profiled("attn_bwd", _attn_work(2.5, "bridge_attn_bwd"))(bridge_attn_bwd)
rmsnorm_routed = _profiled("row", _rows_work(2, "rmsnorm_routed"))(rmsnorm_routed)
rmsnorm_routed_bwd = _profiled("row", _rows_work(4, "rmsnorm_routed_bwd"))(rmsnorm_routed_bwd)
swiglu = _profiled("row", _rows_work(3, "swiglu"))(swiglu)
swiglu_bwd = _profiled("row", _rows_work(5, "swiglu_bwd", arg=1))(swiglu_bwd)
# rope_bridge: reads q | k | v ([N, 3 H 128] = 3 matrices of dq's width), writes q, k_same in place and K_cross, V_cross
rope_bridge = _profiled("row", lambda qkv, *a, **kw: (0.0, 7.0 * qkv.shape[0] * (qkv.shape[1] // 3) * 2, "rope_bridge"))(rope_bridge)
rope_bridge_bwd = _profiled("row", _rows_work(8, "rope_bridge_bwd"))(rope_bridge_bwd)
ce_rows = _profiled("row", _rows_work(1, "ce_rows"))(ce_rows)
ce_rows_bwd = _profiled("row", _rows_work(2, "ce_rows_bwd"))(ce_rows_bwd)
