"""Data-parallel gradient exchange and the (optionally ZeRO-1-sharded) fused AdamW step of the hot path.

One process per GPU.  The trainable parameters' gradients live in **pre-allocated flat buckets with a fixed,
rank-independent layout** (`GradBuckets`): the engines' weight-gradient GEMMs write straight into bucket views
(`grad_out`), nothing is re-packed with `torch.cat`, and every rank launches exactly the same collectives whatever its
micro-batch contained (a rank whose batch has no vision token still contributes its zero gradients).

Overlap with backward: `vit_engine.backward` / `decoder_engine.backward` hand every layer's gradients to the capturing
store (`with buckets.capture(): loss.backward()`) right after the layer's last weight-gradient GEMM is enqueued; a bucket
goes out on RCCL's own stream the moment its last member arrived, so layer i's exchange runs under the backward kernels of
layers < i and only the last bucket is exposed.  Exchange modes:

* ``allreduce``  one `all_reduce(AVG)` per bucket (algorithm left to RCCL);
* ``rs_ag``      `reduce_scatter_tensor` + `all_gather_into_tensor` per bucket - on the xGMI full mesh the direct
                 exchange moves 2·(W-1)/W of the bucket per GPU over 7 links at once instead of a ring's per-link bound
                 (≈14 ms vs ≈98 ms for 8.5 GB);
* ``zero1``      reduce-scatter only; `FlatAdamW(shard=True)` updates the rank's 1/W shard (fp32 master + moments) with
                 the HIP fused AdamW kernel and all-gathers the bf16 parameters (DeepSpeed ZeRO-1/2 of the reference
                 recipes: libra/configs/deepspeed_configs/ZeRO-2.json:15-21, libra_instruction.yaml:66-67,82).

Gradient accumulation (libra_pretrain.yaml:96, `gradient_accumulation_steps: 4`): `capture(sync=False)` on all but the
last micro-step adds into the buckets without communicating.

Works unchanged on the gloo backend with CPU tensors (world_size-2 tests); the optimizer's arithmetic is the HIP kernel
(`kernels.adamw_step`) - there is no CPU implementation in the product (tests inject their own `update_fn`).
"""
from __future__ import annotations

import contextlib
from typing import Callable, Dict, Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist

# The capturing store.  A plain module global, not a thread-local: autograd runs backward on its own worker thread
# (SURVEY §8b "threading conventions"), and one rank drives one backward at a time (capture() refuses to nest).
_capturing = None

ALIGN = 64            # elements: every parameter's slot starts 128 B aligned (GEMM outputs need 16 B)


def emit(grads: Dict[str, torch.Tensor]) -> None:
    """Called by the engines after each layer: {state-dict name: finished gradient}."""
    if _capturing is not None:
        _capturing.add(grads)


def emit_new(g: Dict[str, torch.Tensor], seen: set) -> None:
    """emit() the entries of the running gradient dict `g` that were added since the last call (sorted: rank-independent)."""
    if _capturing is None:
        return
    fresh = {n: g[n] for n in sorted(g) if n not in seen}
    seen.update(fresh)
    _capturing.add(fresh)


def grad_out(name: str) -> Optional[torch.Tensor]:
    """The bucket view a kernel may write gradient `name` into directly (None: no store is capturing, the name is not
    exchanged, or this micro-step accumulates into an already filled bucket)."""
    if _capturing is None:
        return None
    return _capturing.out_view(name)


def is_captured(name: str) -> bool:
    """True when gradient `name` is owned by the capturing store: the autograd bridges then return None for it, so that
    autograd's AccumulateGrad never adds a bucket view onto itself (p.grad is installed by finish_into)."""
    return _capturing is not None and name in _capturing.where


def _is_nccl(group) -> bool:
    return dist.is_initialized() and dist.get_backend(group) == "nccl"


class _Bucket:
    __slots__ = ("flat", "names", "slots", "ready", "work", "launched", "shard")

    def __init__(self):
        self.names: List[str] = []
        self.slots: Dict[str, Tuple[int, int, torch.Size]] = {}
        self.ready = 0
        self.work = None
        self.launched = False


class GradBuckets:
    """Flat gradient buckets with a fixed layout over `named_params` (the parameters to exchange, e.g. the trainable ones).

    Order: `group_fn(name)` (the backward order of the engine: decoder_engine.emit_group / vit_engine.emit_group), then name.
    A bucket is closed at the first parameter boundary past `bucket_bytes`; its length is padded to a multiple of
    world_size · ALIGN so that it reduce-scatters into equal aligned shards."""

    def __init__(self, named_params: Iterable[Tuple[str, torch.Tensor]], *, bucket_bytes: int = 64 << 20,
                 group_fn: Optional[Callable[[str], int]] = None, mode: str = "allreduce", group=None,
                 dtype: torch.dtype = torch.bfloat16):
        if mode not in ("allreduce", "rs_ag", "zero1"):
            raise ValueError(f"unknown exchange mode {mode!r}")
        named = [(n, p) for n, p in named_params]
        if not named:
            raise ValueError("GradBuckets: no parameters")
        self.mode, self.group, self.dtype = mode, group, dtype
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = named[0][1].device
        key = (lambda n: (group_fn(n), n)) if group_fn is not None else (lambda n: (0, n))
        named.sort(key=lambda np_: key(np_[0]))
        esz = torch.empty((), dtype=dtype).element_size()
        quantum = self.world * ALIGN
        self.buckets: List[_Bucket] = []
        self.where: Dict[str, int] = {}
        cur, off = _Bucket(), 0
        sizes: List[int] = []

        def close():
            nonlocal cur, off
            sizes.append((off + quantum - 1) // quantum * quantum)
            self.buckets.append(cur)
            cur, off = _Bucket(), 0
        for n, p in named:
            k = p.numel()
            cur.names.append(n)
            cur.slots[n] = (off, k, p.shape)
            self.where[n] = len(self.buckets)
            off += (k + ALIGN - 1) // ALIGN * ALIGN
            if off * esz >= bucket_bytes:
                close()
        if cur.names:
            close()
        for b, n in zip(self.buckets, sizes):
            b.flat = torch.zeros(n, dtype=dtype, device=self.device)          # pad elements stay zero forever
            k = n // self.world
            b.shard = (self.rank * k, (self.rank + 1) * k)
        self.total_bytes = sum(sizes) * esz
        self._sync = True
        self._micro = 0                 # micro-steps accumulated into the buckets since the last exchange
        self._next = 0                  # next bucket (layout order) to send
        self._seen: set = set()
        self._autograd_owned: Dict[str, tuple] = {}     # name -> (parameter whose gradient autograd leaves in p.grad, installed grad)
        self.bytes_exchanged = 0
        self.launches = 0

    # ---- capture ------------------------------------------------------------------------------------------------
    @contextlib.contextmanager
    def capture(self, sync: bool = True):
        """Route the engines' per-layer emissions of the enclosed backward into the buckets.  `sync=False`: a gradient-
        accumulation micro-step that is not the last one - gradients are summed into the buckets, nothing is sent."""
        global _capturing
        if _capturing is not None:
            raise RuntimeError("another gradient store is already capturing")
        _capturing = self
        self._sync = sync
        self._seen = set()
        # a parameter outside the engines still holds the exchanged gradient finish_into() installed last step; left there,
        # autograd would accumulate this backward's gradient onto it: a captured backward starts from nothing, like the
        # engine-owned gradients do (whoever runs backward OUTSIDE the capture follows PyTorch's rule: zero_grad() first)
        for n, (p, installed) in self._autograd_owned.items():
            if p.grad is installed:
                p.grad = None
        for b in self.buckets:
            b.ready = 0
        self._next = 0
        try:
            yield self
        finally:
            _capturing = None

    def out_view(self, name: str) -> Optional[torch.Tensor]:
        i = self.where.get(name)
        if i is None or self._micro > 0:          # accumulating: the kernel must not overwrite what is already there
            return None
        off, k, shape = self.buckets[i].slots[name]
        return self.buckets[i].flat[off:off + k].view(shape)

    def view(self, name: str) -> torch.Tensor:
        b = self.buckets[self.where[name]]
        off, k, shape = b.slots[name]
        return b.flat[off:off + k].view(shape)

    def add(self, grads: Dict[str, torch.Tensor]):
        for n, g in grads.items():
            i = self.where.get(n)
            if g is None or i is None:
                continue
            if n in self._seen:
                raise ValueError(f"gradient {n!r} was handed to the gradient store twice in one backward")
            self._seen.add(n)
            b = self.buckets[i]
            off, k, shape = b.slots[n]
            dst = b.flat[off:off + k]
            if g.numel() != k:
                raise ValueError(f"gradient {n!r} has {g.numel()} elements, its parameter {k}")
            if self._micro > 0:
                dst.add_(g.reshape(-1).to(self.dtype))
            elif g.data_ptr() != dst.data_ptr() or g.dtype != self.dtype or not g.is_contiguous():
                dst.copy_(g.reshape(-1))
            b.ready += 1
        # collectives must be issued in the SAME order on every rank: buckets go out strictly in layout order (a rank whose
        # batch lacks a modality completes some bucket only in finish(); it then simply overlaps less, it never reorders)
        while self._sync and self._next < len(self.buckets) and \
                self.buckets[self._next].ready == len(self.buckets[self._next].names):
            self._launch(self.buckets[self._next])
            self._next += 1

    # ---- exchange -----------------------------------------------------------------------------------------------
    def _launch(self, b: _Bucket):
        if b.launched:
            return
        b.launched = True
        self.launches += 1
        if self.world == 1:
            return
        nb = b.flat.numel() * b.flat.element_size()
        if self.mode == "allreduce":
            op = dist.ReduceOp.AVG if _is_nccl(self.group) else dist.ReduceOp.SUM
            b.work = [dist.all_reduce(b.flat, op=op, group=self.group, async_op=True)]
            self.bytes_exchanged += 2 * nb * (self.world - 1) // self.world
        else:
            lo, hi = b.shard
            op = dist.ReduceOp.AVG if _is_nccl(self.group) else dist.ReduceOp.SUM
            b.work = [dist.reduce_scatter_tensor(b.flat[lo:hi], b.flat, op=op, group=self.group, async_op=True)]
            self.bytes_exchanged += nb * (self.world - 1) // self.world
            if self.mode == "rs_ag":
                if not _is_nccl(self.group):          # RCCL runs a group's collectives in issue order on its stream; gloo's
                    b.work[0].wait()                  # worker threads do not - keep the pair ordered there (CPU tests only)
                b.work.append(dist.all_gather_into_tensor(b.flat, b.flat[lo:hi], group=self.group, async_op=True))
                self.bytes_exchanged += nb * (self.world - 1) // self.world

    def wait_bucket(self, b: _Bucket):
        """Block the current stream on bucket b's exchange and finish the average (SUM backends)."""
        if b.work is not None:
            for w in b.work:
                w.wait()
            b.work = None
            if not _is_nccl(self.group):
                lo, hi = b.shard if self.mode == "zero1" else (0, b.flat.numel())
                b.flat[lo:hi].div_(self.world)

    def finish(self) -> Dict[str, torch.Tensor]:
        """After the backward: zero-fill what no layer emitted (first micro-step only), send what is still unsent, wait.
        Returns {name: averaged gradient view} (mode zero1: only this rank's shard of each bucket is meaningful - use
        FlatAdamW).  With capture(sync=False) nothing is sent and the buckets keep accumulating."""
        for b in self.buckets:
            if b.ready != len(b.names) and self._micro == 0:
                for n in b.names:
                    if n not in self._seen:
                        off, k, _ = b.slots[n]
                        b.flat[off:off + k].zero_()
        if not self._sync:
            self._micro += 1
            return {}
        for b in self.buckets:
            self._launch(b)
        for b in self.buckets:
            self.wait_bucket(b)
            b.launched = False
        self._micro = 0
        return {n: self.view(n) for n in self.where}

    def finish_into(self, named_params: Iterable[Tuple[str, torch.nn.Parameter]]) -> None:
        """Hand in what the capture did not see (gradients autograd left in ``p.grad`` for parameters outside the
        engines), exchange, and install the bucket views as ``p.grad``."""
        named = [(n, p) for n, p in named_params if n in self.where]
        extra = {n: p.grad for n, p in named if n not in self._seen and p.grad is not None
                 and p.grad.data_ptr() != self.view(n).data_ptr()}
        if extra:
            self.add(extra)
            for n, p in named:
                if n in extra:
                    self._autograd_owned[n] = (p, None)
                    p.grad = None             # its value now lives in the bucket: the next micro-step's autograd starts fresh
        out = self.finish()
        if not self._sync:
            return
        for n, p in named:
            if n in self._autograd_owned:
                # never a bucket VIEW for these: autograd's in-place accumulation of the next backward would write into the
                # bucket behind the store's back (and the slot would look "never emitted" and be zeroed)
                p.grad = out[n].clone()
                self._autograd_owned[n] = (p, p.grad)
            else:
                p.grad = out[n]

    def grad_norm_sq(self, out: Optional[torch.Tensor] = None, *, sumsq_fn: Optional[Callable] = None) -> torch.Tensor:
        """Squared global L2 norm of the exchanged (averaged) gradients as an fp32 device scalar - identical on every rank.
        mode zero1: each rank sums its own shard (the only part of a bucket it holds reduced) and the W partial sums are
        all-reduced; other modes: every rank holds every averaged bucket and sums them all, no communication.  Alignment
        pad elements are zero.  `sumsq_fn(x, out, accumulate=)`: the HIP pass (kernels.sumsq) unless a test injects its own."""
        if sumsq_fn is None:
            from . import kernels as K
            sumsq_fn = K.sumsq
        if out is None:
            out = torch.zeros(1, dtype=torch.float32, device=self.device)
        sharded = self.mode == "zero1" and self.world > 1
        for i, b in enumerate(self.buckets):
            lo, hi = b.shard if sharded else (0, b.flat.numel())
            sumsq_fn(b.flat[lo:hi], out, accumulate=i > 0)
        if sharded:
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
        return out


# =====================================================================================================================
# AdamW on the flat buckets (fp32 master + moments; optionally ZeRO-1 sharded)
# =====================================================================================================================
class FlatAdamW:
    """AdamW (decoupled weight decay; libra_pretrain.yaml:83-91: lr 1e-4, betas (0.9, 0.99), wd 0.01) over GradBuckets.

    * fp32 master weights and moments (what DeepSpeed's bf16 mode keeps; 12 B/param) for the rank's shard only when
      `shard=True` (buckets.mode == "zero1": 16 B/param → 16/W B/param of optimizer state, 22 GB instead of 176 GB per rank
      for the 11 B full finetune on 8 GPUs);
    * the bf16 parameters are re-bound (`p.data`) to views of flat parameter buckets with the gradient buckets' layout, so
      the update kernel writes the new bf16 value in place and the all-gather lands directly in the parameters;
    * one pass over HBM per element: `kernels.adamw_step` reads grad (2 B) + master, m, v (12 B), writes master, m, v and
      the bf16 parameter (14 B).  Segments of a shard with different weight decay (norm weights / biases: no decay, as HF
      Trainer's `get_decay_parameter_names`) are separate launches.
    """

    def __init__(self, buckets: GradBuckets, named_params: Iterable[Tuple[str, torch.nn.Parameter]], *, lr: float = 1e-4,
                 betas=(0.9, 0.99), eps: float = 1e-8, weight_decay: float = 0.01,
                 no_decay: Callable[[str, torch.nn.Parameter], bool] = lambda n, p: p.ndim < 2,
                 max_grad_norm: Optional[float] = None, update_fn: Optional[Callable] = None,
                 sumsq_fn: Optional[Callable] = None):
        self.buckets = buckets
        self.shard = buckets.mode == "zero1" and buckets.world > 1
        if buckets.world > 1 and buckets.mode == "zero1" and not self.shard:
            raise AssertionError
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        # global-norm clipping of the exchanged gradients (libra_pretrain.yaml `max_grad_norm: 1.0`, ZeRO-2.json
        # "gradient_clipping": "auto"): one 2 B/element norm pass over the buckets, the coefficient is applied inside the update
        self.max_grad_norm = max_grad_norm if max_grad_norm and max_grad_norm > 0 else None
        self.sumsq_fn = sumsq_fn
        self.last_grad_norm_sq: Optional[torch.Tensor] = None
        self._params = None
        self.t = 0
        if update_fn is None:
            from . import kernels as K                       # HIP only: raises on CPU tensors
            update_fn = K.adamw_step
        self.update_fn = update_fn
        params = dict(named_params)
        self._params = params
        missing = [n for n in buckets.where if n not in params]
        if missing:
            raise ValueError(f"FlatAdamW: parameters missing for bucket entries {missing[:3]}...")
        self.pflat: List[torch.Tensor] = []
        self.state: List[dict] = []
        with torch.no_grad():
            for b in buckets.buckets:
                pf = torch.zeros_like(b.flat)
                for n in b.names:
                    off, k, shape = b.slots[n]
                    p = params[n]
                    pf[off:off + k].copy_(p.detach().reshape(-1))
                    p.data = pf[off:off + k].view(shape)     # (packed operand copies are refreshed every forward: safe)
                    p._libra_flat_owner = id(self)           # PackedOperands.adopt() must not re-point it to a fused operand
                lo, hi = b.shard if self.shard else (0, b.flat.numel())
                master = pf[lo:hi].float()
                segs = []                                    # (start, end, weight decay) inside [lo, hi), param-aligned
                for n in b.names:
                    off, k, _ = b.slots[n]
                    s, e = max(off, lo), min(off + k, hi)
                    if s < e:
                        segs.append((s - lo, e - lo, 0.0 if no_decay(n, params[n]) else weight_decay))
                merged = []
                for s, e, wd in segs:                        # merge neighbours with equal decay across the alignment pads
                    if merged and merged[-1][2] == wd and s - merged[-1][1] < ALIGN:
                        merged[-1] = (merged[-1][0], e, wd)
                    else:
                        merged.append((s, e, wd))
                self.pflat.append(pf)
                self.state.append(dict(master=master, m=torch.zeros_like(master), v=torch.zeros_like(master), segs=merged,
                                       lo=lo, hi=hi))
        self.state_bytes = sum(3 * s["master"].numel() * 4 for s in self.state)

    @torch.no_grad()
    def step(self, lr: Optional[float] = None):
        """Update from the buckets' current (exchanged) gradients.  Call after `buckets.finish()`; with mode zero1 the
        per-bucket all-gathers of the new parameters overlap the next bucket's update."""
        dev = self.buckets.device
        if torch.device(dev).type == "cuda":
            # the device error word is polled asynchronously by the backward; raise here only if a poll has ALREADY landed - the
            # optimizer never blocks the host (no kernel of the current library writes the word: a blocking wait here only
            # removed CPU run-ahead and would fail under graph capture)
            from . import kernels as K
            K.errors.check(dev, wait=False)
        self.t += 1
        lr = self.lr if lr is None else lr
        b1, b2 = self.betas
        bc1, bc2 = 1.0 - b1 ** self.t, 1.0 - b2 ** self.t
        works = []
        clip = {}
        if self.max_grad_norm is not None:
            self.last_grad_norm_sq = self.buckets.grad_norm_sq(sumsq_fn=self.sumsq_fn)
            clip = dict(grad_norm_sq=self.last_grad_norm_sq, max_grad_norm=self.max_grad_norm)
        for b, pf, st in zip(self.buckets.buckets, self.pflat, self.state):
            lo = st["lo"]
            for s, e, wd in st["segs"]:
                self.update_fn(st["master"][s:e], st["m"][s:e], st["v"][s:e], b.flat[lo + s:lo + e], pf[lo + s:lo + e],
                               lr=lr, beta1=b1, beta2=b2, eps=self.eps, weight_decay=wd, bias_corr1=bc1, bias_corr2=bc2, **clip)
            if self.shard:
                works.append(dist.all_gather_into_tensor(pf, pf[lo:st["hi"]], group=self.buckets.group, async_op=True))
        for w in works:
            w.wait()

    def zero_grad(self, set_to_none: bool = True):
        """Engine-owned gradients live in the buckets and are overwritten by the next backward; gradients autograd owns
        (parameters outside the engines) are dropped so that the next backward does not accumulate onto an exchanged value."""
        for p in (self._params or {}).values():
            p.grad = None

    # ---- checkpoint / resume (HF Trainer `save_strategy`, DeepSpeed checkpoint of the fp32 master + moments) ----------------
    def state_dict(self) -> dict:
        """This rank's optimizer state: step count, and per bucket the fp32 master weights and both moments of the range
        [lo, hi) it owns (the whole bucket unless sharded) with the layout that range belongs to.  zero1: one file per rank,
        as DeepSpeed's `zero_pp_rank_*_optim_states`."""
        return {"t": self.t, "world": self.buckets.world, "rank": self.buckets.rank, "sharded": self.shard,
                "hyper": {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd,
                          "max_grad_norm": self.max_grad_norm},
                "buckets": [{"names": list(b.names), "numel": b.flat.numel(), "lo": st["lo"], "hi": st["hi"],
                             "master": st["master"].clone(), "m": st["m"].clone(), "v": st["v"].clone()}
                            for b, st in zip(self.buckets.buckets, self.state)]}

    @torch.no_grad()
    def load_state_dict(self, sd: dict, *, strict_hyper: bool = False) -> None:
        """Resume: restores step count, master weights and moments, and re-derives the bf16 parameters from the masters (so a
        resumed run continues from the fp32 values, not from bf16-rounded weights with zero moments).  A ZeRO-1 shard must come from the
        same world size / rank; un-sharded state loads on any rank of any world size (only the unpadded payload of a bucket is
        compared and copied: the tail padding to a multiple of world·ALIGN is the one thing that depends on the world size).  Hyper-parameters are the CONSTRUCTOR's (as torch.optim lets a resumed run change
        lr or clipping), but a difference from the saved ones is never silent: a warning, or ValueError with strict_hyper."""
        # the sharding MODE must match; world size and rank only matter when the state is a shard (un-sharded state is the same on
        # every rank: rank 0 saves, every rank loads, and a resume on another world size is legitimate)
        sharded = bool(sd.get("sharded", self.shard))
        if sharded != self.shard:
            raise ValueError(f"FlatAdamW.load_state_dict: saved sharded = {sharded!r}, this optimizer has {self.shard!r}")
        if sharded:
            for k, mine in (("world", self.buckets.world), ("rank", self.buckets.rank)):
                if k in sd and sd[k] != mine:
                    raise ValueError(f"FlatAdamW.load_state_dict: saved {k} = {sd[k]!r}, this optimizer has {mine!r} (a ZeRO-1 "
                                     "shard can only be resumed by the same rank of the same world size)")
        mine_h = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd,
                  "max_grad_norm": self.max_grad_norm}
        saved_h = sd.get("hyper") or {}
        diff = {k: (tuple(saved_h[k]) if isinstance(saved_h[k], (list, tuple)) else saved_h[k], v) for k, v in mine_h.items()
                if k in saved_h and (tuple(saved_h[k]) if isinstance(saved_h[k], (list, tuple)) else saved_h[k]) != v}
        if diff:
            msg = ("FlatAdamW.load_state_dict: hyper-parameters differ from the checkpoint's (saved, current): "
                   + ", ".join(f"{k}={a!r}->{b!r}" for k, (a, b) in diff.items()))
            if strict_hyper:
                raise ValueError(msg)
            import warnings
            warnings.warn(msg + "; continuing with the current ones", RuntimeWarning, stacklevel=2)
        if len(sd["buckets"]) != len(self.state):
            raise ValueError("FlatAdamW.load_state_dict: bucket count differs (different parameter set or bucket size)")
        for b, st, pf, src in zip(self.buckets.buckets, self.state, self.pflat, sd["buckets"]):
            if src["names"] != list(b.names):
                raise ValueError("FlatAdamW.load_state_dict: bucket membership differs from the saved one (same parameters and "
                                 "bucket_bytes are required)")
            if sharded:
                if src["numel"] != b.flat.numel() or (src["lo"], src["hi"]) != (st["lo"], st["hi"]):
                    raise ValueError("FlatAdamW.load_state_dict: shard range differs from the saved one (the bucket length is padded "
                                     "to a multiple of world_size * ALIGN: same world size and rank are required for a ZeRO-1 shard)")
                n = st["hi"] - st["lo"]
            else:
                # payload = end of the last slot; everything past it is world-size padding (zero gradient, never a parameter)
                off, k, _ = b.slots[b.names[-1]]
                n = off + (k + ALIGN - 1) // ALIGN * ALIGN
                if min(src["numel"], b.flat.numel()) < n or src["master"].numel() < n:
                    raise ValueError("FlatAdamW.load_state_dict: saved bucket is shorter than this layout's payload")
            for k in ("master", "m", "v"):
                st[k][:n].copy_(src[k][:n].to(st[k].device))
                st[k][n:].zero_()
            pf[st["lo"]:st["hi"]].copy_(st["master"])                     # bf16(master), in place: the parameters are views
        self.t = int(sd["t"])
        if self.shard:
            for pf, st in zip(self.pflat, self.state):
                dist.all_gather_into_tensor(pf, pf[st["lo"]:st["hi"]], group=self.buckets.group)
