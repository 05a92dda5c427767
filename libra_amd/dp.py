"""Data-parallel gradient exchange for the hot path: one process per GPU, gradients summed
with RCCL all-reduce over xGMI in flat buckets, each launched asynchronously the moment it fills.

Overlap with backward: the two engines (`vit_engine.backward`, `decoder_engine.backward`) hand every layer's
parameter gradients to the reducer that is capturing (`with reducer.capture(): loss.backward()`) as soon as
the layer's last weight-gradient GEMM is enqueued, so the exchange of layer i runs on RCCL's own stream under
the backward kernels of layers < i and only the last bucket is exposed.  Without a capturing reducer the
engines' `emit` calls are no-ops.  Works unchanged on the gloo backend (CPU tensors) for the world_size-2 tests.
"""
from __future__ import annotations

import contextlib
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist

# The capturing reducer.  A plain module global, not a thread-local: autograd runs backward on its own worker
# thread (SURVEY §8b "threading conventions"), and one rank drives one backward at a time.
_capturing: Optional["BucketedGradReducer"] = None


def emit(grads: Dict[str, torch.Tensor]) -> None:
    """Called by the engines after each layer: {state-dict name: finished gradient}."""
    if _capturing is not None:
        _capturing.add(grads)


def emit_new(g: Dict[str, torch.Tensor], seen: set) -> None:
    """emit() the entries of the running gradient dict `g` that were added since the last call."""
    if _capturing is None:
        return
    fresh = {n: t for n, t in g.items() if n not in seen}
    seen.update(fresh)
    _capturing.add(fresh)


class BucketedGradReducer:
    """Collects parameter gradients into flat buckets of ~bucket_bytes and launches one asynchronous
    all-reduce (SUM) per full bucket; ``finish()`` waits, divides by world size in place and returns views
    of the buckets under the names the gradients were added with."""

    def __init__(self, bucket_bytes: int = 64 << 20, group=None, only: Optional[Iterable[str]] = None):
        self.bucket_bytes = bucket_bytes
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.only = None if only is None else set(only)      # names to exchange (e.g. the trainable ones); None = all
        self._pending: List = []           # (work, flat, [(name, shape, numel)])
        self._cur: List[Tuple[str, torch.Tensor]] = []
        self._cur_bytes = 0
        self.seen: set = set()
        self.bytes_reduced = 0
        self.launches = 0

    @contextlib.contextmanager
    def capture(self):
        """Route the engines' per-layer emissions of the enclosed backward into this reducer."""
        global _capturing
        if _capturing is not None:
            raise RuntimeError("another BucketedGradReducer is already capturing")
        _capturing = self
        try:
            yield self
        finally:
            _capturing = None

    def add(self, grads: Dict[str, torch.Tensor]):
        for n, g in grads.items():
            if g is None or (self.only is not None and n not in self.only):
                continue
            if n in self.seen:
                raise ValueError(f"gradient {n!r} was handed to the reducer twice in one step")
            self.seen.add(n)
            self._cur.append((n, g))
            self._cur_bytes += g.numel() * g.element_size()
            if self._cur_bytes >= self.bucket_bytes:
                self._flush()

    def _flush(self):
        if not self._cur:
            return
        items, self._cur, self._cur_bytes = self._cur, [], 0
        dt = items[0][1].dtype
        flat = torch.cat([g.reshape(-1).to(dt) for _, g in items])
        self.bytes_reduced += flat.numel() * flat.element_size()
        self.launches += 1
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if self.world > 1 else None
        self._pending.append((work, flat, [(n, g.shape, g.numel(), g.dtype) for n, g in items]))

    def finish(self) -> Dict[str, torch.Tensor]:
        self._flush()
        out: Dict[str, torch.Tensor] = {}
        for work, flat, items in self._pending:
            if work is not None:
                work.wait()
            if self.world > 1 and flat.dtype.is_floating_point:
                flat.div_(self.world)
            off = 0
            for n, shape, k, dt in items:
                out[n] = flat[off:off + k].view(shape).to(dt)        # (.to is a no-op view for the common single-dtype bucket)
                off += k
        self._pending = []
        self.seen = set()
        return out

    def finish_into(self, named_params: Iterable[Tuple[str, torch.nn.Parameter]]) -> None:
        """Exchange whatever the capture did not see (parameters outside the engines), wait, and install the averaged
        gradients as ``p.grad``.  Not for gradient accumulation: captured gradients are this backward's only."""
        named = [(n, p) for n, p in named_params if p.grad is not None]
        self.add({n: p.grad for n, p in named if n not in self.seen})
        out = self.finish()
        for n, p in named:
            g = out.get(n)
            if g is not None:
                p.grad = g.view(p.shape) if g.shape != p.shape else g
