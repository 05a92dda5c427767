"""Data-parallel gradient exchange for the hot path: one process per GPU, gradients summed
with RCCL all-reduce over xGMI in per-layer buckets as the backward produces them, on RCCL's own stream so
the exchange of layer i overlaps the backward kernels of layers < i.  Works unchanged on the gloo backend
(CPU tensors) for the world_size-2 tests.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.distributed as dist


class BucketedGradReducer:
    """Collects parameter gradients into flat bf16 buckets of ~bucket_bytes and launches one asynchronous
    all-reduce (SUM) per full bucket; ``finish()`` waits, divides by world size and scatters back."""

    def __init__(self, bucket_bytes: int = 64 << 20, group=None):
        self.bucket_bytes = bucket_bytes
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._pending: List = []           # (work, flat, [(name, view_shape, numel)])
        self._cur: List = []
        self._cur_bytes = 0
        self.bytes_reduced = 0

    def add(self, grads: Dict[str, torch.Tensor]):
        for n, g in grads.items():
            self._cur.append((n, g))
            self._cur_bytes += g.numel() * g.element_size()
            if self._cur_bytes >= self.bucket_bytes:
                self._flush()

    def _flush(self):
        if not self._cur:
            return
        items, self._cur, self._cur_bytes = self._cur, [], 0
        flat = torch.cat([g.reshape(-1) for _, g in items])
        self.bytes_reduced += flat.numel() * flat.element_size()
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if self.world > 1 else None
        self._pending.append((work, flat, items))

    def finish(self) -> Dict[str, torch.Tensor]:
        self._flush()
        out: Dict[str, torch.Tensor] = {}
        for work, flat, items in self._pending:
            if work is not None:
                work.wait()
            if self.world > 1:
                flat = flat / self.world if flat.dtype.is_floating_point else flat
            off = 0
            for n, g in items:
                k = g.numel()
                out[n] = flat[off:off + k].view(g.shape).to(g.dtype)
                off += k
        self._pending = []
        return out
