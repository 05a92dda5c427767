"""world_size-2 gloo test of the bucketed gradient all-reduce used by the data-parallel path."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libra_amd.dp import BucketedGradReducer
    g = torch.Generator().manual_seed(100 + rank)
    grads_a = {f"l1.p{i}": torch.randn(33 + i, 7, generator=g) for i in range(5)}
    grads_b = {f"l0.p{i}": torch.randn(1000, generator=g) for i in range(3)}
    red = BucketedGradReducer(bucket_bytes=4096)
    red.add(grads_a)          # "layer 1" grads arrive first (backward order)
    red.add(grads_b)
    out = red.finish()
    # numpy payloads are pickled by value (torch tensors would travel as shared-memory handles that die with the child)
    q.put((rank, {k: v.numpy().copy() for k, v in out.items()},
           {k: v.numpy().copy() for k, v in {**grads_a, **grads_b}.items()}, red.bytes_reduced))
    dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in range(2)]
    [p.join(60) for p in ps]
    res.sort(key=lambda t: t[0])
    (_, out0, loc0, nbytes), (_, out1, loc1, _) = res
    assert set(out0) == set(loc0)
    import numpy as np
    for k in out0:
        mean = (loc0[k] + loc1[k]) / 2
        assert np.allclose(out0[k], mean, atol=1e-6), k
        assert np.array_equal(out0[k], out1[k]), k
        assert out0[k].shape == loc0[k].shape
    assert nbytes == sum(v.size * 4 for v in loc0.values())


def test_single_process_passthrough():
    from libra_amd.dp import BucketedGradReducer
    red = BucketedGradReducer(bucket_bytes=64)
    g = {"a": torch.arange(10.0), "b": torch.ones(3, 3)}
    red.add(g)
    out = red.finish()
    assert torch.equal(out["a"], g["a"]) and torch.equal(out["b"], g["b"])
