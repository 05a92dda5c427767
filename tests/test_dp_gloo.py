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


def _worker_capture(rank, world, port, q):
    """The overlap path: a fake 3-layer backward emits each layer's gradients through libra_amd.dp.emit_new (what the
    engines do), one parameter lives outside the engine (picked up by finish_into), one emitted name is frozen."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libra_amd import dp
    gen = torch.Generator().manual_seed(7 + rank)
    params = {f"layers.{i}.w": torch.nn.Parameter(torch.zeros(50, 9)) for i in range(3)}
    params["outside.b"] = torch.nn.Parameter(torch.zeros(11))
    local = {n: torch.randn(p.shape, generator=gen) for n, p in params.items()}
    red = dp.BucketedGradReducer(bucket_bytes=2048, only=set(params))
    launches = []
    with red.capture():
        g, seen = {}, set()
        for i in (2, 1, 0):                         # backward order
            g[f"layers.{i}.w"] = local[f"layers.{i}.w"]
            g[f"layers.{i}.frozen"] = torch.ones(3)  # not in `only`: never exchanged
            dp.emit_new(g, seen)
            launches.append(red.launches)
    dp.emit({"late": torch.ones(2)})                 # outside the capture: a no-op
    for n, p in params.items():
        p.grad = local[n].clone()                    # what autograd would have installed
    red.finish_into(params.items())
    q.put((rank, {n: p.grad.detach().numpy().copy() for n, p in params.items()},
           {n: v.numpy().copy() for n, v in local.items()}, launches))
    dist.destroy_process_group()


def test_capture_overlap_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_capture, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in range(2)]
    [p.join(60) for p in ps]
    res.sort(key=lambda t: t[0])
    (_, g0, l0, launches), (_, g1, l1, _) = res
    import numpy as np
    assert set(g0) == {"layers.0.w", "layers.1.w", "layers.2.w", "outside.b"}
    for k in g0:
        assert np.allclose(g0[k], (l0[k] + l1[k]) / 2, atol=1e-6), k
        assert np.array_equal(g0[k], g1[k]), k
    assert launches == [0, 1, 1] or launches[-1] >= 1      # 1800-B layers against 2048-B buckets: one left mid-backward


def test_reducer_rejects_double_add_and_nested_capture():
    import pytest
    from libra_amd.dp import BucketedGradReducer
    red = BucketedGradReducer()
    red.add({"a": torch.ones(2)})
    with pytest.raises(ValueError):
        red.add({"a": torch.ones(2)})
    with red.capture():
        with pytest.raises(RuntimeError):
            with BucketedGradReducer().capture():
                pass
    red.finish()
    red.add({"a": torch.ones(2)})                   # a new step may reuse the names


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
