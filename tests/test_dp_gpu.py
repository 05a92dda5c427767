"""Two ranks, real engine, real overlap hook: each rank runs the HIP ViT backward on its own input under
`reducer.capture()`; the installed gradients must be the mean of the two ranks' plain-backward gradients.
Both ranks share cuda:0 (the GPU box has one device), so the exchange runs over gloo; the code path is the one
bench.py --gpus N drives over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import load_golden, sub
    from test_model_gpu import _build_clip
    from libra_amd.dp import BucketedGradReducer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t, meta = load_golden("vit_tiny.safetensors")
    m = _build_clip(meta, sub(t, "w."))
    m.requires_grad_(True)
    ct = t["in.cotangent"].cuda()
    xs = [(t["in.pixel_values"] * (1.0 + 0.5 * r) + 0.1 * r).to(torch.bfloat16).cuda() for r in range(world)]

    def loss(x):
        hs = m(x, output_hidden_states=True).hidden_states
        return (torch.cat([hs[-2], hs[-3]], -1)[:, 1:].float() * ct).sum()

    named = list(m.named_parameters())
    plain = []
    for r in range(world):                         # every rank can compute both ranks' local gradients: the expectation
        m.zero_grad(set_to_none=True)
        loss(xs[r]).backward()
        plain.append({n: p.grad.float().clone() for n, p in named if p.grad is not None})
    m.zero_grad(set_to_none=True)
    red = BucketedGradReducer(bucket_bytes=1 << 15)
    with red.capture():
        loss(xs[rank]).backward()
    red.finish_into(named)
    worst = 0.0
    for n, p in named:
        if n not in plain[0]:
            continue
        want = sum(pl[n] for pl in plain) / world
        err = float((p.grad.float() - want).abs().max()) / max(float(want.abs().max()), 1e-6)
        worst = max(worst, err)
    torch.cuda.synchronize()
    q.put((rank, worst, len(plain[0]), red.bytes_reduced))
    dist.destroy_process_group()


def test_two_rank_overlapped_exchange_on_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = [q.get(timeout=600) for _ in range(2)]
    [p.join(120) for p in ps]
    for rank, worst, n, nbytes in res:
        assert n >= 37
        assert worst < 1e-2, (rank, worst)         # bf16 sum of two bf16 gradients, then /2: one rounding
        assert nbytes > 0
