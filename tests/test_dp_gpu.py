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


def _worker(rank, world, port, q, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import load_golden, sub
    from test_model_gpu import _build_clip
    from libra_amd import dp, vit_engine
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
    L = meta["cfg"]["num_hidden_layers"]
    red = dp.GradBuckets(named, bucket_bytes=1 << 15, group_fn=lambda n: vit_engine.emit_group(n, L), mode=mode)
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
    q.put((rank, worst, len(plain[0]), red.bytes_exchanged))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allreduce", "rs_ag"])
def test_two_rank_overlapped_exchange_on_one_gpu(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    [p.start() for p in ps]
    res = [q.get(timeout=600) for _ in range(2)]
    [p.join(120) for p in ps]
    for rank, worst, n, nbytes in res:
        assert n >= 37
        assert worst < 1e-2, (rank, worst)         # bf16 sum of two bf16 gradients, then /2: one rounding
        assert nbytes > 0


def _worker_decoder(rank, world, port, q, mode):
    """The routed DECODER (tiny config, frozen language = pretraining) under the 2-rank exchange; rank 1's micro-batch is
    text-only (no vision token): every rank must still run the same collectives and the result is the mean over ranks."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import load_golden, sub, torch_adamw_update
    from libra_amd import decoder_engine as DE
    from libra_amd import dp
    from libra_amd.libra import LibraConfig, LibraForCausalLM, apply_freeze_policy
    BF = torch.bfloat16
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t, meta = load_golden("libra_tiny.safetensors")
    c = meta["cfg"]
    m = LibraForCausalLM(LibraConfig(**c))
    m.load_state_dict(sub(t, "w."), strict=True)
    m = m.to(BF).cuda()
    apply_freeze_policy(m, frozen_language=True)
    g = torch.Generator().manual_seed(0)
    ids_t = torch.randint(3, c["vocab_size"], (1, 2, 16), generator=g).repeat(2, 1, 1)
    ids_t[:, :, 0] = 1
    lab_t = ids_t.clone(); lab_t[:, :, 0] = -100
    batches = [dict(input_ids=t["in.input_ids"].cuda(), attention_mask=t["in.attention_mask"].cuda(),
                    vision_indices=t["in.vision_indices"].cuda(), contiguous_signal=t["in.signal"].to(BF).cuda(),
                    labels=t["in.labels"].cuda()),
               dict(input_ids=ids_t.cuda(), attention_mask=torch.ones(2, 16, dtype=torch.long).cuda(),
                    vision_indices=torch.full((2, 16), c["max_vision_token_length"], dtype=torch.long).cuda(),
                    contiguous_signal=None, labels=lab_t.cuda())]
    named = [(n, p) for n, p in m.named_parameters() if p.requires_grad and n != "vision_hidden_placeholder"]
    plain = []
    for r in range(world):
        m.zero_grad(set_to_none=True)
        m(**batches[r]).loss.backward()
        plain.append({n: p.grad.float().clone() for n, p in named})
    m.zero_grad(set_to_none=True)
    L = c["num_hidden_layers"]
    st = dp.GradBuckets(named, bucket_bytes=1 << 15, group_fn=lambda n: DE.emit_group(n, L), mode=mode)
    with st.capture():
        m(**batches[rank]).loss.backward()
    worst = 0.0
    if mode != "zero1":
        st.finish_into(named)
        for n, p in named:
            want = sum(pl[n] for pl in plain) / world
            worst = max(worst, float((p.grad.float() - want).abs().max()) / max(float(want.abs().max()), 1e-6))
    else:
        # ZeRO-1: reduce-scatter -> HIP fused AdamW on this rank's shard -> all-gather of the bf16 parameters, against an
        # unsharded fp32-master AdamW (torch statement of the kernel) on the mean gradient
        before = {n: p.detach().float().clone() for n, p in named}
        opt = dp.FlatAdamW(st, named, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.1)
        st.finish()
        opt.step()
        for n, p in named:
            gm = (sum(pl[n] for pl in plain) / world).to(BF)
            master, mm, vv = before[n].clone(), torch.zeros_like(before[n]), torch.zeros_like(before[n])
            out = torch.empty_like(master, dtype=BF)
            torch_adamw_update(master, mm, vv, gm, out, lr=1e-2, beta1=0.9, beta2=0.99, eps=1e-8,
                               weight_decay=0.0 if p.ndim < 2 else 0.1, bias_corr1=1 - 0.9, bias_corr2=1 - 0.99)
            # first Adam step = lr * sign(g) wherever |g| >> eps: compare away from the sign flips of near-zero means
            ok = gm.float().abs() > 1e-6
            d = (p.detach().float() - out.float()).abs()[ok]
            worst = max(worst, float(d.max()) / max(float(out.float().abs().max()), 1e-6) if d.numel() else 0.0)
    torch.cuda.synchronize()
    q.put((rank, worst, st.bytes_exchanged, st.launches))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allreduce", "rs_ag", "zero1"])
def test_two_rank_decoder_exchange_with_a_text_only_rank(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_decoder, args=(r, 2, port, q, mode)) for r in range(2)]
    [p.start() for p in ps]
    res = [q.get(timeout=600) for _ in range(2)]
    [p.join(120) for p in ps]
    for rank, worst, nbytes, launches in res:
        assert worst < 1e-2, (mode, rank, worst)
        assert nbytes > 0 and launches >= 2


def test_cu_reserved_stream_runs_the_kernels_and_restores_the_budget():
    """The CU-budget knob of the data-parallel layer (include/libra_hip.h "compute-unit budget"): a GEMM and the persistent
    bridge-attention kernels launched on a CU-masked stream (32 CUs left to RCCL, persistent grids sized to the rest) give the
    bit-identical results of the default stream; the budget is restored when the stream is closed; the CU map diagnostic sees
    fewer distinct CUs under the mask."""
    import ctypes
    from libra_amd import _lib, kernels as K
    g = torch.Generator(device="cuda").manual_seed(3)
    rnd = lambda *s: torch.randn(*s, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    a, b = rnd(5000, 512), rnd(7000, 512)                      # 560 tiles of 256^2: the persistent GEMM path
    B, S, H = 2, 512, 160                                      # 640 attention items
    q, ks, kc, vs, vc = [rnd(B * S, H * 128) for _ in range(5)]
    flag = torch.zeros(B * S, dtype=torch.uint8, device="cuda"); flag[100:300] = 1
    lens = torch.full((B,), S, dtype=torch.int32, device="cuda")

    do = rnd(B * S, H * 128)

    def run():
        c = K.gemm_nt(a, b)
        o, lse = K.bridge_attn_fwd(q, ks, kc, vs, vc, flag, lens, B, S, H, 128 ** -0.5, need_lse=True)
        grads = K.bridge_attn_bwd(q, ks, kc, vs, vc, o, do, flag, lens, lse, B, S, H, 128 ** -0.5)     # persistent dQ and dK/dV passes
        return (c, o, lse) + tuple(grads)
    ref = run()
    total = K.cu_count()
    assert K.set_cu_budget(0) == 0                             # default: no budget
    rs = K.ReservedCUStream(32)
    assert rs.cus == total - 32 and rs.persistent_cus == total - 32
    with rs:
        got = run()
        out = torch.zeros(2 * 512, dtype=torch.int32, device="cuda")
        _lib.check(_lib.lib().libra_debug_cu_map(out.data_ptr(), 512, 20, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "cu_map")
    torch.cuda.synchronize()
    rs.close()
    assert K.set_cu_budget(0) == 0                             # restored by close()
    for x, y in zip(ref, got):
        assert torch.equal(x, y)
    w = out.cpu().numpy().astype("int64") & 0xffffffff
    seen = {(int(w[2 * i + 1]) & 15, (int(w[2 * i]) >> 8) & 0xff) for i in range(512)}
    assert 1 <= len(seen) <= total - 32, len(seen)
