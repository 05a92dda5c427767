"""BASELINE configs[2] and configs[3] as they are benchmarked: the pretraining step at S=700 (578 vision + <=122 text tokens per
sequence, bs 8, frozen language stream, GradBuckets + fused AdamW with global-norm clipping; libra_pretrain.yaml:17-19,83-96) and
the headline B=8 x S=2048 x 32-layer step bench.py times - correctness checks on the steps themselves.
(The 8-rank RCCL halves of configs[3]/[4] need an 8-GPU node: the N>1 code path is covered over gloo in test_dp_gloo.py /
test_dp_gpu.py and by the self-launching bench test below.)"""
import json
import math
import os
import subprocess
import sys

import pytest
import torch

from helpers import load_golden, parity_report, rel_err, sub, torch_adamw_update

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _pretrain_batch(c, B, S, L, boi, eoi, g, lens):
    """configs[3]-shaped inputs: BOS | one image (L vision tokens) | caption text, right-padded to S."""
    V, Q = c["vocab_size"], c["vision_codebook_num"]
    ids = torch.zeros(Q, B, S, dtype=torch.long)
    am = torch.zeros(B, S, dtype=torch.long)
    vi = torch.full((B, S), L, dtype=torch.long)
    sig = torch.zeros(B, S, c["contiguous_signal_size"])
    for b in range(B):
        n = int(lens[b])
        text = torch.randint(3, V - 2, (n - 1 - L,), generator=g)
        for q in range(Q):
            img = torch.cat([torch.tensor([boi]), V + torch.randint(0, boi - V, (L - 2,), generator=g), torch.tensor([eoi])])
            ids[q, b, :n] = torch.cat([torch.tensor([1]), img, text])
        am[b, :n] = 1
        vi[b, 1:1 + L] = torch.arange(L)
        sig[b, 2:L] = torch.randn(L - 2, c["contiguous_signal_size"], generator=g)
    return ids, am, vi, sig


def test_configs3_pretrain_step_tiny_width_vs_oracle():
    """S=700, B=8, 578 vision + 26..122 text tokens, frozen language, GradBuckets + FlatAdamW(max_grad_norm=1): loss and every
    trainable gradient against autograd through the fp32 oracle; the clipped AdamW update against torch's arithmetic on the same
    bf16 gradients."""
    from libra_amd import decoder_engine as DE
    from libra_amd import dp
    from libra_amd.libra import LibraConfig, LibraForCausalLM, apply_freeze_policy
    from oracle import libra_oracle as LO
    t, meta = load_golden("libra_tiny.safetensors")
    L, res, S, B = 578, 24, 700, 8
    c = dict(meta["cfg"], max_vision_token_length=L, image_feature_resolution=res, max_position_embeddings=2048)
    m = LibraForCausalLM(LibraConfig(**c))
    m.load_state_dict(sub(t, "w."), strict=True)
    m = m.to(BF).cuda()
    apply_freeze_policy(m, frozen_language=True)
    m.train()
    g = torch.Generator().manual_seed(7)
    lens = torch.tensor([700, 605, 650, 700, 612, 690, 700, 633])
    ids, am, vi, sig = _pretrain_batch(c, B, S, L, meta["boi"], meta["eoi"], g, lens)
    labels = LO.get_labels(ids, am, [[(1 + L, 2 + L)] for _ in range(B)], boi_token_id=meta["boi"], bos_token_id=1)
    named = [(n, p) for n, p in m.named_parameters() if p.requires_grad and n != "vision_hidden_placeholder"]
    assert named and all("vision" in n for n, _ in named)
    nl = c["num_hidden_layers"]
    st = dp.GradBuckets(named, bucket_bytes=1 << 16, group_fn=lambda n: DE.emit_group(n, nl))
    opt = dp.FlatAdamW(st, named, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01, max_grad_norm=1.0)
    before = {n: p.detach().float().clone() for n, p in named}
    kw = dict(input_ids=ids.cuda(), attention_mask=am.cuda(), vision_indices=vi.cuda(), contiguous_signal=sig.to(BF).cuda(),
              labels=labels.cuda())
    with st.capture():
        out = m(**kw)
        out.loss.backward()
    st.finish_into(named)
    grads = {n: p.grad.detach().clone() for n, p in named}
    # ---- oracle (fp32 arithmetic on the same bf16-rounded weights)
    sdf = {k: v.to(BF).float().requires_grad_(k in dict(named)) for k, v in sub(t, "w.").items()}
    kwo = dict(layers=nl, heads=c["num_attention_heads"], vocab=c["vocab_size"], max_vision_token_length=L, eps=c["rms_norm_eps"],
               max_pos=2048)
    hid, flag = LO.model_forward(sdf, ids, am, vi, sig.to(BF).float(), **kwo)
    ref_loss = LO.causal_lm_loss(LO.vl_logits(sdf, hid, flag, 2), labels)
    ref_loss.backward()
    assert abs(float(out.loss) - float(ref_loss)) < 2e-2 * abs(float(ref_loss)), (float(out.loss), float(ref_loss))
    worst = ("", 0.0)
    for n, _ in named:
        e = rel_err(grads[n].float().cpu(), sdf[n].grad)
        worst = max(worst, (n, e), key=lambda x: x[1])
        assert e < 5e-2, (n, e)
    # ---- the optimizer step: global norm over the buckets == norm of the gradients; update == torch AdamW arithmetic
    opt.step()
    gn = math.sqrt(sum(float(v.double().pow(2).sum()) for v in grads.values()))
    assert abs(math.sqrt(float(opt.last_grad_norm_sq)) - gn) < 1e-3 * gn
    coef = min(1.0, 1.0 / (gn + 1e-6))
    for n, p in named:
        master, mm, vv = before[n].cuda(), torch.zeros_like(before[n]).cuda(), torch.zeros_like(before[n]).cuda()
        outp = torch.empty_like(master, dtype=BF)
        torch_adamw_update(master, mm, vv, grads[n], outp, lr=1e-3, beta1=0.9, beta2=0.99, eps=1e-8,
                           weight_decay=0.0 if p.ndim < 2 else 0.01, bias_corr1=1 - 0.9, bias_corr2=1 - 0.99, grad_scale=coef)
        # element-wise within 2 bf16 ulps of the expected NEW value (ulp(x) = 2^-7 |x| at worst): an lr = 1e-3 Adam step moves an
        # O(0.02) weight by ~1e-3 = several ulps, so a parameter the optimizer never touched fails here - and explicitly below
        err = (p.detach().float() - outp.float()).abs()
        tol = outp.float().abs() * 2 ** -6 + 1e-9
        assert bool((err <= tol).all()), (n, float((err / tol).max()))
        assert not torch.equal(p.detach().float(), before[n].to(p.device)), f"{n}: unchanged by opt.step()"
        assert p.data_ptr() != 0 and any(pf.data_ptr() <= p.data_ptr() < pf.data_ptr() + pf.numel() * pf.element_size()
                                         for pf in opt.pflat), f"{n}: not a view of the optimizer's flat parameter buckets"
    parity_report(f"[configs[3] shape, tiny width] S=700 B=8 (578 vision + 26..122 text tokens), frozen language, buckets + clipped AdamW: "
                  f"loss {float(out.loss):.4f} vs oracle {float(ref_loss):.4f}; worst trainable gradient {worst[1]:.2e} ({worst[0]}); "
                  f"grad norm {gn:.3f} (clip coef {coef:.3f})")


def _grads(w):
    return {n: p.grad.detach().clone() for n, p in w.named}


def test_configs3_full_width_two_layers_properties():
    """The same step at FULL width (H=4096, I=11008, 32 heads, V=32000) with 2 decoder layers, exactly as bench.py builds it
    (`--seq 700 --with-optimizer --recompute`, libra_pretrain.yaml:19,94-96): finite loss near ln V, a bit-identical rerun, and
    gradient checkpointing == saved activations bit for bit."""
    import bench
    dev = torch.device("cuda", 0)
    w = bench.make_bridge(dev, 8, 700, 1, "allreduce", layers=2)
    loss1 = w.step(); g1 = _grads(w)
    loss2 = w.step(); g2 = _grads(w)
    assert torch.isfinite(loss1) and 4.0 < float(loss1) < 14.0, float(loss1)
    assert float(loss1) == float(loss2)
    assert all(torch.equal(g1[n], g2[n]) for n in g1)
    assert all(torch.isfinite(v.float()).all() for v in g1.values())
    w.model.gradient_checkpointing_enable()
    loss3 = w.step(); g3 = _grads(w)
    assert float(loss3) == float(loss1) and all(torch.equal(g1[n], g3[n]) for n in g1)
    nz = sum(int((v != 0).any()) for v in g1.values())
    assert nz > 0.95 * len(g1), (nz, len(g1))
    del w, g1, g2, g3
    torch.cuda.empty_cache()
    # the optimizer variant (buckets + clipped AdamW) of the same shape: two steps, the loss moves and stays finite
    wo = bench.make_bridge(dev, 8, 700, 1, "allreduce", with_optimizer=True, recompute=True, layers=2)
    la = float(wo.step()); lb = float(wo.step()); lc = float(wo.step())
    assert all(math.isfinite(x) for x in (la, lb, lc)) and la == float(loss1) and lb != la and lc < la, (la, lb, lc)
    assert math.isfinite(float(wo.opt.last_grad_norm_sq))
    parity_report(f"[configs[3] shape, full width x 2 layers] loss {la:.4f} -> {lb:.4f} -> {lc:.4f} over two clipped AdamW steps; rerun and "
                  f"gradient-checkpointing gradients bit-identical; grad norm {math.sqrt(float(wo.opt.last_grad_norm_sq)):.3f}")
    del wo
    torch.cuda.empty_cache()


def test_headline_step_full_size_loss_and_recompute_identity():
    """THE step bench.py times (configs[2]: B=8 x S=2048 x 32 layers, Libra-11B, frozen language): finite loss inside the band a
    random-init model must land in (text rows ln 32000 = 10.4, vision rows ln 514 = 6.2, mixed by the label counts), finite
    gradients, and gradients identical with and without gradient checkpointing (modeling_libra.py:787-797)."""
    import bench
    dev = torch.device("cuda", 0)
    w = bench.make_bridge(dev, 8, 2048, 1, "allreduce")
    chk = bench.step_check(w, w.named)
    assert chk["finite"] and 6.0 < chk["loss"] < 13.0, chk
    assert chk["grad_elements"] > 4.2e9 and chk["grad_norm"] > 0
    g1 = _grads(w)
    w.model.gradient_checkpointing_enable()
    chk2 = bench.step_check(w, w.named)
    assert chk2["loss"] == chk["loss"] and chk2["grad_norm"] == chk["grad_norm"], (chk, chk2)
    bad = [n for n, p in w.named if not torch.equal(p.grad, g1[n])]
    assert not bad, bad[:5]
    parity_report(f"[configs[2] headline step, full size] loss {chk['loss']:.5f}, gradient L2 norm {chk['grad_norm']:.5f} over "
                  f"{chk['grad_elements'] / 1e9:.2f} B elements, finite; gradient checkpointing bit-identical on all {len(g1)} tensors")
    del w, g1
    torch.cuda.empty_cache()


def test_bench_self_launches_two_ranks_over_gloo_on_one_gpu():
    """`python bench.py --gpus 2` WITHOUT torchrun starts its own two ranks (here sharing the box's one GPU over gloo) and reports
    n_gpus = 2 with the exchange diagnostics."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["LIBRA_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "1",
                        "--seq", "768", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["extra"]["dist_world_size"] == 2
    assert out["extra"]["backend"] == "gloo" and "exposed_comm_ms" in out["extra"] and out["step_check"]["finite"]


def _cfg4_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    runs = []
    for rep in range(2):                                       # the same step sequence twice from scratch: bit-identical or not
        w = bench.make_bridge(dev, 2, 4096, world, "zero1", with_optimizer=True, recompute=True, full_finetune=True, layers=2)
        losses = [float(w.step()) for _ in range(3)]
        norm = float(w.opt.last_grad_norm_sq)
        chk = 0.0
        nparam = 0
        for n, p in w.named:                                   # a checksum over every (all-gathered) bf16 parameter after the updates
            chk += float(p.detach().double().abs().sum())
            nparam += p.numel()
        state = sum(s["master"].numel() for s in w.opt.state)
        total = sum(b.flat.numel() for b in w.buckets.buckets)
        trainable_text = sum(p.numel() for n, p in w.named if "vision" not in n)
        runs.append((losses, norm, chk, nparam, state, total, trainable_text))
        del w
        torch.cuda.empty_cache()
    torch.cuda.synchronize()
    q.put((rank, runs))
    dist.destroy_process_group()


def test_configs4_full_width_two_layers_zero1_on_two_ranks():
    """BASELINE configs[4] (libra_instruction.yaml:64-67,82; deepspeed_configs/ZeRO-2.json:15-21) at FULL width with 2 decoder
    layers, as `bench.py --seq 4096 --batch 2 --full-finetune --with-optimizer --recompute --gpus 2` builds it: every parameter
    trainable (the full-width TEXT weight-gradient GEMMs run: [22016 x 4096] x K 8192 reduction-major operands ...), gradient
    checkpointing, ZeRO-1-style sharded AdamW with clipping over two ranks (sharing the box's one GPU over gloo).  Properties:
    finite losses in the random-init band, the loss moves under the optimizer, each rank holds half of the optimizer state, both
    ranks agree bit for bit, and a second run from scratch reproduces losses, gradient norm and parameter checksum exactly."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_cfg4_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=1500) for _ in range(2)], key=lambda t: t[0])
    [p.join(120) for p in ps]
    (_, r0), (_, r1) = res
    (l_a, n_a, c_a, nparam, state, total, text_elems), (l_b, n_b, c_b, *_rest) = r0
    assert all(math.isfinite(x) for x in l_a) and 4.0 < l_a[0] < 14.0, l_a
    assert l_a[1] != l_a[0] and l_a[2] < l_a[0], l_a                                  # the clipped AdamW steps move the loss down
    assert math.isfinite(n_a) and n_a > 0
    assert (l_a, n_a, c_a) == (l_b, n_b, c_b), "second run from scratch differs"      # deterministic step, exchange and update
    assert r1[0][0] == l_a and r1[0][2] == c_a, "ranks disagree"                      # identical parameters after the all-gather
    assert state == total // 2, (state, total)                                          # ZeRO-1: half of master / m / v per rank
    assert text_elems > 0.5 * nparam                                                    # the text stream IS trainable here (full finetune)
    parity_report(f"[configs[4] shape, full width x 2 layers, zero1 on 2 ranks] losses {l_a[0]:.4f} -> {l_a[1]:.4f} -> {l_a[2]:.4f}, "
                  f"gradient norm {math.sqrt(n_a):.3f}; rerun from scratch and both ranks bit-identical; {nparam / 1e9:.2f} B parameters "
                  f"all trainable, optimizer state {state / total:.2f} of the buckets per rank")
