#!/usr/bin/env python
"""bench.py — images/sec of Libra's vision hot path on MI355X (BASELINE.json configs[1]).

A "step" = one pass of the hot path over one synthetic batch resident in HBM:
    ViT-L/14@336 forward (all 24 layers, 25 hidden states)  ->  feature select [-2,-3]
    -> VQ encode (quant_conv GEMM + LFQ sign/pack -> token ids)
    -> backward of a fixed cotangent on the 2048-d feature through the ViT (dgrad + wgrad of every
       parameter that feeds it; bf16 grads), + RCCL gradient all-reduce when N > 1 (data parallel, weak scaling)
bs = 32 images / GPU, bf16, random-init weights, synthetic N(0,1) pixels.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel = the bf16 MFMA GEMM, per-launch
times from events on the launch stream in a separate instrumented step) and "cpu_baseline" (the CPU oracle
timed on this box's host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VIT_L = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336,
             patch_size=14)
GFLOP_FWD_PER_IMG = 381.9
GFLOP_LAYER = 15.884               # one encoder layer forward
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def build(device, batch, embed_dim=512):
    from transformers import CLIPVisionConfig
    from libra_amd.clip import CLIPVisionModel
    from libra_amd.libra import ImageTokenizer
    torch.manual_seed(42)
    clip = CLIPVisionModel(CLIPVisionConfig(**VIT_L))
    # the reference initialiser leaves biases at 0 and LN at (1,0); perturb so every term is numerically live
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in clip.named_parameters():
            if p.ndim == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
    clip = clip.to(torch.bfloat16).to(device)
    cfg = {"params": {"ddconfig": {"encoder_name": "clip_vit_l_336", "select_layer": [-2, -3]}, "embed_dim": embed_dim,
                      "codebook_size": 512, "num_codebook": 2}, "max_vision_token_length": 578}
    tok = ImageTokenizer(cfg, token_offset=32000, vision_model=clip)
    with torch.no_grad():
        tok.model.quant_conv.weight.normal_(0, 2048 ** -0.5, generator=None)
        tok.model.quant_conv.bias.normal_(0, 0.05)
    tok = tok.to(torch.bfloat16).to(device)
    # fwd/bwd config: the ViT's parameters take gradients (extension over the reference)
    clip.requires_grad_(True)
    tok.model.encoder.allow_grad = True
    g = torch.Generator().manual_seed(42)
    pixel = torch.randn(batch, 3, 336, 336, generator=g).to(torch.bfloat16).to(device)
    cot = torch.randn(batch, 576, 2048, generator=g).to(torch.bfloat16).to(device)
    return clip, tok, pixel, cot


def make_step(clip, tok, pixel, cot, world):
    from libra_amd.dp import BucketedGradReducer
    named = list(clip.named_parameters())

    def step():
        for _, p in named:
            p.grad = None
        feat, h2d, idx, ids, _, _ = tok.model.encode_flat(pixel, offset=32000, boi=32512, eoi=32513, want_ids=True,
                                                         want_quant=False)
        if world > 1:
            # each layer's gradients enter their RCCL all-reduce while the layers below are still in backward
            red = BucketedGradReducer(bucket_bytes=48 << 20)
            with red.capture():
                feat.backward(cot)
            red.finish_into(named)
        else:
            feat.backward(cot)
        return ids
    return step


def build_libra(device, batch, seq=2048, world=1):
    """BASELINE configs[2]/[3] shape: the reference's real pretraining step — frozen CLIP ViT + VQ encode under no_grad
    (clip_encoder.py:53, image_tokenizer.py:70) -> tensor assembly -> Libra-11B routed decoder fwd+bwd with the language
    stream frozen (modeling_libra.py:1342-1346: 4.27 B trainable "vision" parameters)."""
    from libra_amd.libra import LibraConfig, LibraForCausalLM, apply_freeze_policy, assemble_inputs, get_labels
    clip, tok, pixel, _ = build(device, batch)
    clip.requires_grad_(False)
    tok.model.encoder.allow_grad = False
    with torch.device(device):
        dec = LibraForCausalLM(LibraConfig())
    dec = dec.to(torch.bfloat16)
    with torch.no_grad():
        for n, p in dec.named_parameters():
            if "bridge" in n and n.endswith("weight_B"):
                p.normal_(0, 0.02)             # zero-initialised upstream; make the bridge path numerically live
    apply_freeze_policy(dec, frozen_language=True)
    V, L = 32000, 578
    PH = V - 1
    g = torch.Generator().manual_seed(42)
    text = torch.randint(3, V - 2, (batch, seq), generator=g)
    text[:, 0] = 1
    text[:, 1:1 + L] = PH
    text = text.to(device)
    am = torch.ones(batch, seq, dtype=torch.long, device=device)
    spans = [[(1 + L, 2 + L)] for _ in range(batch)]
    from libra_amd.dp import BucketedGradReducer
    named = [(n, p) for n, p in dec.named_parameters() if p.requires_grad]
    params = [p for _, p in named]
    trainable = {n for n, _ in named}

    def step():
        for p in params:
            p.grad = None
        with torch.no_grad():
            img = tok.encode(pixel)
        inp = assemble_inputs(text, am, img, img_ph_token_id=PH, img_gen_token_id=V - 2, boi_token_id=tok.boi_token_id,
                              num_codebook=2, max_vision_token_length=L)
        labels = get_labels(inp, spans, boi_token_id=tok.boi_token_id, bos_token_id=1)
        out = dec(input_ids=inp["input_ids"], attention_mask=inp["attention_mask"], vision_indices=inp["vision_indices"],
                  contiguous_signal=inp["coninous_signal"], labels=labels)
        if world > 1:
            # 8.5 GB of bf16 gradients per step: each decoder layer's ~267 MB goes out while the layers below are in backward
            red = BucketedGradReducer(bucket_bytes=256 << 20, only=trainable)
            with red.capture():
                out.loss.backward()
            red.finish_into(named)
        else:
            out.loss.backward()
        return out.loss.detach()
    return step, params


def cpu_baseline(sample_iters=3):
    """The CPU oracle (oracle/vit_oracle.py, proven equal to the reference's modules on the golden fixtures) timed on
    this box's host cores: ViT-L/14@336 fwd+bwd, B=1, fp32."""
    from oracle import vit_oracle as VO
    try:
        n = len(os.sched_getaffinity(0))          # cores this container may actually use
    except AttributeError:
        n = os.cpu_count() or 1
    n = max(1, min(n, 64))                        # torch CPU GEMMs stop scaling (and oversubscribe) beyond this
    torch.set_num_threads(n)
    sd = VO.random_vit_state_dict(hidden=1024, inter=4096, layers=24, patch=14, image=336, seed=42)
    sd = {k: v.requires_grad_(True) for k, v in sd.items()}
    g = torch.Generator().manual_seed(42)
    x = torch.randn(1, 3, 336, 336, generator=g)
    ct = torch.randn(1, 576, 2048, generator=g)

    def one():
        for v in sd.values():
            v.grad = None
        hs = VO.vit_hidden_states(sd, x, patch=14, heads=16, layers=24)
        f = VO.feature_select(hs, [-2, -3], square=False)
        (f * ct).sum().backward()
    t0 = time.perf_counter()
    one()                                          # warm-up (also sizes the sample)
    warm = time.perf_counter() - t0
    iters = max(1, min(sample_iters, int(20.0 / max(warm, 1e-3))))     # ~20 s of CPU work
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    dt = (time.perf_counter() - t0) / iters
    return {"value": round(1.0 / dt, 4), "unit": "images/s", "cores": n, "kind": "port",
            "host_cpus": os.cpu_count(),
            "sample": f"ViT-L/14@336 fwd+bwd (feature cotangent), B=1, fp32, {iters} timed iters after 1 warm-up "
                      f"({warm:.1f} s)"}


def hbm_traffic(workload):
    """Mean HBM bytes per GEMM launch from the committed PMC passes (tools/hbm_traffic.sh -> profiles/); None if absent."""
    path = os.path.join(ROOT, "profiles", f"r01_hbm_traffic_{workload}.json")
    try:
        with open(path) as f:
            return round(json.load(f)["gemm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--workload", choices=["vit", "libra"], default="vit",
                    help="vit = BASELINE configs[1] (headline line); libra = full pretraining step, ViT+VQ (no grad) -> "
                         "Libra-11B routed decoder fwd+bwd, bs 8, seq 2048")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL over xGMI.  (LIBRA_DIST_BACKEND=gloo lets the N>1 code path be exercised on a single-GPU box.)
        backend = os.environ.get("LIBRA_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))

    if args.batch is None:
        args.batch = 32 if args.workload == "vit" else 8
    if args.workload == "vit":
        clip, tok, pixel, cot = build(device, args.batch)
        step = make_step(clip, tok, pixel, cot, world)
    else:
        step, _ = build_libra(device, args.batch, world=world)

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)
    note(f"built model; world={world} batch={args.batch}")
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ms = dt / args.steps * 1e3
    ips = args.batch * world * args.steps / dt
    note(f"timed {args.steps} steps: {ms:.2f} ms/step, {ips:.1f} images/s")

    # ---- roofline leg: one instrumented step, HIP events around every GEMM launch on the launch stream (the whole step
    # runs on one stream, so a bracket contains exactly its own launch) ----
    from libra_amd import kernels as K
    with K.LaunchProfile() as prof:
        step()
    recs = prof.finish()
    gem = [(w[0], t) for k, w, t in recs if k == "gemm"]
    gbytes = sum(w[1] for k, w, t in recs if k == "gemm") / max(len(gem), 1)
    gflop = sum(w for w, _ in gem) / 1e9
    gms = sum(t for _, t in gem)
    achieved = gflop / gms if gms > 0 else 0.0        # GFLOP/ms == TFLOP/s
    if args.workload == "vit":
        gflop_step_img = GFLOP_FWD_PER_IMG + 2 * (GFLOP_FWD_PER_IMG - GFLOP_LAYER)   # bwd skips the unused last layer
    else:
        # ViT fwd + decoder fwd (25.44 T) + decoder bwd: dgrad everywhere, wgrad for the vision weights, attention bwd 2.5x
        dec_bwd = 32 * (595.0 + 153.34 + 2.5 * 34.4 + 153.34) + 385.0 + 2 * 4.8
        gflop_step_img = GFLOP_FWD_PER_IMG + 25440.0 + dec_bwd
    roof = {"bound": "mfma", "kernel": "gemm_bf16_nt_kernel", "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS,
            "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": hbm_traffic(args.workload),
            "traffic_unit": "HBM bytes / launch (PMC, profiles/)", "algorithmic_bytes_per_launch": round(gbytes),
            "timing": "HIP events on the launch stream around every GEMM launch of one extra step",
            "launches": len(gem), "avg_launch_us": round(gms / max(len(gem), 1) * 1e3, 1),
            "gemm_ms_per_step": round(gms, 2),
            "whole_step_frac": round(ips / world * gflop_step_img / 1e3 / PEAK_BF16_TFLOPS, 4)}

    out = {"metric": "images/sec/GPU fwd+bwd (ViT+bridge, 336px, seq2048) at 1/2/4/8 MI355X", "value": round(ips, 2),
           "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16", "data": "synthetic",
           "config": {"workload": ("configs[1]: ViT-L/14@336 + VQ encode fwd/bwd bf16, bs=32/GPU (LLM frozen)"
                                   if args.workload == "vit" else
                                   "configs[2]/[3]: ViT+VQ encode (no grad) -> Libra-11B routed decoder fwd+bwd, frozen language, "
                                   "bs=8/GPU, seq 2048, one 336px image per sequence"),
                      "global_batch": args.batch * world, "image": "3x336x336", "vit_tokens": 577,
                      "parallelism": f"dp{world}", "algorithmic_gflop_per_image": round(gflop_step_img, 1),
                      "value_per_gpu": round(ips / world, 2)},
           "roofline": roof}
    if world == 1 and rank == 0 and not args.no_cpu_baseline and args.workload == "vit":
        note("timing the CPU oracle on the host cores ...")
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
