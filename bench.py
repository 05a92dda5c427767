#!/usr/bin/env python
"""bench.py — images/sec of Libra's vision-to-LLM hot path on MI355X (BASELINE.json metric:
"images/sec/GPU fwd+bwd (ViT+bridge, 336px, seq2048)").

HEADLINE (default, `--workload bridge`, BASELINE configs[2] shapes): one step =
    CLIP ViT-L/14@336 forward + VQ encode (no grad: the reference freezes both, clip_encoder.py:53, image_tokenizer.py:70)
    -> LibraTokenizer tensor assembly + get_labels
    -> Libra-11B routed-bridge decoder forward + backward (32 layers, LLaMA-2-7B text stream frozen = the reference's
       pretraining recipe, 4.27 B trainable "vision" parameters; loss = the dual-head CE)
    [-> fused AdamW with --with-optimizer]  [+ RCCL gradient exchange overlapped with backward when N > 1]
bs = 8 image-sequences / GPU (one 336 px image + 1 470 text tokens per 2 048-token sequence), bf16, random-init weights,
synthetic inputs resident in HBM.  value = image-sequences / s, whole job.

SECOND LEG (`--workload vit`, BASELINE configs[1]; at N=1 it also runs after the headline and is reported under "extra"):
    ViT-L/14@336 forward -> feature select [-2,-3] -> VQ encode -> backward of a fixed feature cotangent through the ViT, bs 32.

One JSON line on rank 0 with "roofline" (dominant kernel family = the bf16 MFMA GEMM; per-launch HIP-event times on the
launch stream in one extra instrumented step) and "cpu_baseline" (the CPU oracle on this box's host cores, N=1 only).
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
PEAK_HBM_GBPS = 8000.0            # HBM3E peak (MI355X_MICROARCH.md)
ACHIEVABLE_HBM_GBPS = 6300.0      # what a streaming kernel reaches on this chip (same guide; the row kernels run 4.7-6.0 TB/s)
METRIC = "images/sec/GPU fwd+bwd (ViT+bridge, 336px, seq2048) at 1/2/4/8 MI355X"


def build_vit(device, batch, embed_dim=512):
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
    g = torch.Generator().manual_seed(42)
    pixel = torch.randn(batch, 3, 336, 336, generator=g).to(torch.bfloat16).to(device)
    cot = torch.randn(batch, 576, 2048, generator=g).to(torch.bfloat16).to(device)
    return clip, tok, pixel, cot


class Workload:
    """step() runs one step; `buckets` (dp.GradBuckets or None) and `exchange` toggle the data-parallel exchange."""
    buckets = None
    opt = None
    exchange = True


def make_vit(device, batch, world, mode):
    """configs[1]: the ViT's parameters take gradients (extension over the reference)."""
    from libra_amd import dp, vit_engine
    clip, tok, pixel, cot = build_vit(device, batch)
    clip.requires_grad_(True)
    tok.model.encoder.allow_grad = True
    named = list(clip.named_parameters())
    w = Workload()
    if world > 1:
        L = VIT_L["num_hidden_layers"]
        w.buckets = dp.GradBuckets(named, bucket_bytes=48 << 20, group_fn=lambda n: vit_engine.emit_group(n, L), mode=mode)

    def step():
        for _, p in named:
            p.grad = None
        feat, h2d, idx, ids, _, _ = tok.model.encode_flat(pixel, offset=32000, boi=32512, eoi=32513, want_ids=True,
                                                         want_quant=False)
        if w.buckets is not None and w.exchange:
            # each layer's gradients enter their RCCL exchange while the layers below are still in backward
            with w.buckets.capture():
                feat.backward(cot)
            w.buckets.finish_into(named)
        else:
            feat.backward(cot)
        return ids
    w.step = step
    w.named = named
    return w


def make_bridge(device, batch, seq, world, mode, *, with_optimizer=False, recompute=False, accum=1, full_finetune=False,
                layers=None, max_grad_norm=1.0):
    """configs[2]/[3]: the reference's real pretraining step (see module docstring).  `layers` (tests only): a shallower decoder
    of the same width - the bench itself always runs the 32 layers of the configuration."""
    from libra_amd import decoder_engine as DE
    from libra_amd import dp
    from libra_amd.libra import LibraConfig, LibraForCausalLM, apply_freeze_policy, assemble_inputs, get_labels, plan_assembly
    clip, tok, pixel, _ = build_vit(device, batch)
    clip.requires_grad_(False)
    tok.model.encoder.allow_grad = False
    cfg = LibraConfig(max_position_embeddings=max(2048, seq), **({} if layers is None else {"num_hidden_layers": layers}))
    with torch.device(device):
        dec = LibraForCausalLM(cfg)
    dec = dec.to(torch.bfloat16)
    with torch.no_grad():
        for n, p in dec.named_parameters():
            if "bridge" in n and n.endswith("weight_B"):
                p.normal_(0, 0.02)             # zero-initialised upstream; make the bridge path numerically live
    apply_freeze_policy(dec, frozen_language=not full_finetune)       # configs[4]: the instruction recipe trains all 11 B
    dec.train()
    if recompute:
        dec.gradient_checkpointing_enable()
    V, L = 32000, 578
    PH = V - 1
    g = torch.Generator().manual_seed(42)
    text = torch.randint(3, V - 2, (batch, seq), generator=g)
    text[:, 0] = 1
    text[:, 1:1 + L] = PH
    text = text.to(device)
    am = torch.ones(batch, seq, dtype=torch.long, device=device)
    spans = [[(1 + L, 2 + L)] for _ in range(batch)]
    named = [(n, p) for n, p in dec.named_parameters() if p.requires_grad and n != "vision_hidden_placeholder"]
    params = [p for _, p in named]
    w = Workload()
    if world > 1 or with_optimizer:
        # 8.5 GB of bf16 gradients per step: one decoder layer's ~267 MB per bucket goes out while the layers below are in backward
        nl = cfg.num_hidden_layers
        w.buckets = dp.GradBuckets(named, bucket_bytes=256 << 20, group_fn=lambda n: DE.emit_group(n, nl), mode=mode)
    if with_optimizer:
        # libra_pretrain.yaml:83-91 (+ max_grad_norm 1.0: the norm pass over the buckets and the clipped update are in the step)
        w.opt = dp.FlatAdamW(w.buckets, named, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01, max_grad_norm=max_grad_norm)

    def step():
        loss = None
        for k in range(accum):
            if w.buckets is None:
                for p in params:
                    p.grad = None
            # (as LibraTokenizer.forward: placeholder positions before the encoder is queued; `text` is this benchmark's static batch,
            #  complete since setup, so the host read may run beside the previous step's backward instead of behind it)
            plan = plan_assembly(text, img_ph_token_id=PH, side_stream=True)
            with torch.no_grad():
                img = tok.encode(pixel)
            inp = assemble_inputs(text, am, img, img_ph_token_id=PH, img_gen_token_id=V - 2, boi_token_id=tok.boi_token_id,
                                  num_codebook=2, max_vision_token_length=L, plan=plan)
            labels = get_labels(inp, spans, boi_token_id=tok.boi_token_id, bos_token_id=1)
            out = dec(input_ids=inp["input_ids"], attention_mask=inp["attention_mask"], vision_indices=inp["vision_indices"],
                      contiguous_signal=inp["coninous_signal"], labels=labels)
            loss = out.loss / accum if accum > 1 else out.loss
            if w.buckets is not None and (w.exchange or w.opt is not None):
                with w.buckets.capture(sync=(k == accum - 1)):
                    loss.backward()
                w.buckets.finish_into(named)
            else:
                loss.backward()
        if w.opt is not None:
            w.opt.step()
        return loss.detach()
    w.step = step
    w.model = dec
    w.named = named
    return w


def _median_time(fn, warm, iters, budget_s):
    """median of up to `iters` timed calls after `warm` warm-ups, stopping early once `budget_s` is spent -> (median, n)."""
    t_start = time.perf_counter()
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s:
            break
    ts.sort()
    return ts[len(ts) // 2], len(ts)


def cpu_leg(name, seq, threads):
    """One leg of the CPU baseline, run in a CHILD process (a hard wall-clock limit is the only reliable bound on a host
    whose thread scaling is unknown).  Prints one JSON object."""
    from oracle import libra_oracle as LO
    from oracle import vit_oracle as VO
    torch.set_num_threads(max(1, threads))
    g = torch.Generator().manual_seed(42)
    if name in ("vit_fp32", "vit_bf16"):
        dt = torch.float32 if name == "vit_fp32" else torch.bfloat16
        sd = VO.cast_sd(VO.random_vit_state_dict(hidden=1024, inter=4096, layers=24, patch=14, image=336, seed=42), dt)
        x = torch.randn(1, 3, 336, 336, generator=g).to(dt)

        def vit():
            with torch.no_grad():
                hs = VO.vit_hidden_states(sd, x, patch=14, heads=16, layers=24)
                return VO.feature_select(hs, [-2, -3], square=False)
        t, n = _median_time(vit, 2, 5, 20.0)
        print(json.dumps({"leg": name, "seconds": t, "iters": n, "warmups": 2, "threads": torch.get_num_threads()}))
        return
    H, heads, L = 4096, 32, 578
    lsd = {k: v.float().requires_grad_(True) for k, v in LO.random_layer_state_dict(seed=5).items()}
    xs = torch.randn(1, seq, H, generator=g)
    vi = torch.full((1, seq), L, dtype=torch.long)
    vi[0, 1:1 + L] = torch.arange(L)
    flag = vi < L
    mask = LO.additive_mask(torch.ones(1, seq, dtype=torch.long), seq, torch.float32)
    pos = torch.arange(seq).unsqueeze(0)
    cos, sin = LO.rope_tables(128, max(2048, seq))
    ct = torch.randn(1, seq, H, generator=g)

    def layer():
        for v in lsd.values():
            v.grad = None
        xin = xs.clone().requires_grad_(True)
        y = LO.decoder_layer(lsd, 0, xin, flag, mask, pos, heads, 1e-6, cos, sin)
        (y * ct).sum().backward()
    t, n = _median_time(layer, 1, 5, 80.0)
    print(json.dumps({"leg": name, "seconds": t, "iters": n, "warmups": 2, "threads": torch.get_num_threads()}))


def physical_cores():
    """(physical cores, logical CPUs) this process may run on: distinct (socket, core id) pairs of /proc/cpuinfo within the
    affinity mask.  SMT siblings share one FPU pipe: a GEMM thread per LOGICAL cpu only oversubscribes it."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    cores, cpu, phys = set(), None, None
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("processor"):
                    cpu = int(ln.split(":")[1])
                elif ln.startswith("physical id"):
                    phys = int(ln.split(":")[1])
                elif ln.startswith("core id") and cpu in allowed:
                    cores.add((phys, int(ln.split(":")[1])))
    except OSError:
        pass
    return (len(cores) or len(allowed)), len(allowed)


def cpu_baseline(seq=2048, workload="bridge"):
    """the CPU oracle (oracle/*.py, proven equal to the reference's modules on the golden fixtures) on the host
    cores of this box, same seeded synthetic inputs:
      (i)  config 1 exactly - ViT-L/14@336 forward, bs 1, fp32 and bf16, median of 5 iterations after 2 warm-ups;
      (ii) one full-width routed decoder layer forward+backward at B=1, S=seq (578 vision tokens), fp32, median of 5 iterations
           after 1 warm-up, x32 layers (a full 11 B fwd+bwd does not fit a sensible CPU time budget).
    value = image-sequences/s of the headline workload = 1 / (ViT fwd + 32 x layer fwd+bwd).
    Threads: one per PHYSICAL core, pinned (OMP_PROC_BIND=spread, OMP_PLACES=cores) - with one thread per logical CPU (SMT
    siblings, unpinned) torch's CPU GEMMs on the 2-socket EPYC hosts did not finish a 0.4 s workload in a minute (round 2).  The
    round-2 setting (64 unpinned threads) is timed beside it; the faster of the two is `value`, both are reported.
    Every leg runs in a child process under a hard time limit."""
    import subprocess
    n_phys, n_log = physical_cores()
    legs, notes = {}, []

    def run_leg(name, n, limit, pinned, tag):
        env = dict(os.environ)
        env["OMP_NUM_THREADS"] = str(n)
        if pinned:
            env.update(OMP_PROC_BIND="spread", OMP_PLACES="cores")
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-leg", name, "--cpu-threads", str(n), "--seq", str(seq)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit, env=env)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode == 0 and line:
                res = json.loads(line[-1])
                res["pinned"] = bool(pinned)
                legs.setdefault(tag, {})[name] = res
                return res
            notes.append(f"{name}@{n} threads ({tag}): rc {r.returncode}")
        except subprocess.TimeoutExpired:
            notes.append(f"{name}@{n} threads ({tag}): no result within {limit:.0f} s")
        return None

    configs = [("physical_cores_pinned", n_phys, True)]
    if n_phys != 64 and n_log > 64:
        configs.append(("threads64_unpinned", 64, False))
    need_layer = workload == "bridge"
    for tag, n, pinned in configs:
        if run_leg("vit_fp32", n, 40.0, pinned, tag) is None:
            continue
        run_leg("vit_bf16", n, 30.0, pinned, tag)
        if need_layer:
            run_leg("layer_fwd_bwd_fp32", n, 100.0, pinned, tag)

    def per_seq(tag):
        l = legs.get(tag, {})
        vit, lay = l.get("vit_fp32"), l.get("layer_fwd_bwd_fp32")
        if vit is None or (need_layer and lay is None):
            return None
        return vit["seconds"] + (32 * lay["seconds"] if need_layer else 0.0)
    done = {tag: per_seq(tag) for tag, _, _ in configs}
    done = {k: v for k, v in done.items() if v is not None}
    out = {"unit": "images/s", "kind": "port", "host_cpus": os.cpu_count(), "logical_cpus": n_log, "physical_cores": n_phys,
           "legs": legs}
    if notes:
        out["notes"] = notes
    if not done:
        out.update(value=None, cores=n_phys, sample="CPU legs did not finish inside their time limits: " + "; ".join(notes))
        return out
    best = min(done, key=done.get)
    bl = legs[best]
    out.update(value=round(1.0 / done[best], 5), cores=bl["vit_fp32"]["threads"], config=best,
               by_config={k: round(1.0 / v, 5) for k, v in done.items()},
               vit_fwd_bs1_images_per_s={k[4:]: round(1.0 / v["seconds"], 3) for k, v in bl.items() if k.startswith("vit_")},
               decoder_layer_fwd_bwd_s=round(bl["layer_fwd_bwd_fp32"]["seconds"], 3) if need_layer else None,
               sample=f"config 1 exactly (ViT-L/14@336 fwd, bs 1, fp32 + bf16, median of <=5 after 2 warm-ups) + ONE full-width routed "
                      f"decoder layer fwd+bwd at B=1, S={seq}, 578 vision tokens, fp32 (median of <=5 after 1 warm-up) x 32 layers; "
                      f"threads / iterations per leg {[(t, k, v['threads'], v['iters']) for t, l in legs.items() for k, v in l.items()]}")
    return out


def gemm_sources_sha():
    """Fingerprint of the sources that decide the GEMM kernels' memory traffic (tile structures, planner, launch order): written
    into the traffic file by tools/hbm_traffic.py and compared here, so a traffic number measured on OTHER kernels says so."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "libra_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.startswith("gemm_") or f == "hip_common.hpp":
            h.update(f.encode())
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def hbm_traffic(workload):
    """Mean HBM bytes per GEMM launch from the committed PMC passes (tools/hbm_traffic.sh -> profiles/) -> (bytes, file name,
    provenance dict); (None, None, None) if absent.  The PMC passes cannot run inside the timed command (they serialise kernels),
    so the number is read from the newest committed file and marked `stale` when the GEMM sources changed since it was measured."""
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_hbm_traffic_{workload}.json")
        try:
            with open(path) as f:
                d = json.load(f)
            now = gemm_sources_sha()
            prov = {"file": os.path.basename(path), "measured_at_gemm_sources_sha": d.get("gemm_sources_sha"),
                    "measured_at_head": d.get("head"), "running_gemm_sources_sha": now,
                    "stale": d.get("gemm_sources_sha") != now}
            return round(d["gemm_bytes_per_launch"]), os.path.basename(path), prov
        except (OSError, KeyError, ValueError):
            continue
    return None, None, None


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` (N > 1) outside torchrun: re-run this command line as N ranks of one node under
    torch.distributed.run (one process per GPU, RCCL over xGMI; rendezvous on 127.0.0.1) and return its exit code.  Without this
    the command would silently measure ONE GPU and report it as the N-GPU point of the scaling curve."""
    import socket
    import subprocess
    backend = os.environ.get("LIBRA_DIST_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < n:
        print(f"[bench] --gpus {n} needs {n} visible GPUs (one rank per GPU over RCCL), found {have}; "
              f"LIBRA_DIST_BACKEND=gloo exercises the N>1 code path on fewer devices", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL's cross-process buffer sharing needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] launching {n} ranks: {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def parse_showtopo(text: str):
    """`rocm-smi --showtopo` -> {"link_type": {(i, j): "XGMI" | "PCIE" | ...}, "hops": {...}, "weight": {...}} (what is present)."""
    out = {}
    sec = None
    cols = []
    for line in text.splitlines():
        if line.startswith("="):
            low = line.lower()
            sec = ("link_type" if "link type between" in low else "hops" if "hops between" in low
                   else "weight" if "weight between" in low else None)
            cols = []
            continue
        if sec is None or not line.strip():
            continue
        tok = line.split()
        if not cols:
            if all(t.startswith("GPU") for t in tok):
                cols = [int(t[3:]) for t in tok]
            continue
        if tok[0].startswith("GPU") and len(tok) == len(cols) + 1:
            i = int(tok[0][3:])
            for j, v in zip(cols, tok[1:]):
                out.setdefault(sec, {})[(i, j)] = v
    return out


def preflight(world: int, backend: str, captured_init_output: str = ""):
    """What the first real multi-GPU run needs to be diagnosable from its JSON line alone (rank 0): the collective library's
    version, the version line it prints under NCCL_DEBUG=VERSION, and the node's link topology between the first and the last rank's
    GPU (xGMI vs PCIe, hops).  Every probe is best effort and bounded in time; failures are recorded, never raised."""
    import subprocess
    pf = {"backend": backend, "visible_gpus": torch.cuda.device_count(),
          "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    try:
        v = torch.cuda.nccl.version()
        pf["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:
        pf["rccl_version"] = f"unavailable: {e!r}"[:120]
    lines = [l.strip() for l in captured_init_output.splitlines() if "RCCL" in l or "NCCL version" in l or "HIP version" in l]
    pf["nccl_debug_version_line"] = lines[:3] if lines else None
    try:
        r = subprocess.run(["rocm-smi", "--showtopo"], capture_output=True, text=True, timeout=30)
        topo = parse_showtopo(r.stdout)
        a, b = 0, max(world - 1, 0)
        pf["topology"] = {k: topo.get(k, {}).get((a, b)) for k in ("link_type", "hops", "weight")}
        pf["topology"]["pair"] = [a, b]
        kinds = sorted(set(v for (i, j), v in topo.get("link_type", {}).items() if i != j and i < world and j < world))
        pf["topology"]["link_types_among_ranks"] = kinds
    except Exception as e:
        pf["topology"] = f"rocm-smi --showtopo failed: {e!r}"[:160]
    return pf


class _CaptureFd1:
    """Redirect the PROCESS's stdout (fd 1) into a temporary file for the duration of the block: RCCL prints its NCCL_DEBUG=VERSION
    line with its own stdio, and stdout must stay the JSON line's alone."""

    def __enter__(self):
        import tempfile
        sys.stdout.flush()
        self.tmp = tempfile.TemporaryFile(mode="w+b")
        self.saved = os.dup(1)
        os.dup2(self.tmp.fileno(), 1)
        self.text = ""
        return self

    def __exit__(self, *exc):
        try:
            os.fsync(1)
        except OSError:
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)
        self.tmp.seek(0)
        self.text = self.tmp.read().decode("utf-8", "replace")
        self.tmp.close()
        return False


def step_check(w, named_params):
    """Correctness evidence of the step that was timed: its loss, the L2 norm of its gradients (libra_sumsq_bf16 over every
    trainable parameter's gradient: deterministic, so the number doubles as a checksum between runs / builds) and whether both
    are finite.  The inputs are fixed and no optimizer runs in the headline step, so every timed step computes exactly this."""
    from libra_amd import kernels as K
    loss = w.step()
    acc = torch.zeros(1, dtype=torch.float32, device=loss.device)
    n_grad = 0
    first = True
    for _, p in named_params:
        g = p.grad
        if g is None:
            continue
        K.sumsq(g.contiguous().view(-1), acc, accumulate=not first)
        first = False
        n_grad += g.numel()
    lf, gn = float(loss), float(acc.sqrt())
    return {"loss": round(lf, 6), "grad_norm": round(gn, 6), "grad_elements": n_grad,
            "finite": bool(torch.isfinite(loss).item() and gn == gn and gn != float("inf"))}


def timed(w, steps, warmup, world, device):
    for _ in range(warmup):
        w.step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        w.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def capture_step(w):
    """The step of `w` as ONE hipGraph (static shapes, no host read inside - true of the ViT + VQ fwd/bwd step): -> (Workload whose
    step() replays the graph, None) or (None, reason).  Warm-up runs on a side stream first (allocator, autograd engine, first-launch
    attribute calls), as torch.cuda.graph requires; gradients land in the graph's private pool and every replay refills them."""
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                w.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = w.step()
        graph.replay()
        torch.cuda.synchronize()
    except Exception as e:
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        return None, repr(e)[:200]
    g = Workload()
    g.named, g.graph, g.out = getattr(w, "named", None), graph, out

    def step():
        graph.replay()
        return out
    g.step = step
    return g, None


def vit_graph_leg(batch=32, steps=20, warmup=3):
    """Child-process entry (`bench.py --vit-graph-leg`): the configs[1] step captured as ONE hipGraph and replayed - in a process of its
    own because a capture that goes wrong can take the HIP context with it, and the parent still has the headline line to print.
    Prints one JSON object: replay throughput, the eager step's gradient checksum and whether the replay reproduces it."""
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    wv = make_vit(device, batch, 1, "allreduce")
    for _ in range(2):
        ids_eager = wv.step().clone()
    chk = grads_check(wv.named)
    wg, why = capture_step(wv)
    if wg is None:
        print(json.dumps({"graph_capture_failed": why}), flush=True)
        return
    dtg = timed(wg, steps, warmup, 1, device)
    chk_g = grads_check(wv.named)
    print(json.dumps({"images_per_s": round(batch * steps / dtg, 2), "ms_per_step": round(dtg / steps * 1e3, 3),
                      "graph_replay_equals_eager": bool(chk_g["grad_norm"] == chk["grad_norm"] and torch.equal(wg.out, ids_eager)),
                      "grad_norm": chk_g["grad_norm"]}), flush=True)


def grads_check(named):
    """L2 norm of every gradient of `named` (libra_sumsq_bf16: deterministic - a checksum between runs / schedules) and finiteness."""
    from libra_amd import kernels as K
    acc, n_grad, first = None, 0, True
    for _, p in named:
        g = p.grad
        if g is None:
            continue
        if acc is None:
            acc = torch.zeros(1, dtype=torch.float32, device=g.device)
        K.sumsq(g.contiguous().view(-1), acc, accumulate=not first)
        first = False
        n_grad += g.numel()
    gn = float(acc.sqrt()) if acc is not None else float("nan")
    return {"grad_norm": round(gn, 6), "grad_elements": n_grad, "finite": bool(gn == gn and gn != float("inf"))}


def roofline(w, workload, ips_per_gpu, gflop_step_img, ms_step=None):
    """One instrumented step: HIP events around every GEMM launch on the launch stream (the whole step runs on one stream,
    so a bracket contains exactly its own launch)."""
    from libra_amd import kernels as K
    with K.LaunchProfile() as prof:
        w.step()
    recs = prof.finish()
    gem = [(wk[0], t) for k, wk, t in recs if k == "gemm"]
    gbytes = sum(wk[1] for k, wk, t in recs if k == "gemm") / max(len(gem), 1)
    gflop = sum(f for f, _ in gem) / 1e9
    gms = sum(t for _, t in gem)
    achieved = gflop / gms if gms > 0 else 0.0        # GFLOP/ms == TFLOP/s
    shapes = {}                                       # per GEMM shape: launches, ms, TFLOP/s inside the step (the 48 largest by time)
    for k, wk, t in recs:
        if k == "gemm" and len(wk) > 2:
            e = shapes.setdefault(wk[2], [0, 0.0, 0.0])
            e[0] += 1; e[1] += t; e[2] += wk[0]
    by_shape = [{"shape": n, "launches": c, "ms": round(ms, 2), "tflops": round(fl / 1e9 / ms, 1) if ms > 0 else 0.0}
                for n, (c, ms, fl) in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:48]]
    traffic, src, prov = hbm_traffic("libra" if workload == "bridge" else "vit")
    # the rest of the step from the same instrumented pass: attention against the MFMA peak (causal-minimal FLOPs), the row kernels
    # against HBM (algorithmic bytes: each operand / result row once; 8 TB/s peak, ~6.3 TB/s achievable per MI355X_MICROARCH.md),
    # and what no bracket covers (torch glue, copies, small kernels, launch gaps) = whole step - bracketed time
    by_kernel = {"gemm": {"launches": len(gem), "ms": round(gms, 2), "tflops": round(achieved, 1), "frac_mfma": round(achieved / PEAK_BF16_TFLOPS, 4)}}
    for kind in ("attn_fwd", "attn_bwd"):
        rs = [(wk, t) for k, wk, t in recs if k == kind]
        if rs:
            tms, fl = sum(t for _, t in rs), sum(wk[0] for wk, _ in rs)
            by_kernel["bridge_" + kind] = {"launches": len(rs), "ms": round(tms, 2), "us_per_launch": round(tms / len(rs) * 1e3, 1),
                                           "tflops": round(fl / 1e9 / tms, 1), "frac_mfma": round(fl / 1e9 / tms / PEAK_BF16_TFLOPS, 4)}
    rows = {}
    for k, wk, t in recs:
        if k == "row":
            e = rows.setdefault(wk[2], [0, 0.0, 0.0])
            e[0] += 1; e[1] += t; e[2] += wk[1]
    if rows:
        rms, rby = sum(e[1] for e in rows.values()), sum(e[2] for e in rows.values())
        by_kernel["row_kernels"] = {"launches": sum(e[0] for e in rows.values()), "ms": round(rms, 2), "GBps": round(rby / rms / 1e6, 1),
                                    "frac_hbm_peak": round(rby / rms / 1e6 / PEAK_HBM_GBPS, 4),
                                    "frac_hbm_achievable": round(rby / rms / 1e6 / ACHIEVABLE_HBM_GBPS, 4),
                                    "each": {n: {"launches": c, "ms": round(ms_, 2), "GBps": round(by / ms_ / 1e6, 1)}
                                             for n, (c, ms_, by) in sorted(rows.items(), key=lambda kv: -kv[1][1])}}
    bracketed = sum(t for _, _, t in recs)
    by_kernel["launches_bracketed"] = len(recs)
    if ms_step is not None:
        by_kernel["unbracketed_ms"] = round(ms_step - bracketed, 2)      # (bracket times come from one extra step; ms_step from the timed region)
    return {"bound": "mfma", "kernel": "gemm_bf16_multi_kernel / gemm_bf16_nt_256_kernel (one 256x256x64 tile body, gemm256_body.hpp: multi-problem "
                                       "persistent launches and single launches; + gemm_bf16_nt_w_kernel 256x128 two per CU / "
                                       "gemm_bf16_nt_kernel 128x128 for tail rows and small problems, split-K wgrad slabs): every "
                                       "launch made through libra_gemm_bf16_nt* / libra_gemm_bf16_multi",
            "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
            "traffic_unit": f"HBM bytes / launch (PMC, profiles/{src})" if src else None, "traffic_provenance": prov,
            "algorithmic_bytes_per_launch": round(gbytes),
            "timing": "HIP events on the launch stream around every GEMM launch of one extra step",
            "launches": len(gem), "avg_launch_us": round(gms / max(len(gem), 1) * 1e3, 1), "gemm_ms_per_step": round(gms, 2),
            "gemm_gflop_per_step": round(gflop, 1), "by_shape": by_shape, "by_kernel": by_kernel,
            "whole_step_frac": round(ips_per_gpu * gflop_step_img / 1e3 / PEAK_BF16_TFLOPS, 4)}


def gflop_per_image(workload, seq=2048, full_finetune=False):
    if workload == "vit":
        return GFLOP_FWD_PER_IMG + 2 * (GFLOP_FWD_PER_IMG - GFLOP_LAYER)   # bwd skips the unused last layer
    # ViT fwd + decoder fwd (25.44 T at S=2048) + decoder bwd: dgrad everywhere, wgrad for the vision weights only (frozen
    # language), attention bwd 2.5x.  Per layer (SURVEY §8d formulas, Nv = 578):
    H, I, r, rg, Nv, V = 4096, 11008, 1024, 2752, 578, 32000
    Nl = seq - Nv
    text = 8 * Nl * H * H + 6 * Nl * H * I
    vis = 16 * Nv * H * r + 4 * Nv * (H * rg + rg * I) + 2 * Nv * (I * r + r * H)
    bridge = 8 * seq * H * 8
    attn = 2 * seq * seq * H
    heads = 2 * Nl * H * V + 4 * Nv * H * 514
    fwd = 32 * (text + vis + bridge + attn) + heads
    tw = 2 if full_finetune else 1                       # frozen text weights need no weight gradient
    bwd = 32 * (tw * text + 2 * vis + 2 * bridge + 2.5 * attn) + tw * 2 * Nl * H * V + 2 * 4 * Nv * H * 514
    return GFLOP_FWD_PER_IMG + (fwd + bwd) / 1e9


class _Watchdog:
    """If `cancel()` has not been called after `seconds`, print one line per rank and re-execute this rank with the plain
    all-reduce and no probe (every rank of a hung collective trips its own watchdog within the same second; the new processes
    rendezvous again on the launcher's MASTER_ADDR / MASTER_PORT).  A rank that is already a fallback gives up instead (exit 4)."""

    def __init__(self, seconds, rank, what, argv, fallback):
        import threading
        self.t = threading.Timer(seconds, self._fire, (seconds, rank, what, list(argv), fallback))
        self.t.daemon = True
        if seconds > 0:
            self.t.start()

    def cancel(self):
        self.t.cancel()

    @staticmethod
    def _fire(seconds, rank, what, argv, fallback):
        print(f"[bench] rank {rank}: watchdog: {what} did not finish in {seconds:.0f} s", file=sys.stderr, flush=True)
        if "--fallback-from" in argv:
            os._exit(4)
        args = [a for a in argv[1:]]
        if "--exchange" in args:
            i = args.index("--exchange")
            del args[i:i + 2]
        try:                                   # the HIP / RCCL context's descriptors (KFD, dmabufs, sockets) need not all be CLOEXEC:
            os.closerange(3, 1 << 16)          # nothing of this image may leak into the re-executed rank
        except OSError:
            pass
        os.execv(sys.executable, [sys.executable, os.path.abspath(argv[0])] + args + ["--exchange", "allreduce", "--fallback-from", fallback])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--workload", choices=["bridge", "vit", "libra"], default="bridge",
                    help="bridge (= libra) = the headline, BASELINE configs[2]: ViT+VQ (no grad) -> Libra-11B routed decoder "
                         "fwd+bwd, bs 8, seq 2048; vit = configs[1]: ViT+VQ fwd/bwd bs 32")
    ap.add_argument("--with-optimizer", action="store_true", help="include the fused AdamW update in the step (configs[3])")
    ap.add_argument("--recompute", action="store_true", help="gradient checkpointing per decoder layer (the recipes' setting)")
    ap.add_argument("--accum", type=int, default=1, help="gradient-accumulation micro-steps per step")
    ap.add_argument("--full-finetune", action="store_true",
                    help="configs[4]: every parameter trainable (instruction recipe) instead of the frozen-language pretraining policy")
    ap.add_argument("--exchange", choices=["auto", "allreduce", "rs_ag", "zero1"], default="auto",
                    help="N>1 gradient exchange; auto = probe allreduce and rs_ag during warm-up and keep the faster")
    ap.add_argument("--cu-reserve", type=int, default=0,
                    help="N>1: run the step on a CU-masked stream that leaves this many compute units to RCCL's reduction kernels "
                         "(and size the persistent attention grids to the rest); reported in extra.cu_budget")
    ap.add_argument("--probe-watchdog-s", type=float, default=120.0,
                    help="N>1: if the exchange probe has not finished after this many seconds every rank re-executes itself with "
                         "--exchange allreduce (no probe), so that a mode that hangs on this node still yields a result")
    ap.add_argument("--fallback-from", default=None, help=argparse.SUPPRESS)   # set by the watchdog's re-exec: what hung
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-leg", default=None, help=argparse.SUPPRESS)         # child-process entry of cpu_baseline()
    ap.add_argument("--vit-graph-leg", action="store_true", help=argparse.SUPPRESS)   # child-process entry of the ViT leg's hipGraph replay
    ap.add_argument("--cpu-threads", type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument("--no-extra", action="store_true", help="skip the extra legs (ViT leg, optimizer leg) at N=1")
    ap.add_argument("--no-multi", action="store_true",
                    help="A/B switch: the round-5 launch schedule (one launch per GEMM) instead of the multi-problem launches")
    ap.add_argument("--graph", action="store_true", help="--workload vit only: time the step as one replayed hipGraph")
    ap.add_argument("--chain", action="store_true",
                    help="A/B switch: fold the first low-rank stage of each vision pair into the multi-problem launch of its second "
                         "stage (device-side producer -> consumer wait; decoder_engine.CHAIN)")
    args = ap.parse_args()
    if args.chain:
        from libra_amd import decoder_engine as _DE
        _DE.CHAIN = True
    if args.no_multi:
        from libra_amd import decoder_engine as _DE, vit_engine as _VE
        _DE.MULTI = False
        if hasattr(_VE, "MULTI"):
            _VE.MULTI = False
    if args.workload == "libra":
        args.workload = "bridge"
    if args.cpu_leg:
        cpu_leg(args.cpu_leg, args.seq, args.cpu_threads)
        return
    if args.vit_graph_leg:
        if args.no_multi:
            from libra_amd import vit_engine as _VE
            _VE.MULTI = False
        vit_graph_leg()
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))       # `python bench.py --gpus N` without torchrun: start the N ranks ourselves

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the two must agree (launch with "
                         f"`python bench.py --gpus N`, or torchrun --nproc-per-node N bench.py --gpus N)")
    if world > 1 and os.environ.get("LIBRA_DIST_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} needs {world} visible GPUs (one rank per GPU over RCCL), found {torch.cuda.device_count()}")
    local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    backend = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL over xGMI.  (LIBRA_DIST_BACKEND=gloo lets the N>1 code path be exercised on a single-GPU box.)
        backend = os.environ.get("LIBRA_DIST_BACKEND", "nccl")
        os.environ.setdefault("NCCL_DEBUG", "VERSION")       # one line per job (rank 0), captured below - not a perf knob
        with _CaptureFd1() as cap:                           # the library's own stdout chatter stays out of the JSON stream
            dist.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
            t = torch.ones(1, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t)                               # first collective: communicator creation prints the version line
            if backend == "nccl":
                torch.cuda.synchronize()
        init_output = cap.text
        if float(t.item()) != float(world):
            raise SystemExit(f"preflight all-reduce returned {float(t.item())}, expected {world}")

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    if args.batch is None:
        args.batch = 32 if args.workload == "vit" else 8
    mode = "allreduce" if args.exchange == "auto" else args.exchange
    if args.with_optimizer and args.exchange == "auto" and world > 1:
        mode = "zero1"                       # sharded optimizer state: the exchange IS reduce-scatter + parameter all-gather

    def make(mode_):
        if args.workload == "vit":
            return make_vit(device, args.batch, world, mode_)
        return make_bridge(device, args.batch, args.seq, world, mode_, with_optimizer=args.with_optimizer,
                           recompute=args.recompute, accum=args.accum, full_finetune=args.full_finetune)
    w = make(mode)
    note(f"built {args.workload}; world={world} batch={args.batch} exchange={mode if world > 1 else None}")

    extra = {}
    if world > 1 and rank == 0:
        extra["preflight"] = preflight(world, backend, init_output)
        note(f"preflight: {extra['preflight']}")
    cu = None
    if args.cu_reserve > 0:
        from libra_amd import kernels as K
        cu = K.ReservedCUStream(args.cu_reserve)
        cu.__enter__()                                   # every step below (probe, timed region, diagnostics) runs on the masked stream
        extra["cu_budget"] = {"reserved_for_rccl": cu.reserve, "compute_cus": cu.cus, "persistent_kernel_cus": cu.persistent_cus,
                              "physical_cus": K.cu_count()}
        note(f"cu budget: {extra['cu_budget']}")
    if args.fallback_from:
        extra["exchange_fallback"] = f"the probe of {args.fallback_from!r} did not finish in {args.probe_watchdog_s:.0f} s; re-executed with allreduce"
    if world > 1 and args.exchange == "auto" and not args.with_optimizer:
        # probe both exchange algorithms on this node (xGMI full mesh: direct reduce-scatter + all-gather vs whatever RCCL's
        # all-reduce picks) and keep the faster for the timed region; both numbers are reported.  A mode that RAISES on any rank is
        # dropped on every rank (the ranks agree through a MIN all-reduce of their success flags, made in the mode that worked);
        # a mode that HANGS trips the watchdog, which re-executes every rank with the plain all-reduce.
        probe, failed = {}, {}
        for m in ("allreduce", "rs_ag"):
            dog = _Watchdog(args.probe_watchdog_s, rank, f"exchange probe '{m}'", sys.argv, fallback=m)
            ok = 1.0
            try:
                w.buckets.mode = m
                probe[m] = timed(w, 2, 1, world, device) / 2 * 1e3
            except Exception as e:
                ok = 0.0
                failed[m] = repr(e)[:200]
                print(f"[bench] rank {rank}: exchange probe '{m}' failed: {failed[m]}", file=sys.stderr, flush=True)
            # the watchdog stays armed ACROSS the agreement: if this rank's probe raised while the others hang inside the failed
            # mode's collective, their watchdogs re-execute them - and this rank, blocked in the flag all-reduce they never join,
            # must re-execute too or the new rendezvous waits for it forever
            try:
                w.buckets.mode = "allreduce"
                flag = torch.tensor([ok], device=device if backend == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if backend == "nccl":
                    torch.cuda.synchronize()
            finally:
                dog.cancel()
            if float(flag.item()) < 1.0:
                probe.pop(m, None)
                failed.setdefault(m, "failed on another rank")
        if not probe:
            raise SystemExit(f"every exchange mode failed on this node: {failed}")
        mode = min(probe, key=probe.get)
        w.buckets.mode = mode
        extra["exchange_probe_ms_per_step"] = {k: round(v, 2) for k, v in probe.items()}
        if failed:
            extra["exchange_probe_failed"] = failed
        note(f"exchange probe {extra['exchange_probe_ms_per_step']} -> {mode}" + (f" (failed: {failed})" if failed else ""))

    w_timed = w
    if args.graph:
        if args.workload != "vit" or world > 1:
            raise SystemExit("--graph: only the single-GPU ViT step has no host read inside (the headline step reads the placeholder plan)")
        w_timed, why = capture_step(w)
        if w_timed is None:
            raise SystemExit(f"--graph: capture failed: {why}")
        extra["launch"] = "one hipGraph per step (replay)"
    dt = timed(w_timed, args.steps, args.warmup, world, device)
    ms = dt / args.steps * 1e3
    ips = args.batch * args.accum * world * args.steps / dt
    note(f"timed {args.steps} steps: {ms:.2f} ms/step, {ips:.2f} images/s")

    if world > 1:
        # exposed communication = step time with the exchange minus step time without it (same process, same buffers).
        # (Every rank takes the same path through these diagnostics - their collectives match -, and an error in them must not cost
        #  the headline line: the timed region above is already done.)
        extra.update(exchange=mode, backend=backend, dist_world_size=dist.get_world_size())
        try:
            w.exchange = False
            dt0 = timed(w, max(2, args.steps // 2), 1, world, device)
            w.exchange = True
            ms0 = dt0 / max(2, args.steps // 2) * 1e3
            extra.update(ms_per_step_without_exchange=round(ms0, 3), exposed_comm_ms=round(ms - ms0, 3),
                         exchanged_bytes_per_step=int(w.buckets.bytes_exchanged // max(w.buckets.launches, 1)
                                                      * len(w.buckets.buckets)),
                         grad_bucket_bytes=w.buckets.total_bytes, buckets=len(w.buckets.buckets))
        except Exception as e:
            w.exchange = True
            extra["exposed_comm_error"] = repr(e)[:200]

    check = None
    if args.workload == "bridge" and not args.with_optimizer:
        try:
            check = step_check(w, w.named)
            note(f"step check: {check}")
        except Exception as e:
            check = {"error": repr(e)[:200]}
    gpi = gflop_per_image(args.workload, args.seq, args.full_finetune)
    try:
        roof = roofline(w, args.workload, ips / world, gpi, ms)
    except Exception as e:                           # (instrumented extra step; the timed result stands without it)
        roof = {"bound": "mfma", "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "achieved": None, "frac": None, "traffic": None,
                "error": repr(e)[:200]}
    if args.full_finetune or args.seq > 2048:
        cfg_name = "configs[4]-shaped (instruction tuning)"
    elif args.seq != 2048 or args.with_optimizer:
        cfg_name = "configs[3]-shaped (pretraining step)"
    else:
        cfg_name = "configs[2]"
    names = {"bridge": f"{cfg_name}: ViT-L/14@336 + VQ encode (no grad) -> tensor assembly -> Libra-11B routed-bridge decoder "
                       f"fwd+bwd, {'all 11.0 B parameters trainable' if args.full_finetune else 'LLaMA-2-7B text stream frozen (4.27 B trainable)'}, bs={args.batch}/GPU, seq {args.seq}, one "
                       "336px image per sequence, random-init",
             "vit": f"configs[1]: ViT-L/14@336 + VQ encode fwd/bwd bf16, bs={args.batch}/GPU (LLM frozen)"}
    out = {"metric": METRIC, "value": round(ips, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16", "data": "synthetic",
           "config": {"workload": names[args.workload], "global_batch": args.batch * args.accum * world, "image": "3x336x336",
                      "seq_len": args.seq if args.workload == "bridge" else 577, "parallelism": f"dp{world}",
                      "algorithmic_gflop_per_image": round(gpi, 1), "value_per_gpu": round(ips / world, 3),
                      "optimizer_in_step": bool(args.with_optimizer), "recompute": bool(args.recompute),
                      "full_finetune": bool(args.full_finetune),
                      "grad_accum": args.accum,
                      "gemm_schedule": "one launch per GEMM (--no-multi)" if args.no_multi else
                                       "multi-problem launches (libra_gemm_bf16_multi)" + (" + chained low-rank stages" if args.chain else "")},
           "roofline": roof}
    try:                                                 # which launch schedule actually ran (the engines ask the device once, K.gemm_multi_ok)
        from libra_amd import kernels as _K
        out["config"]["gemm_multi_selfcheck"] = _K._MULTI_CHECKED.get(device.index if device.index is not None else 0)
    except Exception:
        pass
    if check is not None:
        out["step_check"] = check         # loss + gradient norm of exactly the timed step (tests/test_configs_gpu.py::test_headline_step_full_size_loss_and_recompute_identity bounds them)

    if world == 1 and rank == 0 and not args.no_extra and args.workload == "bridge" and not args.with_optimizer \
            and not args.full_finetune:
        # ---- optimizer leg: the same step + fused AdamW on the 4.27 B trainable parameters (configs[3] names AdamW)
        try:
            del w
            torch.cuda.empty_cache()
            wo = make_bridge(device, args.batch, args.seq, 1, "allreduce", with_optimizer=True, recompute=args.recompute)
            n_opt = max(3, args.steps // 2)
            dto = timed(wo, n_opt, 2, 1, device)
            extra["with_optimizer"] = {"ms_per_step": round(dto / n_opt * 1e3, 3),
                                       "images_per_s": round(args.batch * n_opt / dto, 3),
                                       "optimizer_state_bytes": wo.opt.state_bytes,
                                       "note": "same step + libra_adamw_step over the flat gradient buckets (fp32 master, m, v)"}
            note(f"optimizer leg: {extra['with_optimizer']}")
            del wo
            torch.cuda.empty_cache()
        except Exception as e:                       # never lose the headline line to an extra leg
            extra["with_optimizer"] = {"error": repr(e)[:200]}
        # ---- second leg: configs[1]
        try:
            wv = make_vit(device, 32, 1, "allreduce")
            dtv = timed(wv, 20, 3, 1, device)
            ips_eager = 32 * 20 / dtv
            ids_eager = wv.step().clone()
            chk = grads_check(wv.named)                  # the eager step's gradients: checksum + finiteness
            chk["ids_checksum"] = int(ids_eager.sum())
            leg = {"workload": names["vit"].replace(f"bs={args.batch}", "bs=32"), "images_per_s": round(ips_eager, 2),
                   "ms_per_step": round(dtv / 20 * 1e3, 3), "launch": "eager", "step_check": chk,
                   "roofline": roofline(wv, "vit", ips_eager, gflop_per_image("vit"))}
            extra["vit_leg"] = leg                       # (the eager result stands whatever happens to the capture below)
            # the same step as one hipGraph: ~1000 launches per 49 ms are close to the host's launch rate in eager mode, which made the
            # eager number swing by 20 % between runs - the replayed graph measures the GPU
            ipsv = ips_eager
            del wv
            torch.cuda.empty_cache()
            try:
                import subprocess
                cmd = [sys.executable, os.path.abspath(__file__), "--vit-graph-leg"] + (["--no-multi"] if args.no_multi else [])
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
                lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode != 0 or not lines:
                    leg["graph_capture_failed"] = f"child rc={r.returncode}: {r.stderr[-200:]}"
                else:
                    gl = json.loads(lines[-1])
                    if "images_per_s" in gl:
                        ipsv = gl["images_per_s"]
                        chk["graph_replay_equals_eager"] = gl["graph_replay_equals_eager"]
                        leg.update(images_per_s=ipsv, ms_per_step=gl["ms_per_step"], launch="one hipGraph per step (replay, child process)",
                                   eager_images_per_s=round(ips_eager, 2), eager_ms_per_step=round(dtv / 20 * 1e3, 3))
                        leg["roofline"]["whole_step_frac"] = round(ipsv * gflop_per_image("vit") / 1e3 / PEAK_BF16_TFLOPS, 4)
                    else:
                        leg.update(gl)
            except Exception as e:
                leg["graph_capture_failed"] = repr(e)[:200]
            extra["vit_leg"] = leg
            note(f"vit leg: {ipsv:.1f} images/s ({leg['launch']}; eager {ips_eager:.1f}); check {chk}")
        except Exception as e:
            extra["vit_leg"] = {"error": repr(e)[:200]}
    if cu is not None:                                   # leave the CU-masked stream: restore the persistent-grid budget, destroy the stream
        cu.__exit__(None, None, None)
        cu.close()
    if extra:
        out["extra"] = extra
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        note("timing the CPU oracle on the host cores ...")
        out["cpu_baseline"] = cpu_baseline(seq=args.seq, workload=args.workload)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if check is not None and check.get("finite") is False:
        raise SystemExit("[bench] the timed step produced a non-finite loss / gradient: the throughput above is not a valid result")


if __name__ == "__main__":
    main()
