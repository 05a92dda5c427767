"""Forward throughput of the routed decoder at BASELINE config 3 shape (Libra-11B: 32 layers, H=4096, B=8, S=2048,
one 578-token image per sequence), random-init bf16, on one MI355X.  Reports seq/s and algorithmic TFLOP/s
(25.44 TFLOP / sequence forward, causal-minimal attention)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd.libra import LibraConfig, LibraForCausalLM

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mode = sys.argv[3] if len(sys.argv) > 3 else "fwd"          # fwd | pretrain (frozen language, modeling_libra.py:1342-1346) | full
S = 2048
cfg = LibraConfig(num_hidden_layers=layers)
t0 = time.time()
with torch.device("cuda"):
    m = LibraForCausalLM(cfg)
m = m.to(torch.bfloat16).eval()
with torch.no_grad():       # bridges are zero-initialised upstream; make the bridge path numerically live
    for n, p in m.named_parameters():
        if "bridge" in n and n.endswith("weight_B"):
            p.normal_(0, 0.02)
print(f"built {sum(p.numel() for p in m.parameters())/1e9:.2f} B params in {time.time()-t0:.1f}s", flush=True)
g = torch.Generator().manual_seed(42)
V, L = 32000, 578
ids = torch.randint(3, V, (2, B, S), generator=g)
ids[:, :, 0] = 1
ids[:, :, 1] = V + 512
ids[:, :, 2:578] = V + torch.randint(0, 512, (2, B, 576), generator=g)
ids[:, :, 578] = V + 513
ids[1, :, 579:] = ids[0, :, 579:]
vi = torch.full((B, S), L, dtype=torch.long); vi[:, 1:579] = torch.arange(L)
am = torch.ones(B, S, dtype=torch.long)
sig = torch.zeros(B, S, 2048); sig[:, 2:578] = torch.randn(B, 576, 2048, generator=g)
labels = ids.clone(); labels[:, :, :2] = -100; labels[:, :, 579] = -100
ids, vi, am, sig, labels = ids.cuda(), vi.cuda(), am.cuda(), sig.to(torch.bfloat16).cuda(), labels.cuda()

if mode == "pretrain":
    for n, p in m.named_parameters():
        p.requires_grad_("vision" in n)
elif mode == "full":
    m.requires_grad_(True)
ntrain = sum(p.numel() for p in m.parameters() if p.requires_grad) if mode != "fwd" else 0

def step():
    if mode == "fwd":
        with torch.no_grad():
            return m(input_ids=ids, attention_mask=am, vision_indices=vi, contiguous_signal=sig, labels=labels).loss
    for p in m.parameters():
        p.grad = None
    loss = m(input_ids=ids, attention_mask=am, vision_indices=vi, contiguous_signal=sig, labels=labels).loss
    loss.backward()
    return loss.detach()
loss = step(); step()
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
per_layer = 782.65e9
fwd = B * (layers * per_layer + 0.39e12)
# backward: dgrad everywhere (1x fwd) + wgrad only for trainable weights.  Per layer and sequence: text GEMMs 595.0 G, vision
# GEMMs 153.3 G, attention 34.4 G (its backward is 2.5x: dQ, dK, dV + recompute is not credited)
if mode == "fwd":
    flops = fwd
else:
    text, vis, att, heads_t, heads_v = 595.0e9, 153.34e9, 34.4e9, 0.385e12, 0.0048e12
    wl = (text if mode == "full" else 0.0) + vis
    flops = fwd + B * (layers * (text + vis + 2.5 * att + wl) + heads_t + heads_v + (heads_t if mode == "full" else 0) + heads_v)
print(json.dumps({"workload": f"Libra decoder {mode}, {layers} layers, B={B}, S={S}", "trainable_params_B": round(ntrain / 1e9, 3),
                  "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1), "ms": round(dt*1e3, 2),
                  "seq_per_s": round(B/dt, 3), "algorithmic_TFLOPs": round(flops/dt/1e12, 1),
                  "frac_of_2.5PF": round(flops/dt/2.5e15, 4), "loss": float(loss)}), flush=True)
