"""Cached generation at the Libra-11B shape: prefill of B prompts (BOS + one 578-token image + text) and N decode steps.
Reports ms per decode step (all B sequences advance by one token) and the HBM floor of reading the weights once."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd.libra import LibraConfig, LibraForCausalLM

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda")
cfg = LibraConfig()
with torch.device(dev):
    m = LibraForCausalLM(cfg)
m = m.to(torch.bfloat16).eval()
with torch.no_grad():
    for n, p in m.named_parameters():
        if "bridge" in n and n.endswith("weight_B"):
            p.normal_(0, 0.02)
V, L = cfg.vocab_size, cfg.max_vision_token_length
g = torch.Generator().manual_seed(0)
ids = torch.randint(3, V, (2, B, P), generator=g)
ids[0, :, 0] = 1
ids[:, :, 1] = V + 512
ids[:, :, 2:L] = V + torch.randint(0, 512, (2, B, L - 2), generator=g)
ids[:, :, L] = V + 513
ids[1, :, L + 1:] = ids[0, :, L + 1:]
ids[1, :, 0] = 1
vi = torch.full((B, P), L, dtype=torch.long)
vi[:, 1:L + 1] = torch.arange(L)
sig = torch.zeros(B, P, cfg.contiguous_signal_size, dtype=torch.bfloat16)
sig[:, 2:L] = torch.randn(B, L - 2, cfg.contiguous_signal_size, generator=g).to(torch.bfloat16)
ids, vi, sig = ids.to(dev), vi.to(dev), sig.to(dev)
with torch.no_grad():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = m(input_ids=ids, vision_indices=vi, contiguous_signal=sig, use_cache=True)
    torch.cuda.synchronize(); t_pre = time.perf_counter() - t0
    past = out.past_key_values
    nxt = out.logits[0, :, -1, :V].argmax(-1)
    tok = torch.stack([nxt, nxt])[:, :, None]
    vin = torch.full((B, 1), L, dtype=torch.long, device=dev)
    for warm in range(3):
        out = m(input_ids=tok, vision_indices=vin, past_key_values=past, use_cache=True)
        tok = out.logits[0, :, -1, :V].argmax(-1); tok = torch.stack([tok, tok])[:, :, None]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(STEPS):
        out = m(input_ids=tok, vision_indices=vin, past_key_values=past, use_cache=True)
        tok = out.logits[0, :, -1, :V].argmax(-1); tok = torch.stack([tok, tok])[:, :, None]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / STEPS
nparam = sum(p.numel() for n, p in m.named_parameters() if "vision" not in n) * 2
print(f"B={B} prompt={P}: prefill {t_pre * 1e3:.1f} ms (first call, includes operand packing); decode {dt * 1e3:.2f} ms/step = "
      f"{B / dt:.0f} tokens/s; text-weight bytes per step {nparam / 1e9:.1f} GB -> {nparam / dt / 1e12:.2f} TB/s effective; "
      f"cache length {past.get_seq_length()}")
