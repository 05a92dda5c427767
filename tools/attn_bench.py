"""Bridge-attention kernels at the Libra-11B decoder shape (B=8, S=2048, H=32, d=128, one 578-token image span per
sequence): time forward / backward and print algorithmic TFLOP/s (causal-minimal: 2*2*S^2/2*d per head forward)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd import kernels as K

B, S, H = int(os.environ.get("ATTN_B", 8)), int(os.environ.get("ATTN_S", 2048)), 32
which = sys.argv[1] if len(sys.argv) > 1 else "all"
N, D = B * S, H * 128
g = torch.Generator(device="cuda").manual_seed(0)
q, ks, kc, vs, vc, do = [torch.randn(N, D, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16) for _ in range(6)]
flag = torch.zeros(B, S, dtype=torch.uint8); flag[:, 1:min(579, S - 1)] = 1
flag = flag.reshape(N).cuda()
lens = torch.full((B,), S, dtype=torch.int32).cuda()
sc = 128 ** -0.5
fl_fwd = B * H * 2 * 2 * (S * (S + 1) / 2) * 128


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {}
o_lo = torch.empty_like(q) if os.environ.get("ATTN_LO", "1") == "1" else None      # the training step asks for the residual too
o, lse = K.bridge_attn_fwd(q, ks, kc, vs, vc, flag, lens, B, S, H, sc, need_lse=True, out_lo=o_lo)
if which in ("all", "fwd"):
    ms = timeit(lambda: K.bridge_attn_fwd(q, ks, kc, vs, vc, flag, lens, B, S, H, sc, need_lse=True, out_lo=o_lo))
    res["fwd_ms"] = round(ms, 3); res["fwd_TF"] = round(fl_fwd / ms / 1e9, 1)
if which in ("all", "bwd"):
    ms = timeit(lambda: K.bridge_attn_bwd(q, ks, kc, vs, vc, o, do, flag, lens, lse, B, S, H, sc, out_lo=o_lo))
    res["bwd_ms"] = round(ms, 3); res["bwd_TF"] = round(2.5 * fl_fwd / ms / 1e9, 1)
print(json.dumps(res))
