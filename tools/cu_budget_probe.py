"""What reserving compute units for RCCL costs the compute stream (1 GPU, no collective involved): the step's biggest GEMM and the
bridge attention forward / backward on a CU-masked stream (libra_stream_create_cu_reserved + libra_set_cu_budget) for several
reserve sizes.  One JSON line per reserve size."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd import kernels as K

dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
a, b = rnd(11760, 4096), rnd(22016, 4096)
B, S, H = 8, 2048, 32
q, ks, kc, vs, vc, do = [rnd(B * S, H * 128) for _ in range(6)]
flag = torch.zeros(B, S, dtype=torch.uint8); flag[:, 1:579] = 1
flag = flag.reshape(-1).cuda()
lens = torch.full((B,), S, dtype=torch.int32).cuda()
sc = 128 ** -0.5


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(n):
        fn()
    e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run():
    o, lse = K.bridge_attn_fwd(q, ks, kc, vs, vc, flag, lens, B, S, H, sc, need_lse=True)
    return {"gemm_11760x22016x4096_ms": round(timeit(lambda: K.gemm_nt(a, b)), 4),
            "attn_fwd_ms": round(timeit(lambda: K.bridge_attn_fwd(q, ks, kc, vs, vc, flag, lens, B, S, H, sc, need_lse=True)), 4),
            "attn_bwd_ms": round(timeit(lambda: K.bridge_attn_bwd(q, ks, kc, vs, vc, o, do, flag, lens, lse, B, S, H, sc)), 4)}


print(json.dumps({"reserve": 0, "cus": K.cu_count(), **run()}), flush=True)
for reserve in (8, 16, 32):
    rs = K.ReservedCUStream(reserve)
    with rs:
        r = run()
    rs.close()
    print(json.dumps({"reserve": reserve, "cus": rs.cus, **r}), flush=True)
print(json.dumps({"reserve": 0, "again": True, **run()}), flush=True)
