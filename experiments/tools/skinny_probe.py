"""The decode step's four weight-streaming GEMMs at M = 8 rows, each over a POOL of distinct weight buffers (2.9 GB per shape at most:
the Infinity Cache cannot hold them, as in a real step where every layer has its own weights).  Reports us and TB/s of weight bytes.
(profiles/r03_skinny_probe.txt also holds timing-only builds selected by LIBRA_SKINNY_DBG - 1 = no A loads, 2 = no dot products - that
were not kept in the kernel; with the shipped library every run is the dbg=0 line.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd import kernels as K
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tot = 0.0
for name, N, Kd in (("qkv", 12352, 4096), ("o", 4096, 4096), ("gate|up", 22016, 4096), ("down", 4096, 11008)):
    pool = max(4, min(32, int(3.0e9 / (N * Kd * 2))))
    ws = [torch.randn(N, Kd, device="cuda", dtype=torch.bfloat16) for _ in range(pool)]
    a = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
    res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for w in ws[:3]:
        K.gemm_nt(a, w, out=out, resid=res)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    reps = 4
    for _ in range(reps):
        for w in ws:
            K.gemm_nt(a, w, out=out, resid=res)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / (reps * pool) * 1e3
    tot += us
    print(f"dbg={os.environ.get('LIBRA_SKINNY_DBG','0')} {name:8s} N={N:6d} K={Kd:6d} pool={pool:2d} {us:7.1f} us  {N*Kd*2/us/1e6:5.2f} TB/s", flush=True)
    del ws
print(f"dbg={os.environ.get('LIBRA_SKINNY_DBG','0')} sum {tot:.1f} us per layer")
