"""Time one bf16 GEMM shape: python tools/gemm_one.py M N K [a_t b_t] [iters]   (GEMM_TILE=1|2|3 = 128 / 256 / W tile structure;
GEMM_ZERO=1: zero-filled operands - the data-dependent power draw of the MFMA datapath is then near its minimum, rule 25 of the guide)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd import kernels as K
M, N, Kd = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
a_t = len(sys.argv) > 4 and sys.argv[4] == "1"
b_t = len(sys.argv) > 5 and sys.argv[5] == "1"
it = int(sys.argv[6]) if len(sys.argv) > 6 else 10
TILE = int(os.environ.get("GEMM_TILE", 0))
a = torch.randn((Kd, M) if a_t else (M, Kd), device="cuda").to(torch.bfloat16)
b = torch.randn((Kd, N) if b_t else (N, Kd), device="cuda").to(torch.bfloat16)
if os.environ.get("GEMM_ZERO") == "1":
    a.zero_(); b.zero_()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    K.gemm_nt(a, b, out=out, a_t=a_t, b_t=b_t, tile=TILE)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); s.record()
for _ in range(it):
    K.gemm_nt(a, b, out=out, a_t=a_t, b_t=b_t, tile=TILE)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / it * 1e3
print(f"M={M} N={N} K={Kd} a_t={a_t} b_t={b_t}: {us:.1f} us  {2.0*M*N*Kd/us/1e6:.1f} TFLOP/s", flush=True)
