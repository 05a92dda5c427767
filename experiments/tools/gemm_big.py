"""bf16 NT GEMM on the decoder's dominant shapes (Libra-11B, 8 x 2048 tokens, 11760 text rows / 4624 vision rows) + squares."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd import kernels as K

SHAPES = [("gate|up fwd", 11760, 22016, 4096, False, False), ("gate|up dgrad", 11760, 4096, 22016, False, True),
          ("qkv fwd", 11760, 12352, 4096, False, False), ("down fwd", 11760, 4096, 11008, False, False),
          ("o fwd", 11760, 4096, 4096, False, False), ("wgrad vis", 11008, 2752, 4672, True, True),
          ("vis A", 4624, 5504, 4096, False, False), ("vis B K=1024", 4624, 11008, 1024, False, False),
          ("vis dgrad", 4624, 1024, 4096, False, True), ("sq8k", 8192, 8192, 8192, False, False)]
it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tot = 0.0
for name, M, N, Kd, a_t, b_t in SHAPES:
    a = torch.randn((Kd, M) if a_t else (M, Kd), device="cuda").to(torch.bfloat16)
    b = torch.randn((Kd, N) if b_t else (N, Kd), device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        K.gemm_nt(a, b, out=out, a_t=a_t, b_t=b_t)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(it):
        K.gemm_nt(a, b, out=out, a_t=a_t, b_t=b_t)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / it * 1e3
    tot += us
    print(f"{name:14s} M={M:6d} N={N:5d} K={Kd:6d}  {us:9.1f} us  {2.0*M*N*Kd/us/1e6:8.1f} TFLOP/s", flush=True)
print(f"sum {tot:.1f} us")
