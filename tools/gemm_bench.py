"""Per-shape timing of the bf16 NT GEMM on the shapes of the ViT-L hot path (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd import kernels as K

SHAPES = [("qkv fwd", 18464, 3072, 1024), ("proj fwd", 18464, 1024, 1024), ("fc1 fwd", 18464, 4096, 1024),
          ("fc2 fwd", 18464, 1024, 4096), ("wgrad qkv", 3072, 1024, 18496), ("wgrad proj", 1024, 1024, 18496),
          ("wgrad fc1", 4096, 1024, 18496), ("wgrad fc2", 1024, 4096, 18496), ("quant_conv", 18432, 512, 2048),
          ("patch", 18432, 1024, 640), ("sq4k", 4096, 4096, 4096), ("sq8k", 8192, 8192, 8192)]
it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
TSHAPES = [("dgrad fc2 bT", 18464, 4096, 1024, False, True), ("dgrad fc1 bT", 18464, 1024, 4096, False, True),
           ("dgrad proj bT", 18464, 1024, 1024, False, True), ("dgrad qkv bT", 18464, 1024, 3072, False, True),
           ("wgrad qkv TT", 3072, 1024, 18496, True, True), ("wgrad proj TT", 1024, 1024, 18496, True, True),
           ("wgrad fc1 TT", 4096, 1024, 18496, True, True), ("wgrad fc2 TT", 1024, 4096, 18496, True, True),
           ("sq4k TT", 4096, 4096, 4096, True, True), ("sq4k bT", 4096, 4096, 4096, False, True)]
for name, M, N, Kd, *tt in [s + (False, False) for s in SHAPES] + TSHAPES:
    a_t, b_t = tt
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
    print(f"{name:14s} M={M:6d} N={N:5d} K={Kd:6d}  {us:9.1f} us  {2.0*M*N*Kd/us/1e6:8.1f} TFLOP/s", flush=True)
