"""The vision-branch GEMM shapes of the Libra-11B step (M = 4624 vision rows; the stream-K candidates) + two text shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd import kernels as K

SHAPES = [("low-rank wgrad", 4096, 1024, 4672, True, True, 1), ("vis down dgrad x2", 4624, 2752, 11008, False, True, 2),
          ("vis wgrad", 11008, 2752, 4672, True, True, 1), ("vis B x2", 4624, 11008, 2752, False, False, 2),
          ("up wgrad", 5504, 4096, 4672, True, True, 1), ("up dgrad", 4624, 4096, 5504, False, True, 1),
          ("vis A", 4624, 5504, 4096, False, False, 1), ("qkv A dgrad", 4624, 4096, 3136, False, True, 1),
          ("B K=1024 x3", 4624, 4096, 1024, False, False, 3), ("B K=1024", 4624, 4096, 1024, False, False, 1),
          ("A N=1024 dgrad", 4624, 1024, 4096, False, True, 1), ("A N=1024", 4624, 1024, 4096, False, False, 1),
          ("down B dgrad", 4624, 11008, 1024, False, True, 1), ("down A", 4624, 1024, 11008, False, False, 1),
          ("qkv A", 4624, 3136, 4096, False, False, 1), ("text o", 11760, 4096, 4096, False, False, 1),
          ("text gate|up", 11760, 22016, 4096, False, False, 1)]
tot = 0.0
for name, M, N, Kd, a_t, b_t, G in SHAPES:
    As = [torch.randn((Kd, M) if a_t else (M, Kd), device="cuda").to(torch.bfloat16) for _ in range(G)]
    Bs = [torch.randn((Kd, N) if b_t else (N, Kd), device="cuda").to(torch.bfloat16) for _ in range(G)]
    outs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(G)]
    fn = (lambda: K.gemm_nt(As[0], Bs[0], out=outs[0], a_t=a_t, b_t=b_t)) if G == 1 else (lambda: K.gemm_nt_grouped(As, Bs, outs, a_t=a_t, b_t=b_t))
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    it = 50
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / it * 1e3
    tot += us
    print(f"{name:18s} {G}x[{M:6d} {N:5d} {Kd:6d}]  {us:8.1f} us  {2.0*G*M*N*Kd/us/1e6:7.1f} TF", flush=True)
print(f"sum {tot:.1f} us")
