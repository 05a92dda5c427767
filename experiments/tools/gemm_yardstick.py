"""Yardstick only (never the product path): this build's bf16 GEMM against the vendor library (torch.matmul -> hipBLASLt / rocBLAS)
on the decoder's dominant shapes, each timed over a sustained ~0.4 s burst so both run at the chip's steady-state clock."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd import kernels as K

SHAPES = [("gate|up fwd", 11760, 22016, 4096, False, False), ("gate|up dgrad", 11760, 4096, 22016, False, True),
          ("qkv fwd", 11760, 12352, 4096, False, False), ("down fwd", 11760, 4096, 11008, False, False),
          ("down dgrad", 11760, 11008, 4096, False, True),
          ("o fwd", 11760, 4096, 4096, False, False), ("wgrad vis", 11008, 2752, 4672, True, True),
          ("vis A", 4624, 5504, 4096, False, False), ("vis B K=1024", 4624, 11008, 1024, False, False),
          ("sq8k", 8192, 8192, 8192, False, False), ("sq8k bT", 8192, 8192, 8192, False, True)]


ONLY_OURS = len(sys.argv) > 1 and sys.argv[1] == "ours"


def timed(fn, burst_s=0.4):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record(); fn(); e.record(); torch.cuda.synchronize()
    it = max(5, int(burst_s * 1e3 / max(s.elapsed_time(e), 1e-3)))
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


tot = [0.0, 0.0]
for name, M, N, Kd, a_t, b_t in SHAPES:
    a = torch.randn((Kd, M) if a_t else (M, Kd), device="cuda").to(torch.bfloat16)
    b = torch.randn((Kd, N) if b_t else (N, Kd), device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ours = timed(lambda: K.gemm_nt(a, b, out=out, a_t=a_t, b_t=b_t))
    A = a.t() if a_t else a
    Bm = b if b_t else b.t()
    lib = timed(lambda: torch.matmul(A, Bm, out=out)) if ONLY_OURS is False else ours
    tot[0] += ours; tot[1] += lib
    fl = 2.0 * M * N * Kd / 1e6
    print(f"{name:14s} M={M:6d} N={N:5d} K={Kd:6d}  ours {ours:8.1f} us {fl/ours:7.1f} TF | vendor {lib:8.1f} us {fl/lib:7.1f} TF | ours/vendor time {ours/lib:.3f}", flush=True)
print(f"sum ours {tot[0]:.1f} us, vendor {tot[1]:.1f} us")
