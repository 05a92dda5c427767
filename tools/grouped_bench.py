"""A/B inside one process: G separate GEMM launches vs one grouped launch (Libra-11B vision low-rank shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd import kernels as K
bf = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for (M, N, Kd, G, b_t) in [(4624, 4096, 1024, 3, False), (4624, 1024, 4096, 3, True), (4624, 11008, 2752, 2, False),
                           (4624, 2752, 11008, 2, True)]:
    a = [torch.randn(M, Kd, device="cuda").to(bf) for _ in range(G)]
    b = [torch.randn((Kd, N) if b_t else (N, Kd), device="cuda").to(bf) for _ in range(G)]
    o = [torch.empty(M, N, device="cuda", dtype=bf) for _ in range(G)]
    sep = lambda: [K.gemm_nt(a[g], b[g], out=o[g], b_t=b_t) for g in range(G)]
    grp = lambda: K.gemm_nt_grouped(a, b, o, b_t=b_t)
    r = []
    for _ in range(3):
        r.append((timeit(sep), timeit(grp)))
    fl = 2.0 * M * N * Kd * G
    print(f"M={M} N={N} K={Kd} G={G} b_t={b_t}: separate {min(x[0] for x in r):.1f} us ({fl / min(x[0] for x in r) / 1e6:.0f} TF)  "
          f"grouped {min(x[1] for x in r):.1f} us ({fl / min(x[1] for x in r) / 1e6:.0f} TF)", flush=True)
