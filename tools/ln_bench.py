"""LayerNorm forward / backward at the ViT bench shape (32 x 577 tokens, D = 1024)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd import kernels as K
M, D = 32 * 577, 1024
bf = torch.bfloat16
x, dy, dres = [torch.randn(M, D, device="cuda").to(bf) for _ in range(3)]
g, b = torch.randn(D, device="cuda").to(bf), torch.randn(D, device="cuda").to(bf)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timeit(fn, what, nbytes, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / it * 1e3
    print(f"{what}: {us:.1f} us  ({nbytes / us / 1e6:.2f} TB/s algorithmic)")


y = torch.empty(M, D, device="cuda", dtype=bf)
timeit(lambda: K.layernorm_fwd(x, g, b, 1e-5, save_stats=True, out=y), "layernorm_fwd", 2 * M * D * 2)
_, mean, rstd = K.layernorm_fwd(x, g, b, 1e-5, save_stats=True, out=y)
dg, db, dxs = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
dx = torch.empty(M, D, device="cuda", dtype=bf)
timeit(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, dgamma=dg, dbeta=db, out=dx), "layernorm_bwd (+dres)", 4 * M * D * 2)
timeit(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, dgamma=dg, dbeta=db, dxsum=dxs, out=dx), "layernorm_bwd (+dres, +dxsum)", 4 * M * D * 2)
