"""libra_rank_outer_wgrad at the Libra-11B shape (N = 16384 tokens, C = 4096): us per call and algorithmic TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd import kernels as K
N, C = 16384, 4096
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, C, device="cuda", generator=g).to(torch.bfloat16)
wide = torch.randn(N, 12352, device="cuda", generator=g).to(torch.bfloat16)
flag = torch.zeros(8, 2048, dtype=torch.uint8); flag[:, 1:579] = 1
flag = flag.reshape(-1).cuda()
for nc, tr in ((8, True), (16, False)):
    coef = wide[:, 12288:12288 + nc]
    fn = lambda: K.rank_outer_wgrad(x, coef, flag, transpose_out=tr)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"ncoef={nc} transpose={tr}: {us:.1f} us  {N * C * 2 / us / 1e6:.2f} TB/s of x", flush=True)
