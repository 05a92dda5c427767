"""colsum (bias gradient) at the ViT-L shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd import kernels as K
for cols in (1024, 3072, 4096):
    x = torch.randn(18464, cols, device="cuda").to(torch.bfloat16)
    out = torch.zeros(cols, device="cuda")
    for _ in range(3): K.colsum(x, out)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(20): K.colsum(x, out)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    ref = x.float().sum(0) * 23
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f"colsum 18464 x {cols}: {us:.1f} us  {x.numel() * 2 / us / 1e6:.2f} TB/s  rel err {err:.2e}")
