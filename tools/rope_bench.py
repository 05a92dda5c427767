"""rope_bridge / rope_bridge_bwd at the Libra-11B shape (N = 8 x 2048 tokens, H = 32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd import kernels as K
from oracle import libra_oracle as LO    # (rope tables only; this is a bench tool, not the product path)
B, S, H = 8, 2048, 32
N, D = B * S, H * 128
bf = torch.bfloat16
qkv = torch.randn(N, 3 * D, device="cuda").to(bf)
tb = torch.zeros(N, 64, device="cuda", dtype=bf); tb[:, :16] = torch.randn(N, 16, device="cuda").to(bf)
w = [torch.randn(D, 8, device="cuda").to(bf) * 0.3 for _ in range(4)]
flag = torch.zeros(B, S, dtype=torch.uint8); flag[:, 1:579] = 1
flag = flag.reshape(N).cuda()
cosf, sinf = LO.rope_tables(128, S)
cos, sin = cosf.to(bf).cuda(), sinf.to(bf).cuda()
def run():
    K.rope_bridge(qkv, tb, *w, flag, cos, sin, S, H)
for _ in range(3): run()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); s.record()
for _ in range(10): run()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 100
print(f"rope_bridge: {us:.1f} us  ({(qkv.numel() * 2 * (1 + 2 / 3) + 2 * N * D * 2) / us / 1e6:.2f} TB/s algorithmic)")


def timeit(fn, what, nbytes):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    print(f"{what}: {us:.1f} us  ({nbytes / us / 1e6:.2f} TB/s algorithmic)")


g = [torch.randn(N, D, device="cuda").to(bf) for _ in range(5)]
dqkvt = torch.empty(N, 3 * D + 64, device="cuda", dtype=bf)
dqkv, dtb = dqkvt[:, :3 * D], dqkvt[:, 3 * D:]
dkb = torch.empty(N, D, device="cuda", dtype=bf)
nb = 9 * N * D * 2
timeit(lambda: K.rope_bridge_bwd(*g, cos, sin, S, H, dqkv, dkb), "rope_bridge_bwd (no dtb)", nb)
timeit(lambda: K.rope_bridge_bwd(*g, cos, sin, S, H, dqkv, dkb, bridge_b=tuple(t.t().contiguous() for t in w), flag=flag, dtb=dtb), "rope_bridge_bwd (+dtb)", nb)

# routed RMSNorm backward at the same token count (H = 4096 channels, residual gradient fused)
Dm = 4096
dy, x, dres = [torch.randn(N, Dm, device="cuda").to(bf) for _ in range(3)]
wl, wv = torch.randn(Dm, device="cuda").to(bf), torch.randn(Dm, device="cuda").to(bf)
rstd = torch.rand(N, device="cuda") + 0.5
dx = torch.empty(N, Dm, device="cuda", dtype=bf)
timeit(lambda: K.rmsnorm_routed_bwd(dy, x, wl, wv, flag, rstd, dres=dres, out=dx), "rmsnorm_routed_bwd (+dres)", 4 * N * Dm * 2)
timeit(lambda: K.rmsnorm_routed_bwd(dy, x, wl, wv, flag, rstd, out=dx), "rmsnorm_routed_bwd", 3 * N * Dm * 2)
dl, dv = torch.zeros(Dm, device="cuda"), torch.zeros(Dm, device="cuda")
timeit(lambda: K.rmsnorm_routed_wgrad(dy, x, rstd, flag, dl, dv), "rmsnorm_routed_wgrad", 2 * N * Dm * 2)
y = torch.empty(N, Dm, device="cuda", dtype=bf)
timeit(lambda: K.rmsnorm_routed(x, wl, wv, flag, 1e-6, out=y), "rmsnorm_routed fwd", 2 * N * Dm * 2)
