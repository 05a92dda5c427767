"""Cycle stamps of workgroup 0 of the bridge-attention forward (library built with -DLIBRA_ATTN_DBG=128[+...]): per wave and unit
[0] SM start [1] softmax done [2] staging wait done [3] barrier passed (M start) [4] MFMAs issued [5] staging wait done; the next
unit's [0] follows the M phase's closing barrier.  Prints the mean duration of each segment per wave over the wave's computed units."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd import kernels as K
B, S, H = 8, 2048, 32
N, D = B * S, H * 128
g = torch.Generator(device="cuda").manual_seed(0)
q, ks, kc, vs, vc = [torch.randn(N, D, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16) for _ in range(5)]
flag = torch.zeros(B, S, dtype=torch.uint8); flag[:, 1:579] = 1
flag = flag.reshape(N).cuda()
lens = torch.full((B,), S, dtype=torch.int32).cuda()
o_lo = torch.zeros_like(q)
for _ in range(3):
    K.bridge_attn_fwd(q, ks, kc, vs, vc, flag, lens, B, S, H, 128 ** -0.5, need_lse=True, out_lo=o_lo)
torch.cuda.synchronize()
w = o_lo.view(torch.int32).reshape(-1)[:2064].cpu().numpy().astype("int64") & 0xffffffff
U = int(w[2048]); Uw = [int(x) for x in w[2049:2057]]
print("units", U, "per-wave computed units", Uw)
names = ["softmax", "wait(SM)", "barrier(SM)", "M phase", "wait(M)", "barrier(M)"]
for wave in range(8):
    st = w[wave * 256: wave * 256 + 256]
    n = min(Uw[wave], 40)
    seg = [[] for _ in range(6)]
    for u in range(1, n - 1):                     # steady units
        a = st[6 * u: 6 * u + 7]
        for i in range(6):
            seg[i].append(int((a[i + 1] - a[i]) & 0xffffffff))
    if seg[0]:
        tot = sum(sum(x) / len(x) for x in seg)
        print(f"wave {wave}: " + "  ".join(f"{nm} {sum(x)/len(x):7.0f}" for nm, x in zip(names, seg)) + f"   | unit {tot:7.0f}")
