"""Cycle stamps of one heavy item (key block 5, "same" variant) of the dK/dV pass (library built with -DLIBRA_DKV_DBG=128): per wave
and unit [0] V phase start [1] arithmetic done [2] staging wait done [3] barrier passed (M start) [4] MFMAs issued [5] staging wait
done; the next unit's [0] follows the M phase's closing barrier.  Mean duration of each segment over the wave's steady units."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd import kernels as K
B, S, H = 8, 2048, 32
N, D = B * S, H * 128
g = torch.Generator(device="cuda").manual_seed(0)
q, ks, kc, vs, vc, do = [torch.randn(N, D, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16) for _ in range(6)]
flag = torch.zeros(B, S, dtype=torch.uint8); flag[:, 1:579] = 1
flag = flag.reshape(N).cuda()
lens = torch.full((B,), S, dtype=torch.int32).cuda()
sc = 128 ** -0.5
o, lse = K.bridge_attn_fwd(q, ks, kc, vs, vc, flag, lens, B, S, H, sc, need_lse=True)
for _ in range(3):
    dq, dks, dkc, dvs, dvc = K.bridge_attn_bwd(q, ks, kc, vs, vc, o, do, flag, lens, lse, B, S, H, sc)
torch.cuda.synchronize()
w = dks.view(torch.int32).reshape(-1)[:2064].cpu().numpy().astype("int64") & 0xffffffff
U = int(w[2048]); Uw = [int(x) for x in w[2049:2057]]
print("units", U, "per-wave computed units", Uw, "item cycles", int(w[2058]))
names = ["arith", "wait(V)", "barrier(V)", "M phase", "wait(M)", "barrier(M)"]
for wave in range(8):
    st = w[wave * 256: wave * 256 + 256]
    n = min(Uw[wave], 40)
    seg = [[] for _ in range(6)]
    for u in range(2, n - 1):                     # steady units
        a = st[6 * u: 6 * u + 7]
        for i in range(6):
            seg[i].append(int((a[i + 1] - a[i]) & 0xffffffff))
    if seg[0]:
        tot = sum(sum(x) / len(x) for x in seg)
        print(f"wave {wave} ({'dV' if wave < 4 else 'dK'}): " + "  ".join(f"{nm} {sum(x)/len(x):7.0f}" for nm, x in zip(names, seg)) + f"   | unit {tot:7.0f}")
    first = st[0:7]
    print(f"        first stamps since unit 0 start: {[int((first[i]-first[0]) & 0xffffffff) for i in range(7)]}")
