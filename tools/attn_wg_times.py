"""Per-workgroup cycle records of the bridge-attention forward (library built with -DLIBRA_ATTN_DBG=256): prologue / loop / epilogue
cycles by query block, cycles per unit, and the busy time per CU (hardware id) against the kernel's span."""
import os, sys, collections
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
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    K.bridge_attn_fwd(q, ks, kc, vs, vc, flag, lens, B, S, H, 128 ** -0.5, need_lse=True, out_lo=o_lo)
e1.record(); torch.cuda.synchronize()
print(f"event-timed: {e0.elapsed_time(e1) / 10 * 1000:.1f} us per call (10 back-to-back calls)")
e0.record()
K.bridge_attn_fwd(q, ks, kc, vs, vc, flag, lens, B, S, H, 128 ** -0.5, need_lse=True, out_lo=o_lo)
e1.record(); torch.cuda.synchronize()
print(f"event-timed: {e0.elapsed_time(e1) * 1000:.1f} us for one call")
nblk = B * H * ((S + 255) // 256)
rec = o_lo.view(torch.int64).reshape(-1)[:nblk * 8].reshape(nblk, 8).cpu().numpy()
by_qt = collections.defaultdict(list)
real = []
cu = collections.defaultdict(list)
for r in rec:
    t0, t1, t2, t3, hw, xcc, qt, U = [int(x) for x in r]
    r0, r1 = (xcc >> 8) & ((1 << 55) - 1), (qt >> 8) & ((1 << 55) - 1)
    xcc &= 0xff; qt &= 0xff
    real.append((r0, r1, t3 - t0))
    by_qt[qt].append((t1 - t0, t2 - t1, t3 - t2, U))
    cu[(xcc, hw & 0xff00)].append((t0, t3))         # hw id without the wave slot bits
print("qt  units  prologue  loop  epilogue  cycles/unit   (mean over workgroups)")
for qt in sorted(by_qt):
    v = by_qt[qt]; n = len(v)
    pro, loop, epi, U = [sum(x[i] for x in v) / n for i in range(4)]
    print(f"{qt:2d} {U:6.1f} {pro:9.0f} {loop:9.0f} {epi:9.0f} {loop / max(U, 1):9.0f}")
# per XCD: the counter is only comparable inside one XCD; CU = (se, sh, cu) = HW_ID bits 15:8
for x in sorted(set(k[0] for k in cu)):
    t_min = min(s for k, v in cu.items() if k[0] == x for s, e in v)
    t_max = max(e for k, v in cu.items() if k[0] == x for s, e in v)
    cus = {k: v for k, v in cu.items() if k[0] == x}
    busy = sorted(sum(e - s for s, e in v) for v in cus.values())
    ends = sorted(max(e for s, e in v) - t_min for v in cus.values())
    nwg = sorted(len(v) for v in cus.values())
    gaps = []
    for v in cus.values():
        v = sorted(v)
        gaps += [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    print(f"XCD {x}: span {t_max - t_min} cycles, {len(cus)} CUs, workgroups per CU {nwg[0]}..{nwg[-1]}; busy per CU min {busy[0]} med {busy[len(busy)//2]} max {busy[-1]};"
          f" last end min {ends[0]} med {ends[len(ends)//2]} max {ends[-1]}; gap between workgroups med {sorted(gaps)[len(gaps)//2] if gaps else 0}")

r_min, r_max = min(x[0] for x in real), max(x[1] for x in real)
tick = sum(x[2] for x in real) / max(1, sum(x[1] - x[0] for x in real))
print(f"kernel span by the 100 MHz counter: {(r_max - r_min) / 100:.1f} us; cycle counter ticks per 100 MHz tick inside workgroups: {tick:.2f} -> core clock {tick / 10:.2f} GHz")

rec2 = o_lo.view(torch.int64).reshape(-1)[nblk * 8: nblk * 12].reshape(nblk, 4).cpu().numpy()
m = rec2.mean(0)
print(f"prologue anatomy (mean cycles): to the mask pass done {m[0]:.0f} | barrier {m[1]:.0f} | classification + barrier {m[2]:.0f} | tables, guess check, constants {m[3]:.0f} | then vmcnt(0) + barrier to the loop")
