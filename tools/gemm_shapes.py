"""Per-shape GEMM time inside one step of a bench workload (HIP events per launch).
usage: python tools/gemm_shapes.py [vit|libra]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from libra_amd import kernels as K

wl = sys.argv[1] if len(sys.argv) > 1 else "libra"
dev = torch.device("cuda", 0)
if wl == "vit":
    step = bench.make_vit(dev, 32, 1, "allreduce").step
else:
    step = bench.make_bridge(dev, 8, 2048, 1, "allreduce").step
recs = []
mods = [m for m in sys.modules.values() if m is not None and getattr(m, "__name__", "").startswith("libra_amd")]
orig = K.gemm_nt


def wrapped(a, b, *args, **kw):
    a_t, b_t = kw.get("a_t", False), kw.get("b_t", False)
    k = kw.get("k") or (a.shape[0] if a_t else a.shape[1])
    m = kw["a_rows"].numel() if kw.get("a_rows") is not None else (a.shape[1] if a_t else a.shape[0])
    n = b.shape[1] if b_t else b.shape[0]
    tag = ("T" if a_t else "N") + ("T" if b_t else "N") + ("+gather" if kw.get("a_rows") is not None else "") + \
          ("+scatter" if kw.get("c_rows") is not None else "")
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = orig(a, b, *args, **kw)
    e.record()
    recs.append(((m, n, k, tag), s, e))
    return r


for _ in range(2):
    step()
K.gemm_nt = wrapped
for m in mods:                       # modules that did `from .kernels import gemm_nt` or use K.gemm_nt
    if getattr(m, "gemm_nt", None) is orig:
        m.gemm_nt = wrapped
step()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for key, s, e in recs:
    a = agg[key]; a[0] += 1; a[1] += s.elapsed_time(e)
tot = sum(v[1] for v in agg.values())
out = open(os.path.join(ROOT, "gpurun_out", f"gemm_shapes_{wl}.txt"), "w")
_p = print
def print(*a, **k):
    _p(*a, **k); _p(*a, **k, file=out)
print(f"{len(recs)} GEMM launches, {tot:.1f} ms")
for (m, n, k, tag), (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tf = 2.0 * m * n * k * c / ms / 1e9
    print(f"M={m:6d} N={n:6d} K={k:6d} {tag:12s} x{c:4d}  {ms:8.2f} ms  {ms / c * 1e3:8.1f} us/launch  {tf:7.1f} TF  {100 * ms / tot:5.1f}%")
