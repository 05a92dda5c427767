"""Where the step's glue kernels come from: one headline step under torch.profiler with Python stacks; device time of the copy /
fill / small-reduction kernels grouped by the innermost libra_amd (or bench.py) source line that launched them."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

import traceback
from libra_amd import kernels as K

own = []          # (site, name, start event, end event) of the library's own glue launches (ctypes: invisible to the profiler's stacks)


def wrap(name):
    fn = getattr(K, name)

    def w_(*a, **kw):
        if not own_on[0]:
            return fn(*a, **kw)
        st = [f for f in traceback.extract_stack()[:-1] if "libra_amd" in f.filename and "kernels.py" not in f.filename]
        site = f"{os.path.basename(st[-1].filename)}:{st[-1].lineno} {st[-1].line[:70]}" if st else "?"
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record(); r = fn(*a, **kw); e_.record()
        own.append((site, name, s_, e_))
        return r
    setattr(K, name, w_)


own_on = [False]
for n_ in ("copy_rows", "gather_rows", "f32_to_bf16", "rank_outer_wgrad", "rmsnorm_routed_wgrad", "add_", "colsum"):
    wrap(n_)

dev = torch.device("cuda", 0)
w = bench.make_bridge(dev, 8, 2048, 1, "allreduce")
for _ in range(2):
    w.step()
torch.cuda.synchronize()
own_on[0] = True
w.step()
torch.cuda.synchronize()
own_on[0] = False
agg = collections.defaultdict(lambda: [0.0, 0])
for site, name, s_, e_ in own:
    a = agg[(name, site)]; a[0] += s_.elapsed_time(e_); a[1] += 1
print(f"library glue launches (HIP events): {sum(v[0] for v in agg.values()):.2f} ms in {sum(v[1] for v in agg.values())} launches")
for (name, site), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"{t:7.3f} ms {n:5d} x {name:22s} {site}")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    w.step()
    torch.cuda.synchronize()
pat = ("copy", "Copy", "fill", "Fill", "copy_rows", "elementwise", "index", "cat", "f32_to_bf16", "splitk_reduce")
by_site = collections.defaultdict(lambda: [0.0, 0, collections.Counter()])
ev = prof.events()
for e in ev:
    if e.device_type is None or str(e.device_type).endswith("CPU"):
        continue
for e in ev:
    kt = sum(k.duration for k in e.kernels) if getattr(e, "kernels", None) else 0.0
    if not kt or not e.stack:
        continue
    names = [k.name for k in e.kernels]
    if not any(any(p in n for p in pat) for n in names):
        continue
    site = next((s for s in e.stack if "libra_amd" in s or "bench.py" in s), e.stack[0] if e.stack else "?")
    b = by_site[site.strip()]
    b[0] += kt; b[1] += len(names)
    for n in names:
        b[2][n.split("(")[0][-60:]] += 1
tot = sum(v[0] for v in by_site.values())
print(f"glue kernel time attributed: {tot / 1e3:.2f} ms")
for site, (t, n, kn) in sorted(by_site.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{t / 1e3:7.3f} ms {n:5d} launches  {site[-110:]}   [{', '.join(f'{k} x{c}' for k, c in kn.most_common(2))}]")
