#!/bin/bash
# kernel timeline of the headline step: busy vs idle time on the stream (are there host-bound gaps?)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_bridge -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/trace_bridge.log 2>&1 ); echo "trace rc=$?"
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_bridge/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in csv.DictReader(open(f))]
rows.sort()
# steps: find bridge_attn_fwd occurrences; a step = 32 of them
idx = [i for i, r in enumerate(rows) if "bridge_attn_fwd" in r[2]]
print("kernels", len(rows), "attn fwd launches", len(idx))
# take the window from the first attn fwd of step 3 (index 64) to the first of step 4 (96)
for s in (2, 3, 4):
    if len(idx) < 32 * (s + 1) + 1: break
    a, b = idx[32 * s], idx[32 * (s + 1)]
    span = rows[b][0] - rows[a][0]
    busy = sum(e - st for st, e, _ in rows[a:b])
    gaps = sorted(((rows[i + 1][0] - rows[i][1], rows[i][2], rows[i + 1][2]) for i in range(a, b - 1)), reverse=True)
    print(f"step window {s}: span {span/1e6:.2f} ms, kernel busy {busy/1e6:.2f} ms, idle {(span-busy)/1e6:.2f} ms over {b-a} kernels; "
          f"gaps > 20us: {sum(1 for g in gaps if g[0] > 20000)} totalling {sum(g[0] for g in gaps if g[0] > 20000)/1e6:.2f} ms; "
          f"median gap {sorted(g[0] for g in gaps)[len(gaps)//2]/1e3:.1f} us")
    for g in gaps[:8]:
        print(f"    {g[0]/1e3:8.1f} us after {g[1]} before {g[2]}")
    if s == 3:                       # context of the three largest gaps: the kernels around them
        big = sorted(range(a, b - 1), key=lambda i: rows[i + 1][0] - rows[i][1], reverse=True)[:3]
        for i in big:
            print(f"  -- gap {(rows[i + 1][0] - rows[i][1]) / 1e3:.1f} us at kernel {i - a} of the window:")
            for j in range(max(a, i - 6), min(b, i + 5)):
                print(f"       {'>>' if j == i + 1 else '  '} +{(rows[j][0] - rows[a][0]) / 1e6:9.3f} ms  {(rows[j][1] - rows[j][0]) / 1e3:8.1f} us  {rows[j][2]}")
PY
rm -rf gpurun_out/trace_bridge
