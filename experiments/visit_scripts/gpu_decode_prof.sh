#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_decode -- python $R/tools/decode_bench.py 8 1024 32 > $R/gpurun_out/prof_decode.log 2>&1 ); echo "prof rc=$?"
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_decode/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("gpurun_out/decode_kernel_stats.csv", "w", newline="") as o:
    w = csv.DictWriter(o, fieldnames=rows[0].keys()); w.writeheader()
    for r in rows:
        r["Name"] = r["Name"][:120]; w.writerow(r)
steps = 36
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls'])/steps:7.1f}/step {float(r['TotalDurationNs'])/1e6/steps:7.3f} ms/step avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
rm -rf gpurun_out/prof_decode
tail -1 gpurun_out/prof_decode.log
