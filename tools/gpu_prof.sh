#!/bin/bash
# rocprofv3 kernel stats of the headline step (4 steps + 2 warm-up; the CSV's kernel names truncated to 120 characters)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bridge -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/prof_bridge.log 2>&1 ); echo "prof rc=$?"
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_bridge/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("gpurun_out/prof_bridge_kernel_stats.csv", "w", newline="") as o:
    w = csv.DictWriter(o, fieldnames=rows[0].keys()); w.writeheader()
    for r in rows:
        r["Name"] = r["Name"][:120]; w.writerow(r)
PY
rm -rf gpurun_out/prof_bridge
tail -2 gpurun_out/prof_bridge.log | cut -c1-300
