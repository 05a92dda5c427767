#!/bin/bash
# rocprofv3 kernel stats of the headline step for the round-2 tree (ab/r02) and the working tree, same box
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in r02 head; do
  if [ $v = r02 ]; then d=$R/ab/r02; else d=$R; fi
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$v -- python $d/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/prof_$v.log 2>&1 ); echo "prof $v rc=$?"
  python - $v <<'PY'
import csv, glob, sys
v = sys.argv[1]
f = glob.glob(f"gpurun_out/prof_{v}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(f"gpurun_out/prof_{v}_kernel_stats.csv", "w", newline="") as o:
    w = csv.DictWriter(o, fieldnames=rows[0].keys()); w.writeheader()
    for r in rows:
        r["Name"] = r["Name"][:120]; w.writerow(r)
PY
  rm -rf gpurun_out/prof_$v
done
