#!/bin/bash
# rocprofv3 kernel stats of the ViT leg (configs[1], bs 32)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vit -- python $R/bench.py --workload vit --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/prof_vit.log 2>&1 ); echo "prof rc=$?"
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_vit/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("gpurun_out/prof_vit_kernel_stats.csv", "w", newline="") as o:
    w = csv.DictWriter(o, fieldnames=rows[0].keys()); w.writeheader()
    for r in rows:
        r["Name"] = r["Name"][:120]; w.writerow(r)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", tot / 1e6)
for r in rows[:22]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:100]}")
PY
rm -rf gpurun_out/prof_vit
tail -1 gpurun_out/prof_vit.log | cut -c1-200
