#!/bin/bash
# round-5 visit 4: CU-budget probe, the bench with roofline.by_kernel, kernel stats of the step
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 python tools/cu_budget_probe.py > gpurun_out/cu_budget_probe.txt 2>&1; echo "cu probe rc=$?"; cat gpurun_out/cu_budget_probe.txt | tail -8
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/v4_bench.log 2>gpurun_out/v4_bench.err
echo "bench rc=$?"; tail -1 gpurun_out/v4_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['value'], json.dumps(d['roofline'].get('by_kernel')), d['roofline']['frac'], d['roofline']['whole_step_frac'])
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk!='roofline'}) for k,v in d.get('extra',{}).items()})
"
./tools/gpu_prof.sh
