#!/bin/bash
# round-5 final evidence: the driver's bench command (with cpu_baseline and the extra legs), rocprofv3 kernel stats of the same workload
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/r05_bench.log 2> gpurun_out/r05_bench.err; echo "bench rc=$?"
tail -1 gpurun_out/r05_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(d['ms_per_step'], d['value'], 'frac', r['frac'], 'whole', r['whole_step_frac'], 'traffic', r['traffic'], r['traffic_provenance']['stale'])
print(json.dumps(r['by_kernel'])[:1500])
print(d.get('cpu_baseline'))
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk!='roofline'}) for k,v in d.get('extra',{}).items()})
print(d.get('step_check'))
"
./tools/gpu_prof.sh
