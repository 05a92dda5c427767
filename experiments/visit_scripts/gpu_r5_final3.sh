#!/bin/bash
# round-5 final evidence, third take (after the persistent GEMM): HBM-side traffic of both workloads on the final GEMM sources, the whole GPU suite +
# smoke, the driver's bench command, rocprofv3 kernel stats
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
./tools/hbm_traffic.sh libra | tail -3
./tools/hbm_traffic.sh vit | tail -2
cp gpurun_out/hbm_libra.json profiles/r05_hbm_traffic_libra.json; cp gpurun_out/hbm_vit.json profiles/r05_hbm_traffic_vit.json     # (bench.py reads profiles/)
rm -f gpurun_out/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
grep -E "^E  |^FAILED" gpurun_out/pytest_gpu.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python bench.py > gpurun_out/r05_bench.log 2> gpurun_out/r05_bench.err; echo "bench rc=$?"
tail -1 gpurun_out/r05_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(d['ms_per_step'], d['value'], 'frac', r['frac'], 'whole', r['whole_step_frac'], 'traffic', r['traffic'], r['traffic_provenance'])
print(json.dumps({k:v for k,v in r['by_kernel'].items() if k!='row_kernels'}))
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk!='roofline'}) for k,v in d.get('extra',{}).items()})
print(d.get('step_check'), d['cpu_baseline']['value'])
"
./tools/gpu_prof.sh
