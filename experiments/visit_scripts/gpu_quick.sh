#!/bin/bash
# quick visit: decoder-side parity tests, then the headline step without the extra legs
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_decoder_kernels_gpu.py tests/test_decoder_model_gpu.py tests/test_parity_fullsize_gpu.py tests/test_boundary_gpu.py tests/test_dp_gpu.py tests/test_f4_variants_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_quick.log 2>&1
echo "quick parity rc=$? $(tail -1 gpurun_out/pytest_quick.log)"
grep -E "^E  |^FAILED" gpurun_out/pytest_quick.log | head -20
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
tail -3 gpurun_out/bench_quick.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_quick.json').read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "img/s", d["value"], "gemm TF", d["roofline"]["achieved"], "launches", d["roofline"]["launches"], d.get("step_check"))
PY
