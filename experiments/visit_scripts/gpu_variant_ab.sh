#!/bin/bash
# parity of the LAST named library on the GEMM tile tests, then the text-shape sweep of every named library:  experiments/visit_scripts/gpu_variant_ab.sh a b
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
last="${@: -1}"
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
cp ab/libs/$last.so libra_amd/lib/liblibra_hip.so
timeout 200 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "every_tile or splitk or gemm_rows" 2>&1 | tail -1
cp $keep libra_amd/lib/liblibra_hip.so
SWEEP_TILES=${SWEEP_TILES:-256} experiments/visit_scripts/gpu_sweep_ab.sh ${SWEEP_SET:-text} "$@"
