#!/bin/bash
# X-kernel visit: parity of every tile structure, then a shape set under 256 vs X
set -u
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "every_tile" 2>&1 | tail -3
SWEEP_TILES=256,X timeout 300 python tools/gemm_sweep.py ${1:-text} 10 > gpurun_out/sweep_x.jsonl 2> gpurun_out/sweep_x.txt
grep -v "^\[run\]" gpurun_out/sweep_x.txt | grep -v amdgpu.ids | tail -40
