#!/bin/bash
# Visit 3: failed tests again (one pytest process per file: a device-side abort must not take the other logs with it),
# timing anatomy of the dkv2 structure (DBG variants: wrong results, timing only), the driver's bench command.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
for f in test_boundary_gpu test_parity_fullsize_gpu test_generation_gpu test_dp_gpu test_decoder_model_gpu; do
  timeout 600 python -m pytest tests/$f.py -m gpu -q --maxfail=30 --timeout 500 -p no:cacheprovider > gpurun_out/pytest_$f.log 2>&1
  echo "$f rc=$? $(tail -1 gpurun_out/pytest_$f.log | cut -c1-200)"
done
for d in 0 1 2 4 6; do echo -n "dkv2 dbg $d: "; LIBRA_ATTN_DKV=2 LIBRA_DKV2_DBG=$d timeout 120 python tools/attn_bench.py bwd 2>&1 | tail -1; done | tee gpurun_out/dkv2_anatomy.txt
echo -n "dkv1: "; LIBRA_ATTN_DKV=1 timeout 120 python tools/attn_bench.py bwd 2>&1 | tail -1 | tee -a gpurun_out/dkv2_anatomy.txt
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-3500
