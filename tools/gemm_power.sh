#!/bin/bash
# socket power, shader clock and the performance level / limit reason sampled at ~10 Hz while the 256^2 GEMM
# runs the biggest text shape for several seconds - once on random operands, once on zero-filled ones.
#   gpurun -- tools/gemm_power.sh       -> gpurun_out/r06_gemm_power.txt
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
out=gpurun_out/r06_gemm_power.txt; : > $out
sample() {      # $1 = label, runs until the file $2 disappears
  while [ -e "$2" ]; do
    ts=$(date +%s.%N)
    if command -v amd-smi >/dev/null 2>&1; then
      line=$(amd-smi metric -p -c --json 2>/dev/null | tr -d '\n ' | cut -c1-700)
    else
      line=$(rocm-smi --showpower --showclocks --showperflevel --json 2>/dev/null | tr -d '\n ' | cut -c1-700)
    fi
    echo "$1 $ts $line" >> $out
    sleep 0.1
  done
}
for mode in random zero; do
  flag=$(mktemp); 
  sample "$mode" "$flag" &
  spid=$!
  sleep 1                                   # a second of idle samples first
  echo "== $mode: start $(date +%s.%N)" >> $out
  if [ $mode = zero ]; then export GEMM_ZERO=1; else unset GEMM_ZERO; fi
  timeout 120 python tools/gemm_one.py 11760 22016 4096 0 0 3000 >> $out 2>&1      # ~1.6 ms per launch: ~5 s of back-to-back GEMMs
  echo "== $mode: end $(date +%s.%N)" >> $out
  sleep 1
  rm -f "$flag"; wait $spid
done
echo "--- rocm-smi static" >> $out
rocm-smi --showmaxpower --showclkfrq 2>/dev/null | head -60 >> $out
python - <<'PY'
import re, json
rows = [l for l in open("gpurun_out/r06_gemm_power.txt") if l.startswith(("random ", "zero "))]
print(len(rows), "samples")
for mode in ("random", "zero"):
    vals = []
    for l in rows:
        if not l.startswith(mode + " "):
            continue
        m = re.findall(r'"(?:socket_power|AverageGraphicsPackagePower\(W\)|CurrentSocketGraphicsPackagePower\(W\)|average_socket_power)"\s*:\s*\{?"?(?:value"?:)?"?([0-9.]+)', l)
        if m:
            vals.append(float(m[0]))
    if vals:
        print(mode, "power samples", len(vals), "max", max(vals), "median", sorted(vals)[len(vals) // 2])
PY
tail -5 $out | cut -c1-400
