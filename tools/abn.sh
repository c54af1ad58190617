#!/bin/bash
# Same-box comparison of several builds: archive-pdf-tools_amd/lib/ab/libmrchip_<V>.so for each V in $VARIANTS.
# Usage (GPU box, repo root): VARIANTS="A B C" bash tools/abn.sh "<bench args>" [kernel-name-filter] [reps]
ARGS=${1:---pages 128 --inflight 1 --steps 4 --warmup 1}
FILT=${2:-sauvola}
REPS=${3:-3}
mkdir -p gpurun_out
for rep in $(seq $REPS); do
  for v in ${VARIANTS:-A B}; do
    MRCHIP_LIB=$PWD/archive-pdf-tools_amd/lib/ab/libmrchip_$v.so python3 bench.py $ARGS --no-cpu-baseline --no-extras > gpurun_out/ab_$v.log 2>&1
    python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab_$v.log").read().strip().splitlines()[-1])
    print("$v", d["value"], d["ms_per_step"], {k:v["ms_per_launch"] for k,v in d["kernels"].items() if any(f in k for f in "$FILT".split(','))})
except Exception as e:
    print("$v", "failed", e, open("gpurun_out/ab_$v.log").read()[-400:])
PY
  done
done
