#!/bin/bash
# Same-box A/B of two builds: archive-pdf-tools_amd/lib/ab/libmrchip_{A,B}.so (box-to-box variance is
# several percent, so small kernel changes are only measurable within one gpurun call).
# Usage (GPU box, repo root): bash tools/ab.sh "<bench args>" [kernel-name-filter]
ARGS=${1:---pages 64 --inflight 1 --steps 4 --warmup 1}
FILT=${2:-optimise}
for rep in 1 2 3; do
  for v in A B; do
    MRCHIP_LIB=$PWD/archive-pdf-tools_amd/lib/ab/libmrchip_$v.so python3 bench.py $ARGS --no-cpu-baseline --no-extras > gpurun_out/ab_$v.log 2>&1
    python3 - <<PY
import json
d=json.loads(open("gpurun_out/ab_$v.log").read().strip().splitlines()[-1])
print("$v", d["value"], d["ms_per_step"], {k:v["ms_per_launch"] for k,v in d["kernels"].items() if "$FILT" in k})
PY
  done
done
