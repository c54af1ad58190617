#!/bin/bash
# ONE purpose: do the wrong results under many processes go away when no transfer touches pageable memory through the
# runtime?  Legs: the default (page-locked staging inside the library) and MRCHIP_DIRECT_PAGEABLE=1 (hipMemcpy*Async on the
# caller's pageable arrays, as before round 6).  <n> processes of tests/fuzz_parity.py, diagnosis mode, large-window Sauvola.
#   gpurun --timeout 1000 -- 'bash tools/runs/pageable_ab.sh 150 32 r06_pageable_ab'
SECS=${1:-150}; N=${2:-32}; TAG=${3:-pageable_ab}; FAMS=${4:-8}
mkdir -p gpurun_out
leg() {
  local name=$1 np=$2; shift 2
  local pids=()
  for i in $(seq $np); do
    env FUZZ_DIAG=1 FUZZ_FAMILIES=$FAMS "$@" timeout $((SECS + 200)) python3 tests/fuzz_parity.py $SECS $((7000 + i)) > gpurun_out/${TAG}_${name}_$i.log 2>&1 &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p; done
  local mism=$(cat gpurun_out/${TAG}_${name}_*.log | grep -c "^DIAG")
  local cases=$(grep -h "fuzz ok" gpurun_out/${TAG}_${name}_*.log | sed 's/.* \([0-9][0-9]*\) cases.*/\1/' | paste -sd+ | python3 -c "import sys; print(eval(sys.stdin.read() or \"0\"))")
  local clean=$(grep -l "fuzz ok" gpurun_out/${TAG}_${name}_*.log | wc -l)
  local ee=$(cat gpurun_out/${TAG}_${name}_*.log | grep "^DIAG" | grep -c "{238:")
  echo "LEG $name: $np processes x $SECS s: $clean ran to the end, $cases cases, $mism mismatches ($ee with host bytes never written) [$*]"
  cat gpurun_out/${TAG}_${name}_*.log | grep "^DIAG" | head -3 | cut -c1-300
  cat gpurun_out/${TAG}_${name}_*.log | grep "Error" | sort | uniq -c | sort -rn | head -3 | cut -c1-300
}
leg staged $N X=1
leg direct $N MRCHIP_DIRECT_PAGEABLE=1
