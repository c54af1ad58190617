#!/bin/bash
# ONE purpose: tests/stress_repeat.py for <seconds> with each of the given seeds, side by side on the one GPU.
#   gpurun --timeout 600 -- 'bash tools/runs/stress.sh 300 1 2 3 4'
SECS=${1:?seconds}; shift
mkdir -p gpurun_out
pids=()
for seed in "$@"; do
  timeout $((SECS + 120)) python3 tests/stress_repeat.py $SECS $seed ${REPS:-25} > gpurun_out/stress_$seed.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
for seed in "$@"; do echo "== seed $seed"; tail -6 gpurun_out/stress_$seed.log | cut -c1-400; done
exit $rc
