#!/bin/bash
# ONE purpose: tests/fuzz_parity.py for <seconds> with each of the given seeds, the processes side by side on the one GPU
# (small cases: the GPU is mostly idle under a single process, and neighbours vary the timing of every launch).
# A failing case leaves gpurun_out/fuzz_fail_<seed>.npz (its arrays) and its stage report in gpurun_out/fuzz_<seed>.log.
#   gpurun --timeout 600 -- 'bash tools/runs/fuzz.sh 300 71 72 73'
SECS=${1:?seconds}; shift
mkdir -p gpurun_out
pids=()
for seed in "$@"; do
  timeout $((SECS + 120)) python3 tests/fuzz_parity.py $SECS $seed > gpurun_out/fuzz_$seed.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
for seed in "$@"; do echo "== seed $seed"; tail -4 gpurun_out/fuzz_$seed.log | cut -c1-600; done
exit $rc
