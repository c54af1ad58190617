#!/bin/bash
# ONE purpose: does draining the stream before a download into pageable memory remove the wrong outputs the 36-process
# hunt saw?  tests/fuzz_parity.py (every family against the oracle, half the cases from the large-window Sauvola family),
# <n> processes side by side for <seconds>: first MRCHIP_DOWNLOAD_ORDER=0 (the old behaviour), then the default.
#   gpurun --timeout 1500 -- 'bash tools/runs/order_ab.sh 480 32 r06_order_ab'
SECS=${1:-480}; N=${2:-32}; TAG=${3:-order_ab}
mkdir -p gpurun_out
for mode in 0 1; do
  pids=()
  for i in $(seq $N); do
    seed=$((5000 + 100 * mode + i))
    FUZZ_BIAS=8 MRCHIP_DOWNLOAD_ORDER=$mode timeout $((SECS + 240)) python3 tests/fuzz_parity.py $SECS $seed > gpurun_out/${TAG}_m${mode}_$seed.log 2>&1 &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p; done
  ok=$(grep -l "fuzz ok" gpurun_out/${TAG}_m${mode}_*.log | wc -l)
  cases=$(grep -h "fuzz ok" gpurun_out/${TAG}_m${mode}_*.log | sed 's/.*, \([0-9]*\) cases.*/\1/' | paste -sd+ | bc)
  echo "MRCHIP_DOWNLOAD_ORDER=$mode: $N processes, $ok finished clean, $cases cases in the clean ones"
  grep -h "AssertionError\|MISMATCH" gpurun_out/${TAG}_m${mode}_*.log | cut -c1-200
done
