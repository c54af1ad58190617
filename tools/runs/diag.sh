#!/bin/bash
# ONE purpose: which condition makes the large-window Sauvola calls go wrong under many processes?  Each leg: <n>
# processes of tests/fuzz_parity.py for <seconds>, only the sauvola_big family, diagnosis mode (mismatches are described
# and counted, the run goes on).  Legs differ in one setting.
#   gpurun --timeout 1700 -- 'bash tools/runs/diag.sh 120 32 r06_diag'
SECS=${1:-120}; N=${2:-32}; TAG=${3:-diag}
mkdir -p gpurun_out
leg() {   # name, nproc, env...
  local name=$1 np=$2; shift 2
  local pids=()
  for i in $(seq $np); do
    env FUZZ_DIAG=1 FUZZ_FAMILIES=8 "$@" timeout $((SECS + 200)) python3 tests/fuzz_parity.py $SECS $((7000 + i)) > gpurun_out/${TAG}_${name}_$i.log 2>&1 &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p; done
  local mism=$(cat gpurun_out/${TAG}_${name}_*.log | grep -c "^DIAG")
  local cases=$(grep -h "fuzz ok" gpurun_out/${TAG}_${name}_*.log | sed 's/.*, \([0-9]*\) cases.*/\1/' | paste -sd+ | bc)
  local again=$(cat gpurun_out/${TAG}_${name}_*.log | grep "^DIAG" | grep -vc "again wrong px 0")
  echo "LEG $name: $np processes x $SECS s: $cases cases, $mism mismatches ($again of them wrong again on the immediate re-run) [$*]"
  cat gpurun_out/${TAG}_${name}_*.log | grep "^DIAG" | head -4 | cut -c1-330
  cat gpurun_out/${TAG}_${name}_*.log | grep -v "^DIAG\|fuzz ok\|canary" | grep -i "error\|Traceback" | sort | uniq -c | head -3
}
leg default $N X=1
leg old_order $N MRCHIP_DOWNLOAD_ORDER=0 MRCHIP_UPLOAD_ORDER=0
leg one_hw_queue $N GPU_MAX_HW_QUEUES=1
leg eight_procs 8 X=1
leg poison $N MRCHIP_POISON=1
