#!/bin/bash
# Round profile on the GPU box: kernel trace + stats, then the two HBM PMC passes (separate runs,
# as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Usage (from the repo root, on the GPU box): bash tools/profile_round.sh r01 "--pages 128 --inflight 2"
set -u
TAG=${1:-r01}
ARGS=${2:---pages 128 --inflight 2}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
CMD="python3 $R/bench.py --steps 5 --warmup 1 $ARGS --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o $TAG -- $CMD > $OUT/kt_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o $TAG -- $CMD > $OUT/pmc_fetch_bench.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o $TAG -- $CMD > $OUT/pmc_write_bench.log 2>&1
# the instruction roofline of the same command: VALU instructions, VALU-busy quad-cycles, GPU-active cycles
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_valu -o $TAG -- $CMD > $OUT/pmc_valu_bench.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_gui -o $TAG -- $CMD > $OUT/pmc_gui_bench.log 2>&1
cd $R
find $OUT -type f | head -40
for f in $OUT/kt_bench.log $OUT/pmc_fetch_bench.log $OUT/pmc_write_bench.log; do grep -o '"value": [0-9.]*' $f | head -1; done
