#!/bin/bash
# Counter probe for one kernel family: several rocprofv3 --pmc passes (kernel trace only) over a short
# bench run, then per-launch averages for kernels whose name contains $1.
# Usage (GPU box, repo root): bash tools/pmc_probe.sh sauvola "--pages 32 --inflight 1" [script] [out-tag] [first-pass last-pass]
#   script: the program profiled (default bench.py with the flags below); e.g. tools/sauvola_bench.py
#   first / last pass: run only counter sets first..last of the twelve (one gpurun call holds six passes comfortably;
#   a second call with 7 12 adds to the same out-tag when gpurun_out/<tag> is copied back in between)
FILT=${1:-sauvola}
ARGS=${2:---pages 32 --inflight 1}
SCRIPT=${3:-}
TAG=${4:-pmc_probe}
FIRST=${5:-1}
LAST=${6:-12}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/$TAG
if [ "$FIRST" = 1 ]; then rm -rf $OUT; fi; mkdir -p $OUT
cd /tmp
CMD="python3 $R/bench.py --steps 2 --warmup 1 $ARGS --no-cpu-baseline --no-extras"
if [ -n "$SCRIPT" ]; then CMD="python3 $R/$SCRIPT $ARGS"; fi
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_THREAD_CYCLES_VALU SQ_IFETCH" "SQ_LEVEL_WAVES SQ_CYCLES" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  if [ $i -lt $FIRST ] || [ $i -gt $LAST ]; then continue; fi
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1 || tail -n 2 $OUT/p$i.log
done
cd $R
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in sorted(glob.glob('$OUT/p*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        if '$FILT' in k:
            agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k,v in agg.items():
    print(k)
    for c,x in sorted(v.items()): print('   %-28s %16.0f per launch (%d launches)'%(c, x/cnt[(k,c)], cnt[(k,c)]))
PY
