#!/bin/bash
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmc_mm
rm -rf $OUT; mkdir -p $OUT
cd /tmp
CMD="python3 $R/bench.py --steps 2 --warmup 1 --pages 32 --inflight 1 --no-cpu-baseline --no-extras"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TA_BUSY_avr TA_TA_BUSY_sum" "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1
  tail -n 2 $OUT/p$i.log | cut -c1-200
done
cd $R
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in sorted(glob.glob('$OUT/p*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        if 'resize_mm' in k or 'luma' in k or 'gauss' in k:
            agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k,v in agg.items():
    print(k)
    for c,x in sorted(v.items()): print('   %-36s %16.0f per launch'%(c, x/cnt[(k,c)]))
PY
