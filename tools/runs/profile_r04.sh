#!/bin/bash
# NOTE (round 4): both runs of this script lost their GPU box ~9 minutes in, at the full `bench.py --config c5` (CPU baseline on 8000x6000
# pages over-committing the container's memory: profiles/r04_README.md).  The c3 / c3gray / c5 lines are now made without the CPU baseline.
# Round-4 measurement set on one GPU box (repo root): rocprofv3 kernel trace + PMC passes of the default bench, of one batch
# at a time and of the Sauvola-only batch; the counter table of every kernel; the bench lines of all four configurations.
# The summaries under profiles/ are made afterwards in the build container: tools/pmc_summary.py gpurun_out/prof_<tag> <tag> "<args>".
mkdir -p gpurun_out
bash tools/profile_round.sh r04 "--pages 384 --inflight 3" > gpurun_out/prof_r04.log 2>&1
bash tools/profile_round.sh r04_inflight1 "--pages 128 --inflight 1" > gpurun_out/prof_r04_inflight1.log 2>&1
bash tools/profile_round.sh r04_c3gray "--config c3gray" > gpurun_out/prof_r04_c3gray.log 2>&1
bash tools/pmc_probe.sh "" "--pages 128 --inflight 1" "" pmc_all_r04 > /dev/null 2>&1
python3 tools/pmc_table.py gpurun_out/pmc_all_r04 > gpurun_out/r04_pmc_table.txt 2>&1
for cfg in c2 c3 c3gray c5; do
  python3 bench.py --config $cfg --no-cpu-baseline > gpurun_out/r04_bench_$cfg.json 2> gpurun_out/r04_bench_$cfg.err
done
# keep the merge small: the per-dispatch traces are not needed, the counter CSVs and stats are
find gpurun_out/prof_r04* gpurun_out/pmc_all_r04 -name "*_kernel_trace.csv" -delete 2>/dev/null
find gpurun_out/prof_r04* gpurun_out/pmc_all_r04 -name "*agent_info.csv" -delete 2>/dev/null
du -sh gpurun_out/prof_r04* gpurun_out/pmc_all_r04 | tail -5
tail -c 300 gpurun_out/r04_bench_c2.json
