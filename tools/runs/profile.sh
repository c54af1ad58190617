#!/bin/bash
# ONE purpose: the five rocprofv3 runs of ONE bench command (kernel trace + stats; FETCH; WRITE; VALU; GUI), the program
# directly after `--` (tools/profile_round.sh).  ~3-4 min on a box.  Summaries are made afterwards in the build container:
#   python3 tools/pmc_summary.py gpurun_out/prof_<tag> <tag> "<args>"
#   gpurun --timeout 900 -- 'bash tools/runs/profile.sh r05 "--pages 384 --inflight 3"'
TAG=${1:?tag}
ARGS=${2:-}
mkdir -p gpurun_out
timeout 800 bash tools/profile_round.sh $TAG "$ARGS" > gpurun_out/prof_$TAG.log 2>&1
# keep the merge small: the per-dispatch traces are not needed, the counter CSVs and stats are
find gpurun_out/prof_$TAG -name "*_kernel_trace.csv" -delete 2>/dev/null
find gpurun_out/prof_$TAG -name "*agent_info.csv" -delete 2>/dev/null
du -sh gpurun_out/prof_$TAG; tail -5 gpurun_out/prof_$TAG.log
