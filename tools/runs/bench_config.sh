#!/bin/bash
# ONE purpose: the bench line of ONE other configuration (c3 | c3gray | c5), no CPU baseline, short e2e leg.
#   gpurun --timeout 700 -- 'bash tools/runs/bench_config.sh c5 r05_bench_c5'
CFG=${1:?config}
TAG=${2:-bench_$CFG}
mkdir -p gpurun_out
python3 -c "import bench, json; print('host memory', json.dumps(bench.host_memory_report()))"
timeout 600 python3 bench.py --config $CFG --no-cpu-baseline --e2e-pages 64 > gpurun_out/$TAG.json 2> gpurun_out/$TAG.err; echo $CFG rc=$?
tail -3 gpurun_out/$TAG.err
python3 tools/runs/show_line.py gpurun_out/$TAG.json
