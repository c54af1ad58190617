#!/bin/bash
# ONE purpose: smoke() and the default bench line, exactly as the driver runs them (`python3 bench.py`: c2, CPU baseline
# under the host-memory budget, e2e, 512-page stack).  ~3 min on a box.
#   gpurun --timeout 900 -- 'bash tools/runs/bench_default.sh r05_bench_c2'
TAG=${1:-bench_c2}
mkdir -p gpurun_out
python3 -c "import bench, json; print('host memory', json.dumps(bench.host_memory_report()), 'cpu workers', bench.cpu_workers(bench.CONFIGS['c2']))"
timeout 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 700 python3 bench.py > gpurun_out/$TAG.json 2> gpurun_out/$TAG.err; echo bench rc=$?
tail -3 gpurun_out/$TAG.err
python3 tools/runs/show_line.py gpurun_out/$TAG.json
