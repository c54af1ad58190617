#!/bin/bash
# ONE purpose: the GPU test suite (what the driver runs first at round end).  ~100 s on a box.
#   gpurun --timeout 900 -- 'bash tools/runs/suite.sh r05_suite'
TAG=${1:-suite}
mkdir -p gpurun_out
timeout 800 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tee gpurun_out/$TAG.log | tail -15
