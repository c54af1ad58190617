#!/bin/bash
# experiment: existing strip kernel at small T for full batches
cd $GRAFT_REPO_ROOT
for pages in 128 96; do
for m in 0 64 128 256; do
  echo "== pages=$pages MRCHIP_OPT_STRIPS=$m"
  MRCHIP_OPT_STRIPS=$m timeout 300 python3 tools/opt_bench.py --pages $pages --reps 3 --digest 2>&1 | tail -2
done
done
