#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/t_all.log
bash tools/ab.sh "--pages 128 --inflight 1 --steps 4 --warmup 1" sauvola > gpurun_out/ab_sauvola.txt 2>&1
for v in A B; do MRCHIP_LIB=$PWD/archive-pdf-tools_amd/lib/ab/libmrchip_$v.so python3 bench.py --config c3gray --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/c3gray_$v.log 2>&1; done
for v in A B; do MRCHIP_LIB=$PWD/archive-pdf-tools_amd/lib/ab/libmrchip_$v.so python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/c2_$v.log 2>&1; done
tail -n 4 gpurun_out/t_all.log; cat gpurun_out/ab_sauvola.txt
for v in A B; do python3 - <<PY
import json
for f in ("gpurun_out/c3gray_$v.log","gpurun_out/c2_$v.log"):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print("$v", f, d["value"], d["ms_per_step"], {k:x["ms_per_launch"] for k,x in d["kernels"].items() if "sauvola" in k})
PY
done
