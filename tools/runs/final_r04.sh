#!/bin/bash
# NOTE (round 4): both runs of this script lost their GPU box ~9 minutes in, at the full `bench.py --config c5` (CPU baseline on 8000x6000
# pages over-committing the container's memory: profiles/r04_README.md).  The c3 / c3gray / c5 lines are now made without the CPU baseline.
# final validation of a head on one GPU box: the GPU suite, smoke, the default bench line (what the driver runs at round end)
mkdir -p gpurun_out
timeout 2000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python3 bench.py > gpurun_out/r04_bench_c2.json 2> gpurun_out/r04_bench_c2.err; echo bench rc=$?
python3 - <<PY
import json
d=json.loads(open("gpurun_out/r04_bench_c2.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("metric","value","unit","ms_per_step","pipeline_alg_GBps","pipeline_frac","rccl_ok","hbm_copy_GBps_measured")})
print("roofline", {k:d["roofline"].get(k) for k in ("kernel","achieved","frac","avg_launch_ms","traffic","frac_of_measured_copy","isolated")})
print("sauvola", {k:d["sauvola_roofline"].get(k) for k in ("achieved","frac","avg_launch_ms","frac_of_measured_copy","isolated")}, d["sauvola_roofline"].get("valu",{}) and d["sauvola_roofline"]["valu"].get("insts_per_px"))
print("cpu", d["cpu_baseline"] and {k:d["cpu_baseline"].get(k) for k in ("value","cores","kind")})
print("parity", d.get("parity"), "stack", d.get("config4_stack") and {k:d["config4_stack"][k] for k in ("pages","mismatches","all_pages_present")})
print("e2e", d["e2e"]["pages_per_s"], {k:v["pages_per_s"] for k,v in d["e2e"]["host_arrays"].items()}, "single", d.get("single_page",{}).get("latency_ms"))
print({k:v["ms_per_launch"] for k,v in d["kernels"].items()})
PY
for cfg in c3 c3gray c5; do
  timeout 900 python3 bench.py --config $cfg --no-cpu-baseline > gpurun_out/r04_bench_$cfg.json 2> gpurun_out/r04_bench_$cfg.err; echo $cfg rc=$?
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/r04_bench_$cfg.json").read().strip().splitlines()[-1])
print("$cfg", d["value"], d["unit"], d["ms_per_step"], d.get("parity"), d["roofline"] and (d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("isolated",{}).get("frac")))
PY
done
