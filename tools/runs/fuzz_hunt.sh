#!/bin/bash
# ONE purpose: the nondeterminism hunt of round 6: many tests/fuzz_parity.py processes side by side
# on the one GPU, in three groups --
#   S: GPU-vs-GPU determinism mode, the families of the unexplained round-5 mismatch (sauvola, gauss, sigma,
#      threshold_mask) at half size, guard bands on
#   O: the same families against the oracle, guard bands on
#   A: every family against the oracle (the regression sweep on the current kernels)
# A failing case leaves gpurun_out/fuzz_fail_<seed>.npz and its log; the summary counts cases and failures per group.
#   gpurun --timeout 1900 -- 'bash tools/runs/fuzz_hunt.sh 1500 3000 12 12 8 r06_hunt1'
SECS=${1:?seconds}; BASE=${2:?first seed}; NS=${3:-12}; NO=${4:-12}; NA=${5:-8}; TAG=${6:-hunt}
mkdir -p gpurun_out
pids=(); seeds=()
launch() {   # group, seed, env...
  local g=$1 seed=$2; shift 2
  env "$@" timeout $((SECS + 180)) python3 tests/fuzz_parity.py $SECS $seed > gpurun_out/fuzz_${g}_$seed.log 2>&1 &
  pids+=($!); seeds+=("$g:$seed")
}
s=$BASE
for i in $(seq $NS); do launch S $s FUZZ_MODE=self FUZZ_FAMILIES=0,4,6,13 FUZZ_SCALE=0.5 MRCHIP_CANARY=64; s=$((s+1)); done
for i in $(seq $NO); do launch O $s FUZZ_FAMILIES=0,4,6,13 FUZZ_SCALE=0.5 MRCHIP_CANARY=64; s=$((s+1)); done
for i in $(seq $NA); do launch A $s FUZZ_X=1; s=$((s+1)); done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
python3 - "$TAG" "${seeds[@]}" <<'PY'
import re, sys, json
tag, items = sys.argv[1], sys.argv[2:]
tot = {}
for it in items:
    g, seed = it.split(':')
    txt = open('gpurun_out/fuzz_%s_%s.log' % (g, seed)).read()
    m = re.search(r'fuzz ok.*?, (\d+) cases (\{.*\})', txt)
    t = tot.setdefault(g, {'processes': 0, 'cases': 0, 'failed': [], 'families': {}})
    t['processes'] += 1
    if m:
        t['cases'] += int(m.group(1))
        for k, v in eval(m.group(2)).items(): t['families'][k] = t['families'].get(k, 0) + v
    else:
        t['failed'].append({'seed': int(seed), 'tail': txt[-600:]})
open('gpurun_out/%s_summary.json' % tag, 'w').write(json.dumps(tot, indent=1))
for g, t in sorted(tot.items()):
    print('group %s: %d processes, %d cases, %d failed processes %s' % (g, t['processes'], t['cases'], len(t['failed']), t['families']))
    for f in t['failed']: print('  FAILED seed', f['seed'], f['tail'][-400:].replace('\n', ' | '))
PY
exit $rc
