#!/bin/bash
# ONE purpose: tests/stress_copy_order.py, <n> processes side by side for <seconds>, first with the old behaviour
# (MRCHIP_DOWNLOAD_ORDER=0: downloads into pageable memory enqueued behind the kernels), then with the stream drained first
# (the default since round 6).  MRCHIP_POISON=1: device blocks are filled with 0xDD when handed out.
#   gpurun --timeout 1500 -- 'bash tools/runs/copy_order.sh 300 32 r06_copy_order'
SECS=${1:-300}; N=${2:-32}; TAG=${3:-copy_order}
mkdir -p gpurun_out
for mode in 0 1; do
  pids=()
  for i in $(seq $N); do
    MRCHIP_POISON=1 MRCHIP_DOWNLOAD_ORDER=$mode timeout $((SECS + 240)) python3 tests/stress_copy_order.py $SECS $((100 * mode + i)) > gpurun_out/${TAG}_m${mode}_$i.log 2>&1 &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p; done
  python3 - $TAG $mode $N <<'PY'
import sys, json, re
tag, mode, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
tot = {'calls': 0, 'bad_calls': 0, 'host_never_written': 0, 'device_never_written': 0, 'other': 0, 'processes': 0, 'examples': []}
for i in range(1, n + 1):
    txt = open('gpurun_out/%s_m%s_%d.log' % (tag, mode, i)).read()
    m = re.search(r'STRESS (\{.*\})', txt)
    if not m:
        print('process', i, 'no summary:', txt[-300:].replace('\n', ' | ')); continue
    d = json.loads(m.group(1)); tot['processes'] += 1
    for k in ('calls', 'bad_calls', 'host_never_written', 'device_never_written', 'other'): tot[k] += d[k]
    tot['examples'] += d['examples'][:2]
tot['examples'] = tot['examples'][:10]
tot['MRCHIP_DOWNLOAD_ORDER'] = mode
open('gpurun_out/%s_mode%s.json' % (tag, mode), 'w').write(json.dumps(tot, indent=1))
print('MRCHIP_DOWNLOAD_ORDER=%s: %d processes, %d calls, %d wrong results (host never written %d, device never written %d, other %d)'
      % (mode, tot['processes'], tot['calls'], tot['bad_calls'], tot['host_never_written'], tot['device_never_written'], tot['other']))
for e in tot['examples'][:6]: print('   ', e)
PY
done
