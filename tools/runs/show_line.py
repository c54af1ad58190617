#!/usr/bin/env python3
"""Digest of a bench line for the tail of a gpurun call: python3 tools/runs/show_line.py gpurun_out/<line>.json"""
import json
import sys

try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:      # noqa: BLE001
    sys.exit('no bench line in %s: %s' % (sys.argv[1], e))
print({k: d.get(k) for k in ('metric', 'value', 'unit', 'ms_per_step', 'pipeline_alg_GBps', 'pipeline_frac', 'rccl_ok',
                             'hbm_copy_GBps_measured', 'head', 'srchash')})
for key in ('roofline', 'sauvola_roofline'):
    r = d.get(key)
    if r:
        print(key, {k: r.get(k) for k in ('kernel', 'achieved', 'frac', 'avg_launch_ms', 'traffic', 'frac_of_measured_copy', 'isolated')})
c = d.get('cpu_baseline')
print('cpu', c and {k: c.get(k) for k in ('value', 'cores', 'kind', 'memory_capped', 'worker_peak_rss_GB', 'single_thread_value', 'error')})
st = d.get('config4_stack')
print('parity', d.get('parity') and {k: d['parity'].get(k) for k in ('pages_checked', 'mismatches', 'error')},
      'stack', st and {k: st[k] for k in ('pages', 'mismatches', 'all_pages_present')})
e = d.get('e2e')
if e:
    print('e2e', e['pages_per_s'], {k: v['pages_per_s'] for k, v in e['host_arrays'].items()}, 'single', d.get('single_page', {}).get('latency_ms'))
print({k: v['ms_per_launch'] for k, v in d.get('kernels', {}).items()})
