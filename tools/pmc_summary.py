#!/usr/bin/env python3
"""Aggregate the rocprofv3 outputs of tools/profile_round.sh into the files committed under
profiles/: <tag>_kernel_stats.csv (copy of --stats), <tag>_pmc_summary.csv/.json (per kernel:
launches, FETCH_SIZE / WRITE_SIZE per launch in bytes, gfx950-corrected read bytes).

FETCH_SIZE/WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-B request,
i.e. exactly half of the bytes of a coalesced streaming read (MI355X_MICROARCH.md, HBM section;
re-checked here on luma601: 2 x FETCH = 3*w*h*pages to 4 digits) -> read bytes = 2 x FETCH_SIZE.
WRITE_SIZE matched the written bytes exactly (luma601, unpack) and is taken as is.
Usage: python tools/pmc_summary.py gpurun_out/prof_r01 r01 "<bench args>"
"""
import collections
import csv
import json
import os
import shutil
import sys


def short(name):
    return name.split('(')[0].replace('void ', '').strip()


def valu_summary(src, tag, out, note, head, scale, srchash=None):
    """<tag>_valu_summary.json: the instruction roofline per kernel from the SQ / GRBM passes of the same command --
    VALU wave-instructions per launch, cycles the VALU was busy per instruction (SQ_ACTIVE_INST_VALU counts quad-cycles),
    fraction of the launch during which a SIMD's VALU was busy (1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs)."""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for kind in ('valu', 'gui'):
        path = os.path.join(src, 'pmc_' + kind, tag + '_counter_collection.csv')
        if not os.path.exists(path):
            return
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r['Kernel_Name'])
                agg[k][r['Counter_Name']] += float(r['Counter_Value'])
                cnt[(k, r['Counter_Name'])] += 1
    kernels = {}
    for k, v in agg.items():
        a = {c: x / cnt[(k, c)] for c, x in v.items()}
        insts, act, gui = a.get('SQ_INSTS_VALU', 0.0), a.get('SQ_ACTIVE_INST_VALU', 0.0), a.get('GRBM_GUI_ACTIVE', 0.0)
        if insts <= 0 or gui <= 0:
            continue
        kernels[k] = {'launches': cnt[(k, 'SQ_INSTS_VALU')], 'valu_wave_insts_per_launch': round(insts),
                      'valu_busy_quadcycles_per_launch': round(act), 'gui_active_cycles_per_launch_sum_xcd': round(gui),
                      'cycles_per_inst': round(4.0 * act / insts, 3), 'busy_frac': round(act / (32.0 * gui), 4),
                      'wave_cycles_per_launch': round(a.get('SQ_WAVE_CYCLES', 0.0))}
    with open(os.path.join(out, tag + '_valu_summary.json'), 'w') as f:
        json.dump({'command': 'rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU | GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace -- python3 bench.py ' + note,
                   'bench_args': note, 'head': head, 'srchash': srchash, 'scale': scale, 'kernels': kernels}, f, indent=1)


def main():
    src, tag = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, 'profiles')
    os.makedirs(out, exist_ok=True)
    shutil.copy(os.path.join(src, 'kt', tag + '_kernel_stats.csv'), os.path.join(out, tag + '_kernel_stats.csv'))
    agg = collections.defaultdict(lambda: {'launches': 0, 'fetch_kib': 0.0, 'write_kib': 0.0, 'grid': 0, 'wg': 0})
    for kind in ('fetch', 'write'):
        with open(os.path.join(src, 'pmc_' + kind, tag + '_counter_collection.csv')) as f:
            for r in csv.DictReader(f):
                a = agg[short(r['Kernel_Name'])]
                if kind == 'fetch':
                    a['launches'] += 1
                    a['grid'] = max(a['grid'], int(r['Grid_Size']))
                    a['wg'] = int(r['Workgroup_Size'])
                a[kind + '_kib'] += float(r['Counter_Value'])
    stats = {}
    with open(os.path.join(src, 'kt', tag + '_kernel_stats.csv')) as f:
        for r in csv.DictReader(f):
            stats[short(r['Name'])] = (int(r['Calls']), float(r['AverageNs']))
    rows = []
    for k, a in agg.items():
        n = max(a['launches'], 1)
        fetch = a['fetch_kib'] / n * 1024.0
        write = a['write_kib'] / n * 1024.0
        calls, avg_ns = stats.get(k, (0, 0.0))
        rows.append({'kernel': k, 'launches': n, 'max_grid_threads': a['grid'], 'workgroup': a['wg'],
                     'fetch_size_bytes_per_launch_raw': round(fetch), 'read_bytes_per_launch_gfx950_corrected': round(2 * fetch),
                     'write_bytes_per_launch': round(write), 'hbm_bytes_per_launch': round(2 * fetch + write),
                     'avg_ns_kernel_trace': round(avg_ns)})
    rows.sort(key=lambda r: -r['hbm_bytes_per_launch'] * r['launches'])
    with open(os.path.join(out, tag + '_pmc_summary.csv'), 'w', newline='') as f:
        wcsv = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        wcsv.writeheader()
        wcsv.writerows(rows)
    # algorithmic bytes per launch of the kernels bench.py quotes traffic for, from the bench line of the kernel-trace
    # pass (so that bench.py can scale the counters to its own launch size), and the commit that was profiled
    scale, head, srchash = {}, None, None
    try:
        with open(os.path.join(src, 'kt_bench.log')) as f:
            line = [ln for ln in f.read().splitlines() if ln.startswith('{')][-1]
        b = json.loads(line)
        head = b.get('head')
        srchash = b.get('srchash')
        for key in ('roofline', 'sauvola_roofline'):
            r = b.get(key)
            if r:
                scale[r['kernel']] = {'alg_bytes_per_launch': r['alg_bytes_per_launch'], 'avg_launch_ms_hip_events': r['avg_launch_ms']}
        shutil.copy(os.path.join(src, 'kt_bench.log'), os.path.join(out, tag + '_bench_under_rocprof.log'))
    except Exception as e:
        print('no bench line in kt_bench.log:', e)
    with open(os.path.join(out, tag + '_pmc_summary.json'), 'w') as f:
        json.dump({'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py ' + note,
                   'bench_args': note, 'head': head, 'srchash': srchash, 'scale': scale, 'kernels': {r['kernel']: r for r in rows}}, f, indent=1)
    valu_summary(src, tag, out, note, head, scale, srchash)
    for r in rows[:12]:
        print('%-45s n=%3d  read %.3f GB  write %.3f GB  avg %.3f ms' % (r['kernel'][:45], r['launches'],
              r['read_bytes_per_launch_gfx950_corrected'] / 1e9, r['write_bytes_per_launch'] / 1e9, r['avg_ns_kernel_trace'] / 1e6))


if __name__ == '__main__':
    main()
