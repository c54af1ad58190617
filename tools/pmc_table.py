#!/usr/bin/env python3
"""Per-kernel table from the passes of tools/pmc_probe.sh: time, VALU / LDS busy fractions, wait share, LDS bank conflicts.
    python tools/pmc_table.py gpurun_out/pmc_all"""
import collections
import csv
import glob
import sys

src = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in sorted(glob.glob(src + '/p*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
print('%-52s %8s %6s %6s %6s %9s %7s %9s' % ('kernel', 'ms', 'VALU%', 'LDS%', 'wait%', 'cyc/VALU', 'bankcf%', 'VALU/launch'))
rows = []
for k, v in agg.items():
    a = {c: x / cnt[(k, c)] for c, x in v.items()}
    g = a.get('GRBM_GUI_ACTIVE', 0)            # summed over the 8 XCDs
    if g < 1e4:
        continue
    ms = g / 8 / 2.4e6
    valu = a.get('SQ_ACTIVE_INST_VALU', 0) / g / 32 * 100          # 32 = every SIMD of every CU busy (quad-cycle units)
    lds = a.get('SQ_LDS_IDX_ACTIVE', 0) / (g / 8 * 256) * 100
    wait = a.get('SQ_WAIT_INST_ANY', 0) / max(a.get('SQ_WAVE_CYCLES', 1), 1) * 100
    rows.append((ms, '%-52s %8.3f %6.1f %6.1f %6.1f %9.2f %7.1f %9.3g' % (
        k[:52], ms, valu, lds, wait, 4 * a.get('SQ_ACTIVE_INST_VALU', 0) / max(a.get('SQ_INSTS_VALU', 1), 1),
        100 * a.get('SQ_LDS_BANK_CONFLICT', 0) / max(a.get('SQ_LDS_IDX_ACTIVE', 1), 1), a.get('SQ_INSTS_VALU', 0))))
for _, line in sorted(rows, reverse=True):
    print(line)
