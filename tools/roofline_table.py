#!/usr/bin/env python3
"""Per-kernel roofline table of a profile set: algorithmic bytes (bench.py's accounting), HBM bytes from the
FETCH_SIZE / WRITE_SIZE passes, kernel-trace time -> achieved GB/s against the 8 TB/s peak and the measured copy rate.
    python tools/roofline_table.py r05_inflight1 [measured_copy_GBps] > profiles/r05_roofline_table.md"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r05_inflight1'
copy = float(sys.argv[2]) if len(sys.argv) > 2 else 5000.0
pm = json.load(open(os.path.join(ROOT, 'profiles', tag + '_pmc_summary.json')))
line = [ln for ln in open(os.path.join(ROOT, 'profiles', tag + '_bench_under_rocprof.log')).read().splitlines() if ln.startswith('{')][-1]
bench = json.loads(line)
# profile name of bench.py -> regular expression of the kernel symbols it times
SYM = {'optimise_rgb': r'optimise_(band|packed|strip)_kernel<3', 'optimise_gray': r'optimise_(band|packed|strip)_kernel<1',
       'sauvola': r'sauvola(_tab)?_kernel<\d+, \w+, false', 'sauvola_boxes': r'sauvola(_tab)?_kernel<\d+, \w+, true',
       'luma601': r'luma601_kernel', 'gauss_fused': r'gauss_(fast|fused)_kernel', 'thumb_resize': r'resize_mm_fused_kernel|resize_mm_kernel',
       'denoise_unpack': r'unpack_bits_kernel', 'hocr_commit': r'hocr_commit_kernel', 'dwt_dd_f32': r'dwt_dd_kernel<float>',
       'dwt_dd_f64': r'dwt_dd_kernel<double>', 'denoise_solve': r'denoise_band_kernel', 'denoise_reconcile': r'denoise_fix_kernel',
       'optimise_bands': r'opt_bands_kernel', 'thumb_reduce': r'reduce_kernel'}
print('# Roofline table of `profiles/%s_*` (`bench.py %s`, head %s, source hash %s)\n' % (tag, pm.get('bench_args'), pm.get('head'), pm.get('srchash')))
print('Algorithmic bytes = the compulsory traffic of SURVEY 8d as `bench.py` accounts it per launch; HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE of the')
print('counter passes (gfx950 correction, MI355X_MICROARCH.md); time = average of the kernel trace.  Peak 8 000 GB/s; measured device copy %.0f GB/s.\n' % copy)
print('| bench name | kernel | ms | alg. GB | HBM GB | HBM / alg. | alg. GB/s | of 8 TB/s | of measured copy | HBM GB/s |')
print('|---|---|---|---|---|---|---|---|---|---|')
rows = []
for name, k in bench['kernels'].items():
    pat = SYM.get(name)
    if not pat:
        continue
    hits = [(sym, v) for sym, v in pm['kernels'].items() if re.search(pat, sym)]
    if not hits:
        continue
    sym, v = max(hits, key=lambda kv: kv[1]['avg_ns_kernel_trace'] * kv[1]['launches'])
    ms = v['avg_ns_kernel_trace'] / 1e6
    alg = k['alg_GBps'] * k['ms_per_launch'] / 1e3            # GB per launch (HIP-event rate x HIP-event time)
    hbm = v['hbm_bytes_per_launch'] / 1e9
    rows.append((ms, '| `%s` | `%s` | %.3f | %.2f | %.2f | %s | %.0f | %.3f | %.3f | %.0f |' % (
        name, sym.replace('mrchip::', '')[:44], ms, alg, hbm, ('%.2f' % (hbm / alg)) if alg > 0 else '--',
        alg / ms * 1e3 if ms else 0, alg / ms * 1e3 / 8000.0 if ms else 0, alg / ms * 1e3 / copy if ms else 0, hbm / ms * 1e3 if ms else 0)))
for _, r in sorted(rows, reverse=True):
    print(r)
tot_ms = sum(ms for ms, _ in rows)
print('\nSum of the listed kernels: %.2f ms per launch set; `value` of the run: %s %s.' % (tot_ms, bench.get('value'), bench.get('unit')))
