#!/usr/bin/env python3
"""Where the time of ONE page goes through the drop-in generator (BASELINE configs[1]: 4000x3000 RGB, dpi None, bg / 3):
wall time of each next() and of the whole page, for the default library, without the shared-mask check, and on the
runtime's pageable path.    python3 tools/single_page_phases.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'archive-pdf-tools_amd'))
import numpy as np
from mrchip import _lib, mrc, synth

img, hocr = synth.synth_page(4000, 3000, 3, seed=202, noise_sigma=6.0, line_div=60)
ctx = _lib.default_context()


def run(label, reps=6):
    best = None
    for r in range(reps):
        td = []
        t0 = time.perf_counter()
        g = mrc.create_mrc_hocr_components(img, hocr, dpi=None, bg_downsample=3, denoise_mask='fast', timing_data=td, ctx=ctx)
        ts = []
        for _ in range(3):
            next(g); ts.append(time.perf_counter())
        try: next(g)
        except StopIteration: pass
        tot = time.perf_counter() - t0
        row = [round((a - b) * 1e3, 2) for a, b in zip(ts, [t0] + ts[:-1])] + [round(tot * 1e3, 2)]
        if r and (best is None or row[-1] < best[-1]): best = row
    print('%-44s mask %.2f  fg %.2f  bg %.2f  page %.2f ms' % ((label,) + tuple(best)), flush=True)


run('default (staged transfers, shared mask)')
mrc.SHARED_MASK = False
run('SHARED_MASK = False')
os.environ['MRCHIP_DIRECT_PAGEABLE'] = '1'
run('... and MRCHIP_DIRECT_PAGEABLE=1 (round 5)')
os.environ.pop('MRCHIP_DIRECT_PAGEABLE')
mrc.SHARED_MASK = True
os.environ['MRCHIP_COPY_THREADS'] = '1'
