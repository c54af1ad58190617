#!/usr/bin/env python3
"""Throughput of BASELINE.json's other configurations through the batch API (device-resident pages,
one batch, HIP-event kernel times).  Usage: python tools/other_configs.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'archive-pdf-tools_amd'))
import numpy as np
from mrchip import _lib, mrc, synth

ctx = _lib.default_context()
CASES = [  # name, w, h, c, pages, window, fg_ds, bg_ds
    ('c2 4000x3000 RGB dpi=None bg/3', 4000, 3000, 3, 128, 51, None, 3),
    ('c2 4000x3000 RGB dpi=400 (window 101) bg/3', 4000, 3000, 3, 128, 101, None, 3),
    ('c3 3300x4600 gray dpi=None bg/3', 3300, 4600, 1, 128, 51, None, 3),
    ('c3 3300x4600 RGB dpi=None bg/3', 3300, 4600, 3, 96, 51, None, 3),
    ('c5 8000x6000 RGB dpi=364 (window 91) fg/4 bg/4', 8000, 6000, 3, 32, 91, 4, 4),
]
for name, w, h, c, n, window, fg_ds, bg_ds in CASES:
    pages = []
    for i in range(2):
        img, hocr = synth.synth_page(w, h, c, seed=300 + i, noise_sigma=6.0, line_div=60)
        pages.append((img, mrc.hocr_boxes(hocr, w, h)))
    bt = mrc.Batch(ctx, n, w, h, c)
    for i in range(n):
        bt.upload(i, pages[i % 2][0]); bt.set_boxes(i, pages[i % 2][1])
    def step():
        bt.mask_begin(window); bt.mask_finish(bt.sigmas(), True); bt.layers(fg_ds, bg_ds)
    step(); ctx.sync()
    ctx.prof_enable(True); ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(3): step()
    ctx.sync(); dt = (time.perf_counter() - t0) / 3
    rep = ctx.prof_report(); ctx.prof_enable(False)
    top = sorted(rep.items(), key=lambda kv: -kv[1]['ms'])[:6]
    print('%-52s %7.1f pages/s  %6.1f Mpx/s  (%d pages, %.1f ms)  ' % (name, n / dt, n * w * h / dt / 1e6, n, dt * 1e3) +
          ' '.join('%s=%.2f' % (k, v['ms'] / v['launches']) for k, v in top))
    bt.close()
