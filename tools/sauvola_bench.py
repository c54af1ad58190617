#!/usr/bin/env python3
"""Sauvola microbenchmark (BASELINE.json configs[2], gray): one launch over 64 pages of 3300x4600, window 51,
and the hOCR-box variant through a full mask pipeline.  MRCHIP_LIB / MRCHIP_SAUVOLA_EXACT select the build/path.
    python tools/sauvola_bench.py [--pages 64] [--reps 10] [--w 3300 --h 4600] [--dpi N]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'archive-pdf-tools_amd'))
import numpy as np  # noqa: E402
from mrchip import _lib, mrc, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pages', type=int, default=64)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--w', type=int, default=3300)
    ap.add_argument('--h', type=int, default=4600)
    ap.add_argument('--dpi', type=int, default=None)
    ap.add_argument('--boxes', action='store_true', help='also time the hOCR-box launch (full mask pipeline)')
    a = ap.parse_args()
    ctx = _lib.default_context()
    made = synth.synth_pages([dict(w=a.w, h=a.h, channels=1, seed=303 + i, noise_sigma=6.0, line_div=60) for i in range(4)])
    bt = mrc.Batch(ctx, a.pages, a.w, a.h, 1)
    for i in range(a.pages):
        bt.upload(i, made[i % 4][0])
        bt.set_boxes(i, mrc.hocr_boxes(made[i % 4][1], a.w, a.h))
    bt.threshold(a.dpi, 0.34)
    bt.sync()
    ref = [bt.download_mask(i).sum() for i in range(4)]
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(a.reps):
        bt.threshold(a.dpi, 0.34)
    if a.boxes:
        for _ in range(a.reps):
            bt.mask_begin(mrc._window_size(a.dpi))
            bt.mask_finish(bt.sigmas(), True)
    bt.sync()
    out = {'lib': os.environ.get('MRCHIP_LIB', 'default'), 'exact': os.environ.get('MRCHIP_SAUVOLA_EXACT', '0'),
           'mask_sums': [int(x) for x in ref]}
    for name, r in ctx.prof_report().items():
        if 'sauvola' in name:
            ms = r['ms'] / r['launches']
            out[name] = {'ms': round(ms, 4), 'alg_GBps': round(r['alg_bytes'] / r['launches'] / ms / 1e6, 1),
                         'frac_of_8TBps': round(r['alg_bytes'] / r['launches'] / ms / 1e6 / 8000, 4), 'launches': r['launches']}
    print(json.dumps(out))
    bt.close()


if __name__ == '__main__':
    main()
