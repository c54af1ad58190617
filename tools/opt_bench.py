#!/usr/bin/env python3
"""optimise alone: fg only / bg only / both page-layers of a resident batch, HIP-event time per launch.
    python tools/opt_bench.py [--pages 128] [--w 4000 --h 3000] [--reps 4] [--digest]
--digest prints a SHA-256 over all fg / bg thumbnails (same-box A/B of two builds must agree)."""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'archive-pdf-tools_amd'))
from mrchip import _lib, mrc, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pages', type=int, default=128)
    ap.add_argument('--w', type=int, default=4000)
    ap.add_argument('--h', type=int, default=3000)
    ap.add_argument('--c', type=int, default=3)
    ap.add_argument('--reps', type=int, default=4)
    ap.add_argument('--distinct', type=int, default=8)
    ap.add_argument('--bg', type=float, default=3)
    ap.add_argument('--digest', action='store_true')
    ap.add_argument('--which', default='1,2,3', help='1 fg, 2 bg, 3 both')
    a = ap.parse_args()
    ctx = _lib.default_context()
    made = synth.synth_pages([dict(w=a.w, h=a.h, channels=a.c, seed=202 + i, noise_sigma=6.0, line_div=60) for i in range(a.distinct)])
    bt = mrc.Batch(ctx, a.pages, a.w, a.h, a.c)
    for i in range(a.pages):
        img, hocr = made[i % a.distinct]
        bt.upload(i, img)
        bt.set_boxes(i, mrc.hocr_boxes(hocr, a.w, a.h))
    bt.mask_begin(mrc._window_size(None))
    bt.mask_finish(bt.sigmas(), True)
    sizes = bt.layers(None, a.bg)
    ctx.sync()
    out = {'pages': a.pages, 'w': a.w, 'h': a.h}
    ctx.prof_enable(True)
    for which, name in ((1, 'fg'), (2, 'bg'), (3, 'both')):
        if str(which) not in a.which.split(','):
            continue
        ctx.prof_reset()
        for _ in range(a.reps):
            bt.layers(None, a.bg, which=which)
            ctx.sync()
        rep = ctx.prof_report()
        out[name] = {k: round(v['ms'] / v['launches'], 4) for k, v in rep.items() if 'optimise' in k}
    ctx.prof_enable(False)
    if a.digest:
        hs = hashlib.sha256()
        for p in range(min(a.pages, a.distinct)):
            hs.update(bt.download_layer(p, 0, sizes[0]).tobytes())
            hs.update(bt.download_layer(p, 1, sizes[1]).tobytes())
        out['digest'] = hs.hexdigest()[:16]
    print(json.dumps(out))


if __name__ == '__main__':
    main()
