#!/usr/bin/env python3
"""Golden vectors for the page-ingest downsample of the reference (recode.py:368-372):
    image.thumbnail((w/downsample, h/downsample), resample=Image.LANCZOS, reducing_gap=None)
made with the real Pillow of this container (12.2.0; the survey found 8.4.0 bit-identical on the
resample paths).  Inputs are seeded random / structured arrays, outputs are what Pillow returned.
Run from the repo root:  python3 tests/golden/make_lanczos.py  ->  tests/golden/lanczos.npz"""
import math
import os

import numpy as np
import PIL
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.RandomState(20240607)
    cases = [  # shape, downsample, filter, reducing_gap
        ((240, 320, 3), 2, 'lanczos', None), ((301, 203), 3, 'lanczos', None), ((133, 217, 3), 2.5, 'lanczos', None),
        ((200, 152), 4, 'lanczos', None), ((97, 64, 3), 1.3, 'lanczos', None), ((480, 360, 3), 6, 'lanczos', 2.0),
        ((180, 240, 3), 3, 'bicubic', None), ((64, 700), 2, 'lanczos', None),
    ]
    out = {'pillow_version': np.array(PIL.__version__)}
    meta = []
    for i, (shape, ds, flt, gap) in enumerate(cases):
        a = rng.randint(0, 256, shape).astype(np.uint8)
        if i % 2:                       # every other case: smooth ramps + edges instead of noise
            yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
            base = ((xx * 3 + yy * 5) % 256).astype(np.uint8)
            base[(xx // 7) % 5 == 0] = 20
            a = base if len(shape) == 2 else np.stack([base, 255 - base, base // 2 + 60], axis=2).astype(np.uint8)
        im = Image.fromarray(a)
        w, h = im.size
        im.thumbnail((w / ds, h / ds), resample=Image.LANCZOS if flt == 'lanczos' else Image.BICUBIC, reducing_gap=gap)
        out['in_%d' % i] = a
        out['out_%d' % i] = np.array(im)
        meta.append('%d|%s|%s|%s|%d|%d' % (i, ds, flt, gap, math.floor(w / ds), math.floor(h / ds)))
    out['meta'] = np.array(meta)
    np.savez_compressed(os.path.join(HERE, 'lanczos.npz'), **out)
    print('wrote', os.path.join(HERE, 'lanczos.npz'), len(cases), 'cases')


if __name__ == '__main__':
    main()
