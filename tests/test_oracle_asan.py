"""The C restatement (oracle/mrc_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer: whole pages through
`oracle/asan_driver` (make -C oracle asan), outputs compared with the reference-made page goldens and with the ordinary
build of the oracle.  CPU only -- GPU sanitizers are not available on this pool."""
import json
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

import mrc_oracle as O
from mrchip import synth
from helpers import load_npz, unpack

ORACLE = os.path.dirname(os.path.abspath(O.__file__))


@pytest.fixture(scope='module')
def driver():
    if not shutil.which('gcc') and not shutil.which('cc'):
        pytest.skip('no C compiler')
    r = subprocess.run(['make', '-s', '-C', ORACLE, 'asan'], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip('sanitizer build unavailable: %s' % r.stderr[-300:])
    return os.path.join(ORACLE, 'asan_driver')


def run_driver(driver, tmp_path, img, boxes, window, denoise_fast, fg_ds, bg_ds):
    h, w = img.shape[:2]
    c = 1 if img.ndim == 2 else 3
    fin, fout = str(tmp_path / 'in.bin'), str(tmp_path / 'out.bin')
    with open(fin, 'wb') as f:
        f.write(struct.pack('<6i2d', w, h, c, len(boxes), window, 1 if denoise_fast else 0, float(fg_ds or 0), float(bg_ds or 0)))
        f.write(np.ascontiguousarray(img, dtype=np.uint8).tobytes())
        f.write(np.asarray(boxes, dtype=np.int32).reshape(-1, 4).tobytes())
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    r = subprocess.run([driver, fin, fout], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, 'sanitizer or driver error:\n' + r.stderr[-3000:]
    assert 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr, r.stderr[-3000:]
    raw = open(fout, 'rb').read()
    mask = np.frombuffer(raw, np.uint8, w * h).reshape(h, w).astype(bool)
    off = w * h
    layers = []
    for _ in range(2):
        lw, lh = struct.unpack_from('<2i', raw, off)
        off += 8
        shape = (lh, lw) if c == 1 else (lh, lw, 3)
        layers.append(np.frombuffer(raw, np.uint8, lw * lh * c, off).reshape(shape))
        off += lw * lh * c
    assert off == len(raw)
    return mask, layers[0], layers[1]


def test_page_goldens_through_the_sanitized_build(driver, tmp_path):
    z, meta = load_npz('pages.npz')
    ran = 0
    for i, m in enumerate(meta):
        w, h, ch, seed, ns, dpi, ds, bgd, fgd, dn = m['case']
        if bgd is not None and bgd > 100:
            continue                                   # 'too-small-to-downsample' is the Python front-end's business
        img, hocr = synth.synth_page(w * (ds or 1), h * (ds or 1), ch, seed=seed, noise_sigma=ns, line_div=16)
        if ds:
            img = np.ascontiguousarray(img[::ds, ::ds])
        boxes = O.hocr_boxes(hocr, w, h, ds)
        mask, fg, bg = run_driver(driver, tmp_path, img, boxes, O.window_size(dpi), dn == 'fast', fgd, bgd)
        # the ordinary build, same entry points, same libm Gaussian table
        o = O.create_mrc_hocr_components(img, hocr, dpi=dpi, downsample=ds, bg_downsample=bgd, fg_downsample=fgd,
                                         denoise_mask=dn, gauss_weights='libm')
        om, ofg, obg = next(o), next(o), next(o)
        assert np.array_equal(mask, om) and np.array_equal(fg, ofg) and np.array_equal(bg, obg), i
        # and what the reference itself yielded (its Gaussian table comes from numpy's exp: equal on these pages)
        assert np.array_equal(mask, unpack(z['pg_mask_%d' % i], w)), i
        assert np.array_equal(fg, z['pg_fg_%d' % i]) and np.array_equal(bg, z['pg_bg_%d' % i]), i
        ran += 1
    assert ran >= 6


@pytest.mark.parametrize('w,h,c', [(1, 1, 1), (3, 2, 3), (7, 300, 1), (300, 7, 3), (64, 1, 3), (2, 65, 1)])
def test_degenerate_shapes_through_the_sanitized_build(driver, tmp_path, w, h, c):
    rng = np.random.RandomState(w * 1000 + h)
    img = rng.randint(0, 256, (h, w) if c == 1 else (h, w, 3)).astype(np.uint8)
    boxes = [[0, 0, w, h]] if w > 1 and h > 1 else []
    mask, fg, bg = run_driver(driver, tmp_path, img, boxes, 51, True, None, 2)
    assert mask.shape == (h, w) and fg.shape[:2] == (h, w)
