#!/usr/bin/env python3
"""Stress of the host-buffer entry points on large inputs (round 6).  Sauvola with wide windows, thumbnails, optimise, the
noise estimate, repeated on a few fixed inputs, the result arrays pre-filled with 0xEE, optionally the device blocks
poisoned with 0xDD (MRCHIP_POISON=1); run several processes side by side on the one GPU.  NOTE: with FIXED shapes this
never failed, not even on the runtime's pageable path with 32 processes (41 907 calls) -- the failures of round 6 needed
fresh host arrays of ever-changing sizes, which is what tests/fuzz_parity.py produces (tools/runs/pageable_ab.sh).  A
result that differs from the oracle's is classified by what its wrong bytes hold:
   0xEE  the host array was never written there      (the download was incomplete when the call returned)
   0xDD  the device buffer was never written there    (the download read it before the kernel had stored it)
   other a wrong computation
    python tests/stress_copy_order.py [seconds] [seed]      with MRCHIP_DIRECT_PAGEABLE=1 for the runtime's pageable path"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'archive-pdf-tools_amd'))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import mrc_oracle as O
from mrchip import _lib, mrc, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.RandomState(seed)
lib, ctx = _lib.load(), _lib.default_context()
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)

cases = []
for i in range(3):
    h, w = int(rng.randint(500, 1800)), int(rng.randint(1500, 2900))
    g = synth.synth_page(w, h, 1, seed=int(rng.randint(1 << 30)), noise_sigma=6.0, line_div=14)[0]
    ww, wh = [(255, 227), (255, 60), (101, 101)][i]
    exp = np.empty(h * w, np.uint8)
    O.binarise_sauvola(g.reshape(-1), exp, w, h, ww, wh, 0.34, 128.0)
    cases.append(('sauvola', g, (ww, wh), exp))
for i in range(2):
    h, w = int(rng.randint(600, 1000)), int(rng.randint(1000, 1500))
    img = synth.synth_page(w, h, 3, seed=int(rng.randint(1 << 30)), noise_sigma=6.0, line_div=14)[0]
    flt = ['lanczos', 'bicubic'][i]
    rw, rh = int(w / 3.0), int(h / 3.0)
    exp = O.thumbnail_ex(img, rw, rh, flt, None if i == 0 else 2.0)
    cases.append(('thumbnail', img, (rw, rh, flt, None if i == 0 else 2.0), exp))
h, w = 700, 1900
img = synth.synth_page(w, h, 3, seed=5, noise_sigma=6.0, line_div=14)[0]
mask = (rng.rand(h, w) < 0.3).astype(np.uint8)
cases.append(('optimise', (mask, img), 10, O.optimise_rgb2(mask, img, w, h, 10)))
g = synth.synth_page(1200, 900, 1, seed=9, noise_sigma=8.0, line_div=14)[0]
cases.append(('sigma', g, None, O.estimate_noise(g.astype(np.float32))))

stats = {'calls': 0, 'bad_calls': 0, 'host_never_written': 0, 'device_never_written': 0, 'other': 0, 'examples': []}
t0 = time.time()
while time.time() - t0 < budget:
    kind, a, par, exp = cases[rng.randint(len(cases))]
    if kind == 'sauvola':
        h, w = a.shape
        out = np.full(h * w, 0xEE, np.uint8)
        _lib.check(lib.mrchip_sauvola_u8(ctx.handle, _lib.ptr(a), _lib.ptr(out), w, h, par[0], par[1], 0.34, 128.0, 0))
        got, want = out, exp
    elif kind == 'thumbnail':
        h, w = a.shape[:2]
        out = np.full(exp.shape, 0xEE, np.uint8)
        _lib.check(lib.mrchip_thumbnail_ex(ctx.handle, _lib.ptr(a), w, h, 3, par[0], par[1], mrc._FILTERS[par[2]],
                                           float(par[3]) if par[3] else 0.0, _lib.ptr(out)))
        got, want = out, exp
    elif kind == 'optimise':
        m, im = a
        h, w = m.shape
        out = np.full(im.shape, 0xEE, np.uint8)
        _lib.check(lib.mrchip_optimise(ctx.handle, _lib.ptr(m), _lib.ptr(im), _lib.ptr(out), w, h, 3, par, 0))
        got, want = out, exp
    else:
        s = mrc.estimate_noise(a)
        got, want = np.array([s]), np.array([exp])
    stats['calls'] += 1
    if not np.array_equal(got.reshape(-1), want.reshape(-1)):
        stats['bad_calls'] += 1
        gb = got.reshape(-1)
        wb = want.reshape(-1)
        if kind == 'sigma':
            stats['other'] += 1
            ex = {'kind': kind, 'got': float(gb[0]), 'want': float(wb[0])}
        else:
            bad = gb != wb
            n_ee, n_dd = int((gb[bad] == 0xEE).sum()), int((gb[bad] == 0xDD).sum())
            nb = int(bad.sum())
            cls = 'host_never_written' if n_ee > 0.9 * nb else ('device_never_written' if n_dd > 0.9 * nb else 'other')
            stats[cls] += 1
            idx = np.flatnonzero(bad)
            ex = {'kind': kind, 'par': [str(p) for p in (par if isinstance(par, tuple) else (par,))], 'bad_bytes': nb, 'ee': n_ee, 'dd': n_dd,
                  'first': int(idx[0]), 'last': int(idx[-1]), 'size': int(gb.size), 'class': cls}
        if len(stats['examples']) < 12:
            stats['examples'].append(ex)
        print('MISMATCH', ex, flush=True)
stats['seconds'] = round(time.time() - t0, 1)
stats['transfers'] = 'direct pageable' if os.environ.get('MRCHIP_DIRECT_PAGEABLE') == '1' else 'page-locked staging (default)' 
print('STRESS ' + json.dumps(stats), flush=True)
sys.exit(1 if stats['bad_calls'] else 0)
