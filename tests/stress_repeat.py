#!/usr/bin/env python3
"""Repeat-run stress of the single-page entry points on the GPU: every random input is checked against the oracle once
and then run REPS more times, each result compared with the first (a result that changes between runs of the same
input is a race; the oracle is far slower than the GPU, so this drives ~20x more launches per second through the
kernels than tests/fuzz_parity.py does).  Other entry points are called in between with random shapes so that the
device allocator hands out blocks with different stale contents.  Not part of the pytest suite; run on a GPU box,
several processes side by side (tools/runs/stress.sh):  python tests/stress_repeat.py [seconds] [seed] [reps]
A mismatch leaves gpurun_out/stress_fail_<seed>_<n>.npz and the run exits 1 at the end."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'archive-pdf-tools_amd'))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import mrc_oracle as O
from mrchip import _lib, mrc, sauvola, optimiser, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rng = np.random.RandomState(seed)
lib, ctx = _lib.load(), _lib.default_context()
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
t0 = time.time()
cases, runs, bad = 0, 0, 0


def fail(tag, **kw):
    global bad
    bad += 1
    print('MISMATCH', tag, {k: (v.shape if isinstance(v, np.ndarray) else v) for k, v in kw.items()}, flush=True)
    np.savez_compressed(os.path.join(ROOT, 'gpurun_out', 'stress_fail_%d_%d.npz' % (seed, bad)), tag=np.array([tag]),
                        env=np.array([os.environ.get('MRCHIP_GAUSS_FAST', ''), os.environ.get('MRCHIP_SAUVOLA_COUNTED_STORES', '')]),
                        **{k: np.asarray(v) for k, v in kw.items()})


def page(h, w):
    g, _ = synth.synth_page(max(w, 64), max(h, 64), 1, seed=int(rng.randint(1 << 30)), noise_sigma=float(rng.choice([0, 3, 8, 20])),
                            line_div=int(rng.choice([8, 16])))
    return np.ascontiguousarray(g[:h, :w])


def neighbour_call():
    """another entry point with a random shape: moves the allocator's blocks around between the repeats"""
    k = rng.randint(3)
    h, w = int(rng.randint(8, 300)), int(rng.randint(8, 700))
    a = rng.randint(0, 256, (h, w)).astype(np.uint8)
    if k == 0:
        sauvola.binarise_sauvola(a.ravel(), np.zeros(h * w, np.uint8), w, h, int(rng.choice([7, 15, 25, 51])), int(rng.choice([7, 15, 25, 51])),
                                 0.34, 128.0)
    elif k == 1:
        m = (rng.rand(h, w) < 0.3)
        optimiser.optimise_gray2(m.astype(np.uint8), a, w, h, int(rng.choice([3, 10])))
    else:
        mrc.estimate_noise(a)


while time.time() - t0 < budget:
    for var in ('MRCHIP_GAUSS_FAST', 'MRCHIP_SAUVOLA_COUNTED_STORES'):
        if rng.rand() < 0.2: os.environ[var] = '0'
        else: os.environ.pop(var, None)
    h, w = int(rng.randint(8, 500)), int(rng.randint(8, 900))
    gimg = page(h, w)
    gf = gimg.astype(np.float32)
    dpi = None if rng.rand() < 0.5 else int(rng.choice([100, 200, 400]))
    m0 = rng.rand(h, w) < 0.05
    sig = O.estimate_noise(gf)
    src = gimg
    if sig > 1.0:
        wts, _r = mrc.gaussian_weights(sig * 0.1)
        src = O.gaussian_filter(gf, sig * 0.1, weights=wts).astype(np.uint8)
    exp = m0 | O.threshold_image(src, dpi)
    for rep in range(REPS):
        got = m0.copy()
        mrc.create_threshold_mask(got, gf, dpi=dpi)
        runs += 1
        if not np.array_equal(got, exp):
            # the stages apart, for the record
            gs = mrc.estimate_noise(gimg)
            gb = src
            if sig > 1.0:
                gb = np.empty_like(gimg)
                _lib.check(lib.mrchip_gaussian_u8(ctx.handle, _lib.ptr(gimg), _lib.ptr(gb), w, h, sig * 0.1, _lib.ptr(wts, _lib.f64p), _r))
            fail('threshold_mask', h=h, w=w, dpi=-1 if dpi is None else dpi, rep=rep, sigma_gpu=gs, sigma_oracle=sig, gimg=gimg, m0=m0, got=got, exp=exp,
                 blurred_oracle=src, blurred_gpu_again=gb, npx=int((got != exp).sum()), where=np.argwhere(got != exp)[:64])
        if rng.rand() < 0.15:
            neighbour_call()
    cases += 1
print('stress %s: %d s, seed %d, %d inputs, %d runs, %d mismatches' % ('ok' if not bad else 'FAILED', int(time.time() - t0), seed, cases, runs, bad), flush=True)
sys.exit(1 if bad else 0)
