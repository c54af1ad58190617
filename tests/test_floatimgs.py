"""mrc.estimate_noise / mrc.create_threshold_mask (mrc.py:273-329) on float32 images that do NOT hold whole numbers
0..255 -- the general form of their signature (VERDICT r5 missing #4; the production path, mrc.py:372, only ever passes
float32(uint8 image), which keeps the uint8 kernels).  Fixtures: tests/golden/floatimgs.npz, made by
tests/golden/make_golden.py from the reference itself (PyWavelets float32 transform, scipy's float32 gaussian_filter,
numpy's truncating cast).  Bit-exact: sigma compared as float64 values, masks byte for byte."""
import os

import numpy as np
import pytest

import mrc_oracle as O
from mrchip import _lib, mrc
from helpers import GOLDEN


def _cases():
    z = np.load(os.path.join(GOLDEN, 'floatimgs.npz'))
    for i in range(int(z['n'])):
        img = z['in_%d' % i]
        h, w = img.shape
        dpi = int(z['dpi_%d' % i])
        yield (i, img, np.unpackbits(z['m0_%d' % i], axis=1)[:, :w].astype(bool), np.unpackbits(z['out_%d' % i], axis=1)[:, :w].astype(bool),
               float(z['sigma_%d' % i]), None if dpi < 0 else dpi)


def _same(a, b):
    return (np.isnan(a) and np.isnan(b)) or a == b


def _oracle_threshold_mask(m0, img, dpi):
    sig = O.estimate_noise(img)
    src = img
    if sig > 1.0:
        wts, _ = mrc.gaussian_weights(sig * 0.1)
        src = O.gaussian_filter(img, sig * 0.1, weights=wts)
    return m0 | O.threshold_image(src.astype(np.uint8), dpi)


def test_oracle_equals_the_reference_on_float32_images():
    n = 0
    for i, img, m0, exp, sig, dpi in _cases():
        assert img.dtype == np.float32 and not np.array_equal(img, img.astype(np.uint8))
        assert _same(O.estimate_noise(img), sig), i
        assert np.array_equal(_oracle_threshold_mask(m0, img, dpi), exp), i
        n += 1
    assert n >= 10


@pytest.mark.gpu
def test_gpu_float32_images_reference_vectors():
    for i, img, m0, exp, sig, dpi in _cases():
        assert _same(mrc.estimate_noise(img), sig), (i, mrc.estimate_noise(img), sig)
        m = m0.copy()
        td = []
        mrc.create_threshold_mask(m, img, dpi=dpi, timing_data=td)
        assert np.array_equal(m, exp), (i, int((m != exp).sum()))
        assert [k for k, _ in td] == (['est_1', 'blur_1', 'threshold'] if sig > 1.0 else ['est_1', 'threshold']), (i, td)


@pytest.mark.gpu
def test_gpu_float32_images_random_shapes_against_the_oracle():
    rng = np.random.RandomState(8)
    lib, ctx = _lib.load(), _lib.default_context()
    for k in range(40):
        h, w = int(rng.randint(1, 260)), int(rng.randint(1, 400))
        kind = k % 4
        if kind == 0: img = rng.uniform(0, 255.99, (h, w))
        elif kind == 1: img = np.clip(rng.normal(rng.uniform(30, 220), rng.uniform(0.1, 30), (h, w)), 0, 255.9)
        elif kind == 2: img = np.full((h, w), rng.uniform(0, 255)) + (rng.rand(h, w) < 0.05) * rng.uniform(0.1, 0.9)
        else: img = rng.randint(0, 256, (h, w)) / 3.0
        img = np.ascontiguousarray(img, dtype=np.float32)
        a, b = mrc.estimate_noise(img), O.estimate_noise(img)
        assert _same(a, b), (k, h, w, a, b)
        assert _same(mrc.mean_estimate_sigma(img), O.estimate_sigma(img)), (k, h, w)
        m0 = rng.rand(h, w) < 0.05
        m = m0.copy()
        dpi = [None, 120, 300][k % 3]
        mrc.create_threshold_mask(m, img, dpi=dpi)
        exp = _oracle_threshold_mask(m0, img, dpi)
        assert np.array_equal(m, exp), (k, h, w, dpi, int((m != exp).sum()))
        # the Gaussian entry point alone, several radii, library-built table too
        for sig in (0.3, 0.7, 1.6):
            wts, radius = mrc.gaussian_weights(sig)
            out = np.empty((h, w), np.uint8)
            _lib.check(lib.mrchip_gaussian_f32(ctx.handle, _lib.ptr(img, _lib.f32p), _lib.ptr(out), w, h, sig, _lib.ptr(wts, _lib.f64p), radius))
            assert np.array_equal(out, O.gaussian_filter(img, sig, weights=wts).astype(np.uint8)), (k, h, w, sig)
        out = np.empty((h, w), np.uint8)
        _lib.check(lib.mrchip_gaussian_f32(ctx.handle, _lib.ptr(img, _lib.f32p), _lib.ptr(out), w, h, 0.0, None, 0))
        assert np.array_equal(out, img.astype(np.uint8))
