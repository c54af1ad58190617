"""internetarchivepdf/grayconvert.py `special_gray_convert` (recode.py:362).

Fixtures (tests/golden/grayconvert.npz, made by tests/golden/make_golden.py from the reference module and scikit-image
0.18.3): 24 small RGB images with the reference's outputs, and the rgb2hsv + lightness step over every (max, min) pair.
CPU: the oracle's restatement and the host side of the drop-in (scalar arithmetic, level tables, the 256 x 256 table)
with the two device passes replaced by numpy stand-ins that only do what the kernels do.  GPU: the kernels themselves,
byte for byte, on the fixtures, ragged shapes and a config-2 sized page (digest of the reference's output).
Tolerance: none -- results are compared for equality (the float path of the reference ends in uint8 truncation; the one
place where the drop-in's arithmetic is not numpy's own, the std from exact moments, is described in
mrchip/grayconvert.py)."""
import ctypes as C
import os

import numpy as np
import pytest

import mrc_oracle as O
from mrchip import _lib, grayconvert, synth
from helpers import GOLDEN, load_digests, sha


def _fixtures():
    z = np.load(os.path.join(GOLDEN, 'grayconvert.npz'))
    return z, int(z['n'])


def test_oracle_equals_the_reference_vectors():
    z, n = _fixtures()
    assert n >= 20
    for i in range(n):
        got = O.special_gray_convert(z['in_%d' % i].copy())
        assert got.dtype == np.uint8 and np.array_equal(got, z['out_%d' % i]), i
    mx, mn = np.mgrid[0:256, 0:256]
    img = np.zeros((256, 256, 3), np.uint8)
    img[:, :, 1] = np.maximum(mx, mn); img[:, :, 0] = np.minimum(mx, mn); img[:, :, 2] = np.minimum(mx, mn)
    assert np.array_equal(O.hsl_lightness_u8(img), z['hsl_table'])


def test_the_drop_in_table_is_scikit_images_result_for_every_max_min_pair():
    z, _ = _fixtures()
    t = grayconvert._hsl_table()
    ref = z['hsl_table']                       # [a][b] = pixel with max(a, b), min(a, b)
    tri = np.arange(256)[:, None] >= np.arange(256)[None, :]
    assert t.shape == (256, 256) and t.dtype == np.uint8
    assert np.array_equal(t[tri], ref[tri])


class _NumpyPasses:
    """Stand-in for the two device passes: exactly what the kernels compute (integer statistics; table look-ups)."""

    def __init__(self):
        self.img = None

    def mrchip_special_gray_begin(self, _ctx, p, w, h, st):
        a = np.ctypeslib.as_array(p, shape=(h, w, 3))
        self.img = a.copy()
        for c in range(3):
            x = a[:, :, c].astype(np.int64)
            st[c], st[3 + c], st[6 + c], st[9 + c] = int(x.min()), int(x.max()), int(x.sum()), int((x * x).sum())
        return 0

    def mrchip_special_gray_finish(self, _ctx, luts, tab, out):
        h, w = self.img.shape[:2]
        L = np.ctypeslib.as_array(luts, shape=(3, 256))
        T = np.ctypeslib.as_array(tab, shape=(256, 256))
        lev = np.stack([L[c][self.img[:, :, c]] for c in range(3)], axis=-1)
        res = T[lev.max(-1), lev.min(-1)]
        np.ctypeslib.as_array(out, shape=(h, w))[...] = res
        return 0

    def mrchip_last_error(self):
        return b''


def test_host_side_of_the_drop_in_reproduces_the_reference_vectors(monkeypatch):
    """Everything of mrchip.grayconvert except the kernels: statistics -> bright_adjust / thresholds -> level tables ->
    table of (max, min), on the reference's vectors and on random images against the oracle."""
    fake = _NumpyPasses()
    monkeypatch.setattr(_lib, 'load', lambda: fake)
    monkeypatch.setattr(_lib, 'default_context', lambda: type('Ctx', (), {'handle': None})())
    z, n = _fixtures()
    for i in range(n):
        got = grayconvert.special_gray_convert(z['in_%d' % i])
        assert got.dtype == np.uint8 and np.array_equal(got, z['out_%d' % i]), i
    rng = np.random.RandomState(3)
    for _ in range(60):
        h, w = int(rng.randint(1, 90)), int(rng.randint(1, 120))
        lo, hi = sorted(rng.randint(0, 256, 2)); hi = max(hi, lo + 1)
        img = rng.randint(lo, hi + 1, (h, w, 3)).astype(np.uint8)
        if img[:, :, 2].max() == 0:
            continue                                   # b_max = 0: the reference divides by zero and raises
        assert np.array_equal(grayconvert.special_gray_convert(img), O.special_gray_convert(img.copy())), (h, w, lo, hi)


def test_errors_like_the_reference(monkeypatch):
    fake = _NumpyPasses()
    monkeypatch.setattr(_lib, 'load', lambda: fake)
    monkeypatch.setattr(_lib, 'default_context', lambda: type('Ctx', (), {'handle': None})())
    with pytest.raises(ValueError):
        grayconvert.special_gray_convert(np.zeros((4, 4), np.uint8))
    with pytest.raises(ValueError):
        grayconvert.special_gray_convert(np.zeros((4, 4, 3), np.float32))
    # an all-black blue channel: b_max = 0 -> the quotient is nan / inf -> int() raises in the reference, and here
    img = np.zeros((5, 7, 3), np.uint8); img[:, :, 0] = 9
    with pytest.raises((ValueError, OverflowError)):
        O.special_gray_convert(img.copy())
    with pytest.raises((ValueError, OverflowError)):
        grayconvert.special_gray_convert(img)


@pytest.mark.gpu
def test_gpu_special_gray_convert_reference_vectors_and_ragged_shapes():
    z, n = _fixtures()
    for i in range(n):
        got = grayconvert.special_gray_convert(z['in_%d' % i])
        assert got.dtype == np.uint8 and np.array_equal(got, z['out_%d' % i]), i
    rng = np.random.RandomState(11)
    for (h, w) in ((1, 1), (1, 5), (2, 3), (7, 1021), (16, 1024), (17, 1025), (333, 257), (64, 4099), (1200, 3)):
        for kind in range(3):
            if kind == 0: img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
            elif kind == 1: img = np.clip(rng.normal(150, 40, (h, w, 3)), 0, 255).astype(np.uint8)
            else: img = synth.synth_page(max(w, 64), max(h, 64), 3, seed=h * 7 + w, noise_sigma=5.0, line_div=12)[0][:h, :w].copy()
            if img[:, :, 2].max() == 0:
                img[0, 0, 2] = 1
            exp = O.special_gray_convert(img.copy())
            got = grayconvert.special_gray_convert(img)
            assert np.array_equal(got, exp), (h, w, kind, int((got != exp).sum()))


@pytest.mark.gpu
def test_gpu_special_gray_statistics_are_exact():
    """the first pass alone: min / max / sum / sum of squares per channel against numpy's integers"""
    lib, ctx = _lib.load(), _lib.default_context()
    rng = np.random.RandomState(5)
    for (h, w) in ((1, 1), (3, 2), (5, 4), (31, 1023), (129, 2050), (700, 1301)):
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        img[rng.randint(h), rng.randint(w)] = (0, 255, 7)
        st = (C.c_ulonglong * 12)()
        _lib.check(lib.mrchip_special_gray_begin(ctx.handle, _lib.ptr(img), w, h, st))
        for c in range(3):
            x = img[:, :, c].astype(np.int64)
            assert (int(st[c]), int(st[3 + c]), int(st[6 + c]), int(st[9 + c])) == (int(x.min()), int(x.max()), int(x.sum()), int((x * x).sum())), (h, w, c)
    # finish without a pending page is a state error, not a crash: the page above is consumed by one finish
    luts = np.zeros((3, 256), np.uint8); out = np.empty((700, 1301), np.uint8)
    assert lib.mrchip_special_gray_finish(ctx.handle, _lib.ptr(luts), _lib.ptr(grayconvert._hsl_table()), _lib.ptr(out)) == 0
    assert lib.mrchip_special_gray_finish(ctx.handle, _lib.ptr(luts), _lib.ptr(grayconvert._hsl_table()), _lib.ptr(out)) != 0


@pytest.mark.gpu
def test_gpu_special_gray_convert_config2_page_digest():
    dj = load_digests()['c2_special_gray']
    img, _ = synth.synth_page(4000, 3000, 3, seed=2024, noise_sigma=6.0, line_div=60)
    assert sha(img) == dj['in']
    assert sha(grayconvert.special_gray_convert(img)) == dj['out']
