"""Drop-in for internetarchivepdf/grayconvert.py (`special_gray_convert`, recode.py:362): same name, argument and
result; the two passes over the pixels run in libmrchip.so on the GPU, the scalar arithmetic in between is the
reference's own, expression for expression, in this process's numpy.

What the device returns are exact integers (per-channel min, max, sum, sum of squares).  mean = sum / n is numpy's own
value (a float64 sum of uint8 values is exact).  std is taken from the exact moments, correctly rounded; numpy's
np.std adds the squared deviations pairwise in float64 and may differ from it in the last bits (relative 1e-15).  The
value only enters `bright_adjust = round(..., 4)` and from there three `int(...)` thresholds, so the result is the
reference's unless that quotient lies within ~1e-14 of a rounding boundary of the fourth decimal (never seen; the
tests compare with the reference's output byte for byte)."""
import ctypes as C
import math
from fractions import Fraction

import numpy as np

from . import _lib

perc2val = lambda x: (x * 255) / 100        # grayconvert.py:22  # noqa: E731


def level_arr(arr, minv=0, maxv=255):
    """grayconvert.py:24-31, unchanged semantics (in place on a uint8 array)."""
    interval = (maxv / 255.) - (minv / 255.)
    arr_zero = arr < minv
    arr_max = arr > maxv
    with np.errstate(all='ignore'):
        arr[::] = ((arr[::] - minv) / interval)
    arr[arr_zero] = 0
    arr[arr_max] = 255
    return arr


_HSL = None


def _hsl_table():
    """uint8[256, 256]: what grayconvert.py:62-66 returns for a pixel whose largest channel is `a` and smallest `b`
    (index [a][b], a >= b used).  skimage.color.rgb2hsv (colorconv.py:190-265) on uint8 input converts with
    `np.multiply(image, 1 / 255, dtype=float64)` (util/dtype.py:312-320), V = max, S = ptp / V (0 where ptp == 0);
    l = V * (1 - S / 2); uint8(l * 255) truncates.  Checked against scikit-image 0.18.3 over all pairs
    (tests/golden/grayconvert.npz: hsl_table)."""
    global _HSL
    if _HSL is None:
        v = np.multiply(np.arange(256, dtype=np.uint8), 1. / 255, dtype=np.float64)
        out_v = np.repeat(v[:, None], 256, axis=1)
        out_min = np.repeat(v[None, :], 256, axis=0)
        delta = out_v - out_min
        with np.errstate(all='ignore'):
            out_s = delta / out_v
        out_s[delta == 0.] = 0.
        out_s[np.isnan(out_s)] = 0
        with np.errstate(all='ignore'):                  # (entries with min > max are not pixels: zeroed below)
            l = out_v * (1 - (out_s / 2))
            t = np.array(l * 255, dtype=np.uint8)
        _HSL = np.ascontiguousarray(np.where(np.arange(256)[:, None] >= np.arange(256)[None, :], t, 0).astype(np.uint8))
    return _HSL


def special_gray_convert(imd, ctx=None):
    """grayconvert.special_gray_convert(imd): uint8[H, W, 3] -> uint8[H, W]."""
    a = np.asarray(imd)
    if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
        raise ValueError('special_gray_convert: a uint8 array of shape (H, W, 3) is expected, got %s %r' % (a.dtype, a.shape))
    a = np.ascontiguousarray(a)
    h, w = a.shape[:2]
    if h == 0 or w == 0:
        raise ValueError('special_gray_convert: empty image')         # (np.min of an empty array raises in the reference)
    ctx = ctx or _lib.default_context()
    lib = _lib.load()
    st = (C.c_ulonglong * 12)()
    _lib.check(lib.mrchip_special_gray_begin(ctx.handle, _lib.ptr(a), w, h, st), 'mrchip_special_gray_begin')
    n = h * w
    components = ('r', 'g', 'b')
    d = {}
    for i, k in enumerate(components):                                  # grayconvert.py:41-44
        mn, mx, sm, sq = int(st[i]), int(st[3 + i]), int(st[6 + i]), int(st[9 + i])
        d[k + '_min'] = np.uint8(mn) / 255.
        d[k + '_max'] = np.uint8(mx) / 255.
        d[k + '_mean'] = np.float64(np.float64(sm) / n) / 255.
        var = Fraction(n * sq - sm * sm, n * n)                         # exact
        d[k + '_std'] = np.float64(math.sqrt(var)) / 255.
    with np.errstate(all='ignore'):
        bright_adjust = round(d['r_mean'] * d['g_mean'] * d['b_mean'] /
                              (d['b_max'] * (1 - d['r_std']) * (1 - d['g_std']) * (1 - d['b_std'])), 4)   # :46-47
    low_thres = min(int((196 * d['r_min'] + 14.5) / 1), 50)             # :49
    high_thres = {                                                       # :51-55  (int(nan) raises ValueError, as there)
        'r': min(int((35.66 * bright_adjust + 48.5) / 1), 95),
        'g': min(int((39.22 * bright_adjust + 44.5) / 1), 95),
        'b': min(int((45.16 * bright_adjust + 36.5) / 1), 95),
    }
    luts = np.empty((3, 256), np.uint8)
    for i, c in enumerate(components):                                   # :57-60 on every byte value
        luts[i] = level_arr(np.arange(256, dtype=np.uint8), minv=perc2val(low_thres), maxv=perc2val(high_thres[c]))
    out = np.empty((h, w), np.uint8)
    _lib.check(lib.mrchip_special_gray_finish(ctx.handle, _lib.ptr(luts), _lib.ptr(_hsl_table()), _lib.ptr(out)),
               'mrchip_special_gray_finish')
    return out
