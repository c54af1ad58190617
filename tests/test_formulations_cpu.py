"""CPU checks of the two reformulations the round-4 kernels rest on, against the oracle (no GPU needed):

* Sauvola's decision as a table: for integer mean and pixel the reference's fp64 predicate (sauvola.pyx:143-151) is
  monotone in the integer variance, so `form <=> Q >= count * T2[mean][px]` (k_sauvola.hip).  The table is rebuilt here in
  numpy float64 with the reference's operation order; the GPU builds its own with its own predicate and checks it
  exhaustively (mrchip_selftest_sauvola_table).
* optimise cut into bands: runs of rows separated by >= n rows without an unselected pixel are independent
  (k_optimise.hip, OptBand); here each band is computed by the oracle from its own crop and pasted into a copy.
"""
import numpy as np
import pytest

import mrc_oracle as O
from mrchip import synth


def build_t2(k, R):
    """T2[m][px]: 0 = always foreground, 65026 = never, else the smallest floor(Q / count) that gives foreground."""
    k2 = k * k / R / R                                   # sauvola.pyx:62
    m = np.arange(256, dtype=np.float64)[:, None]
    px = np.arange(256, dtype=np.float64)[None, :]
    tmp = px + m * (k - 1)
    lhs = tmp * tmp
    A = np.broadcast_to((m * m) * k2, (256, 256))
    mi = np.arange(256)[:, None]
    vmax = np.broadcast_to(np.minimum(65025, 255 * mi + 254) - mi * mi, (256, 256))     # reachable variances
    lo = np.zeros((256, 256), np.int64)
    hi = (vmax + 1).astype(np.int64)
    for _ in range(18):                                  # smallest var in [0, vmax] with tmp <= 0 or lhs <= A * var
        mid = (lo + hi) // 2
        ok = (tmp <= 0) | (lhs <= A * mid.astype(np.float64))
        act = lo < hi
        hi = np.where(act & ok, mid, hi)
        lo = np.where(act & ~ok, mid + 1, lo)
    vmin = lo
    return np.where(vmin == 0, 0, np.where(vmin > vmax, 65026, vmin + mi * mi)).astype(np.int64), vmin, vmax


def window_sums(img, ww, wh):
    """S, Q, count of the reference's window of every pixel (rows (y-o, y+u], cols (x+r-ww, x+r], clipped)."""
    h, w = img.shape
    l, r, o, u = (ww + 1) // 2, ww // 2, (wh + 1) // 2, wh // 2
    a = img.astype(np.int64)
    I1 = np.zeros((h + 1, w + 1), np.int64); I1[1:, 1:] = a.cumsum(0).cumsum(1)
    I2 = np.zeros((h + 1, w + 1), np.int64); I2[1:, 1:] = (a * a).cumsum(0).cumsum(1)
    ys = np.arange(h)[:, None]; xs = np.arange(w)[None, :]
    y0 = np.clip(ys - o + 1, 0, h); y1 = np.clip(ys + u + 1, 0, h)
    x0 = np.clip(xs + r - ww + 1, 0, w); x1 = np.clip(xs + r + 1, 0, w)
    def box(I):
        return I[y1, x1] - I[y0, x1] - I[y1, x0] + I[y0, x0]
    return box(I1), box(I2), (y1 - y0) * (x1 - x0)


@pytest.mark.parametrize('k', [0.34, 0.1, 0.0, 0.5])
def test_sauvola_decision_table_model_equals_the_oracle(k):
    T2, vmin, vmax = build_t2(k, 128.0)
    rng = np.random.RandomState(int(k * 100) + 1)
    yy, xx = np.mgrid[0:90, 0:300]
    imgs = [rng.randint(0, 256, (120, 400)).astype(np.uint8), np.clip(rng.normal(140, 3, (100, 333)), 0, 255).astype(np.uint8),
            synth.synth_page(500, 260, 1, seed=5, noise_sigma=5.0, line_div=9)[0], np.zeros((70, 80), np.uint8),
            np.full((60, 90), 255, np.uint8), np.where((yy // 7 + xx // 5) % 2 == 0, 0, 255).astype(np.uint8),
            rng.randint(0, 256, (30, 20)).astype(np.uint8)]               # the last one is narrower and shorter than the window
    for img in imgs:
        for ww, wh in ((51, 51), (31, 31)):
            S, Q, c = window_sums(img, ww, wh)
            mean = S // c
            form = Q >= c * T2[mean, img.astype(np.int64)]
            h, w = img.shape
            exp = np.empty(h * w, np.uint8)
            O.binarise_sauvola(img.reshape(-1), exp, w, h, ww, wh, k, 128.0)
            assert np.array_equal(np.where(form, 0, 1).astype(np.uint8), exp.reshape(h, w)), (k, img.shape, ww)
            assert int((Q // c - mean * mean).max()) <= int(vmax[mean, 0].max())          # variances stay in the reachable range


def test_sauvola_table_band_and_monotone_predicate():
    """The numbers DESIGN.md and the kernel's LDS budget rest on: the entries that are neither constant lie in a band of
    px - mean; the predicate is monotone in the variance."""
    for k, band, nbytes in ((0.34, (-86, 0), 256 * 89 * 2), (0.1, (-25, 0), 256 * 28 * 2)):
        T2, vmin, vmax = build_t2(k, 128.0)
        d = np.arange(256)[None, :] - np.arange(256)[:, None]
        dlo, dhi = int(d[T2 != 0].min()), int(d[T2 != 65026].max())
        assert (dlo, dhi) == band, (k, dlo, dhi)
        assert 256 * (dhi - dlo + 3) * 2 == nbytes
        assert T2.max() <= 65026 and T2[(T2 != 0) & (T2 != 65026)].max() <= 65025
    k2 = 0.34 * 0.34 / 128.0 / 128.0
    rng = np.random.RandomState(3)
    for _ in range(200):
        m, px = float(rng.randint(1, 256)), float(rng.randint(0, 256))
        var = np.arange(0, 65026, dtype=np.float64)
        tmp = px + m * (0.34 - 1)
        p = (tmp <= 0) | (tmp * tmp <= ((m * m) * k2) * var)
        assert not np.any(p[:-1] & ~p[1:])            # once true, true for every larger variance


def bands_of(mask, n):
    """[y0, y1) runs of rows separated by >= n rows without an unselected (mask == 0) pixel -- opt_bands_kernel's rule"""
    act = np.flatnonzero((mask == 0).any(axis=1))
    out = []
    for y in act:
        if out and y - (out[-1][1] - 1) <= n:             # fewer than n untouched rows since the last active one
            out[-1][1] = y + 1
        else:
            out.append([int(y), int(y) + 1])
    return [(a, b) for a, b in out]


def _row_masks(rng, h, w, n):
    def rows_to_mask(rows, dens=0.3):
        m = np.ones((h, w), np.uint8)
        for y in rows:
            if 0 <= y < h:
                m[y] = (rng.rand(w) >= dens).astype(np.uint8)
        return m
    out = [rows_to_mask(list(range(20, 26)) + list(range(26 + gap, 26 + gap + 5))) for gap in (n - 1, n, n + 1, 2 * n + 3)]
    out += [rows_to_mask([0, 1, 2 + n + 5, h - 1]), rows_to_mask([h - 1]), rows_to_mask(range(h), dens=0.02),
            np.ones((h, w), np.uint8), rows_to_mask(range(3, h, n + 1)), rows_to_mask(range(3, h, n)),
            rows_to_mask([y for y in range(h) if rng.rand() < 0.06])]
    return out


@pytest.mark.parametrize('c,n', [(3, 10), (1, 3), (3, 7), (1, 1)])
def test_optimise_bands_computed_from_their_own_crops_equal_the_whole_page(c, n):
    rng = np.random.RandomState(10 * c + n)
    h, w = 120, 90
    f = O.optimise_gray2 if c == 1 else O.optimise_rgb2
    for mask in _row_masks(rng, h, w, n):
        img = rng.randint(0, 256, (h, w, c) if c == 3 else (h, w)).astype(np.uint8)
        whole = f(mask, img, w, h, n)
        out = img.copy()                                  # rows between bands are copies
        bands = bands_of(mask, n)
        for (a, b) in zip(bands, bands[1:]):
            assert b[0] - a[1] >= n                       # >= n untouched rows between bands
        for y0, y1 in bands:
            ca, cb = max(0, y0 - n), min(h, y1 + n)       # the rows a band's sums are rebuilt from / look ahead to
            crop_m = np.ascontiguousarray(mask[ca:cb]); crop_i = np.ascontiguousarray(img[ca:cb])
            res = f(crop_m, crop_i, w, cb - ca, n)
            out[y0:y1] = res[y0 - ca:y1 - ca]
        assert np.array_equal(out, whole), (c, n, bands[:4], int((out != whole).sum()))


def test_bench_memory_budget_respects_a_cgroup_limit(tmp_path, monkeypatch):
    """bench.py sizes its CPU-baseline worker pool and its host buffers from the container's memory, not the host's."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    b = bench.host_memory_budget()
    assert 0 < b <= 64 << 40
    with open('/proc/meminfo') as f:
        avail = [int(ln.split()[1]) * 1024 for ln in f if ln.startswith('MemAvailable')][0]
    assert b <= avail * 1.05


# ---- round 5: the Gaussian pre-blur in float32 with an undecided band (k_gauss.hip, gauss_fast_kernel) -------------------
def _fma32(a, b, c):
    """float32 fma: the product of two float32 is exact in float64; one rounding of the sum to float32 (the float64 sum is
    within 2^-53 relative of the real one: far inside the float32 half-ulp except at exact ties, which the bound absorbs)"""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def _gauss_fast_model(g, wts):
    """the kernel's float32 sequence on a whole image (reflect borders): returns y = a - 1/2 as float32"""
    R = (len(wts) - 1) // 2
    w32 = wts.astype(np.float32)
    h, w = g.shape
    gp = np.pad(g, ((R, R), (0, 0)), mode='symmetric').astype(np.float32)
    v = gp[R:R + h] * w32[R]                                     # v_pk_mul_f32
    for j in range(R, 0, -1):
        s = gp[R - j:R - j + h] + gp[R + j:R + j + h]            # exact: integers <= 510
        v = _fma32(s, np.broadcast_to(w32[R - j], s.shape), v)
    vp = np.pad(v, ((0, 0), (R, R)), mode='symmetric')
    y = _fma32(vp[:, R:R + w], np.broadcast_to(w32[R], (h, w)), np.full((h, w), -0.5, np.float32))
    for j in range(R, 0, -1):
        t = (vp[:, R - j:R - j + w] + vp[:, R + j:R + j + w]).astype(np.float32)     # v_pk_add_f32: one rounding
        y = _fma32(t, np.broadcast_to(w32[R - j], t.shape), y)
    return y


@pytest.mark.parametrize('sig', [0.3, 0.6, 0.9, 1.2, 1.9])
def test_gaussian_float32_model_decides_like_the_oracle_outside_its_band(sig):
    """|a - A| <= (2R+4) 2^-17 + 2^-15 for the float32 sequence against the reference's float64 one (checked here on images:
    noise, flats, ramps, two-level art), and wherever |fract(a - 1/2) - 1/2| >= E(R) = (2R+12) 2^-17 the byte
    round-half-even(a - 1/2) IS the reference's trunc(float32(A)) -- the pixels the GPU kernel does not recompute."""
    from mrchip import mrc
    wts, R = mrc.gaussian_weights(sig)
    rng = np.random.RandomState(int(sig * 10))
    h, w = 96, 700
    yy, xx = np.mgrid[0:h, 0:w]
    imgs = [rng.randint(0, 256, (h, w)).astype(np.uint8), np.clip(rng.normal(225, 6, (h, w)), 0, 255).astype(np.uint8),
            np.full((h, w), 255, np.uint8), np.full((h, w), 131, np.uint8), (xx * 255 // (w - 1)).astype(np.uint8),
            np.where((xx // 5 + yy // 3) % 2 == 0, 0, 255).astype(np.uint8), ((3 * xx + 7 * yy) % 256).astype(np.uint8)]
    E = (2 * R + 12) * 2.0 ** -17
    bound = (2 * R + 4) * 2.0 ** -17 + 2.0 ** -15
    for g in imgs:
        exact32 = O.gaussian_filter(g.astype(np.float32), sig, weights=wts)          # float32(A): the reference before its truncation
        y = _gauss_fast_model(g, wts).astype(np.float64)
        a = y + 0.5
        assert np.abs(a - exact32.astype(np.float64)).max() <= bound + 2.0 ** -17     # (float32(A) is within 2^-17 of A)
        decided = np.abs((y - np.floor(y)) - 0.5) >= E
        byte = np.clip(np.rint(y), 0, 255).astype(np.uint8)                           # v_cvt_pk_u8_f32: nearest even, saturating
        assert np.array_equal(byte[decided], exact32.astype(np.uint8)[decided])
        if g is imgs[0] or g is imgs[1]:
            assert decided.mean() > 0.99                                             # the band is thin on noise (sigma 0.3: weights 0.004 | 0.992 | 0.004, results hug the integers: 0.5 %)
