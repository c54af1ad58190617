"""GPU parity (through the C ABI) for denoise, optimise, luma, noise estimate, blur,
thumbnail and the hOCR-box mask: vs the reference goldens and vs the oracle."""
import ctypes as C

import numpy as np
import pytest

import mrc_oracle as O
from mrchip import _lib, mrc, optimiser, synth
from helpers import kernel_cases, thirdparty_cases, load_npz, load_digests, unpack, sha

pytestmark = pytest.mark.gpu


def lib_ctx():
    return _lib.load(), _lib.default_context()


# ---- fast_mask_denoise -------------------------------------------------------------
def test_denoise_golden():
    z, cases = kernel_cases('denoise')
    for _, i, h, w, mincnt, n, _u in cases:
        m = unpack(z['dn_in_%d' % i], w)
        r = optimiser.fast_mask_denoise(m, w, h, mincnt, n)
        assert r is m
        exp = unpack(z['dn_out_%d' % i], w)
        assert np.array_equal(m, exp), (i, h, w, mincnt, n, int((m != exp).sum()))


@pytest.mark.parametrize('h,w,dens', [(200, 300, 0.1), (333, 2049, 0.3), (64, 4100, 0.5), (1000, 77, 0.05),
                                      (5, 5, 0.9), (6, 40, 0.7), (300, 8200, 0.2), (17, 33, 1.0)])
def test_denoise_random_vs_oracle(h, w, dens):
    rng = np.random.RandomState(h + w)
    m = rng.rand(h, w) < dens
    exp = m.copy()
    O.fast_mask_denoise(exp, w, h, 4, 2)
    optimiser.fast_mask_denoise(m, w, h, 4, 2)
    assert np.array_equal(m, exp), int((m != exp).sum())


def test_denoise_cascades_and_kat3():
    # 1-px rules: the removal cascades along the whole row / column (worst case for a parallel sweep)
    m = np.zeros((300, 4000), dtype=bool)
    m[10, 3:3990] = True
    m[50:290, 100] = True
    m[100:103, 200:3000] = True
    exp = m.copy()
    O.fast_mask_denoise(exp, 4000, 300, 4, 2)
    optimiser.fast_mask_denoise(m, 4000, 300, 4, 2)
    assert np.array_equal(m, exp)
    d = load_digests()
    pat = synth.kat_pattern(1200, 1600)
    m0 = O.threshold_image(pat, 124)
    yy, xx = np.mgrid[0:1600, 0:1200].astype(np.int64)
    for M in (97, 13, 5):
        m = m0 | (((7919 * xx + 104729 * yy + 31 * xx * yy) % M) == 0)
        optimiser.fast_mask_denoise(m, 1200, 1600, 4, 2)
        assert sha(m) == d['kat3'][str(M)]['out']


def test_denoise_general_parameters():
    rng = np.random.RandomState(5)
    for (mincnt, n) in [(2, 1), (6, 3), (0, 2), (30, 2), (3, 0)]:
        m = rng.rand(90, 110) < 0.35
        exp = m.copy()
        O.fast_mask_denoise(exp, 110, 90, mincnt, n)
        optimiser.fast_mask_denoise(m, 110, 90, mincnt, n)
        assert np.array_equal(m, exp), (mincnt, n)


# ---- optimise ------------------------------------------------------------------------
def test_optimise_golden():
    z, cases = kernel_cases('optimise')
    for _, i, h, w, n, _a, _b in cases:
        m = unpack(z['opt_mask_%d' % i], w)
        g, c = z['opt_g_%d' % i], z['opt_c_%d' % i]
        got = optimiser.optimise_gray2(m, g, w, h, n)
        assert np.array_equal(got, z['opt_g2_%d' % i]), ('gray', i, int((got != z['opt_g2_%d' % i]).sum()))
        got = optimiser.optimise_rgb2(m, c, w, h, n)
        assert np.array_equal(got, z['opt_c2_%d' % i]), ('rgb', i, int((got != z['opt_c2_%d' % i]).sum()))


@pytest.mark.parametrize('h,w,dens,n', [(120, 700, 0.1, 3), (120, 700, 0.1, 10), (300, 4111, 0.08, 10),
                                        (64, 4096, 0.5, 3), (500, 17, 0.2, 10), (40, 8200, 0.1, 10),
                                        (33, 100, 0.0, 3), (33, 100, 1.0, 10), (50, 60, 0.3, 1), (50, 60, 0.3, 0)])
def test_optimise_random_vs_oracle(h, w, dens, n):
    rng = np.random.RandomState(h * 31 + w + n)
    m = rng.rand(h, w) < dens
    g = rng.randint(0, 256, (h, w)).astype(np.uint8)
    c = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    assert np.array_equal(optimiser.optimise_gray2(m, g, w, h, n), O.optimise_gray2(m, g, w, h, n))
    assert np.array_equal(optimiser.optimise_rgb2(m, c, w, h, n), O.optimise_rgb2(m, c, w, h, n))
    lib, ctx = lib_ctx()
    out = np.empty_like(c)
    _lib.check(lib.mrchip_optimise(ctx.handle, _lib.ptr(m.view(np.uint8)), _lib.ptr(c), _lib.ptr(out), w, h, 3, n, 1))
    assert np.array_equal(out, O.optimise_rgb2(~m, c, w, h, n))          # fused mask inversion (mrc.py:439)


def test_optimise_kat4():
    d = load_digests()
    pat, pat3 = synth.kat_pattern(1200, 1600), synth.kat_pattern(1200, 1600, 3)
    m0 = O.threshold_image(pat, 124)
    assert sha(optimiser.optimise_rgb2(m0, pat3, 1200, 1600, 3)) == d['kat4a']
    assert sha(optimiser.optimise_rgb2(~m0, pat3, 1200, 1600, 10)) == d['kat4b']
    assert sha(optimiser.optimise_gray2(m0, pat, 1200, 1600, 3)) == d['kat4c']


# ---- luma / sigma / gaussian / thumbnail ------------------------------------------------
def test_luma_golden_and_random():
    z, _ = load_npz('thirdparty.npz')
    lib, ctx = lib_ctx()
    for rgb in (z['luma_in'], np.random.RandomState(1).randint(0, 256, (301, 1003, 3)).astype(np.uint8)):
        h, w = rgb.shape[:2]
        out = np.empty((h, w), np.uint8)
        _lib.check(lib.mrchip_luma601(ctx.handle, _lib.ptr(np.ascontiguousarray(rgb)), _lib.ptr(out), w, h))
        assert np.array_equal(out, O.luma601(rgb))
    out = np.empty(z['luma_out'].shape, np.uint8)
    _lib.check(lib.mrchip_luma601(ctx.handle, _lib.ptr(np.ascontiguousarray(z['luma_in'])), _lib.ptr(out),
                                  out.shape[1], out.shape[0]))
    assert np.array_equal(out, z['luma_out'])


def test_sigma_golden_and_random():
    z, cases = thirdparty_cases('sigma')
    for _, i, h, w in cases:
        f = z['sig_f_%d' % i]
        b = unpack(z['sig_b_%d' % i], w)
        vals = z['sig_vals_%d' % i]
        assert mrc.mean_estimate_sigma(f.astype(np.float32)) == vals[0], i
        assert mrc.mean_estimate_sigma(b) == vals[1], i
        assert mrc.estimate_noise(f.astype(np.float32)) == vals[2], i
    rng = np.random.RandomState(3)
    for (h, w) in [(1500, 2000), (751, 1001), (4, 4), (3, 50), (50, 3), (2, 2), (1, 9)]:
        f = rng.randint(0, 256, (h, w)).astype(np.uint8)
        assert mrc.mean_estimate_sigma(f) == O.estimate_sigma(f.astype(np.float32)), (h, w)
        b = rng.rand(h, w) < 0.15
        got, exp = mrc.mean_estimate_sigma(b), O.estimate_sigma(b)
        assert got == exp or (np.isnan(got) and np.isnan(exp)), (h, w, got, exp)
    z0 = np.zeros((40, 40), np.uint8)
    assert np.isnan(mrc.estimate_noise(z0))                  # all-zero detail band -> NaN (no blur)
    p = synth.kat_pattern(800, 600, 3)
    assert mrc.estimate_noise(O.luma601(p)) == 15.725569182346643


def test_gaussian_golden_and_random():
    z, cases = thirdparty_cases('gauss')
    lib, ctx = lib_ctx()
    for _, i, h, w, sig in cases:
        g = np.ascontiguousarray(z['gau_in_%d' % i])
        wts = np.ascontiguousarray(z['gau_w_%d' % i])
        out = np.empty_like(g)
        _lib.check(lib.mrchip_gaussian_u8(ctx.handle, _lib.ptr(g), _lib.ptr(out), w, h, float(sig),
                                          _lib.ptr(wts, _lib.f64p), (len(wts) - 1) // 2))
        assert np.array_equal(out, z['gau_out_%d' % i].astype(np.uint8)), (i, sig)
    rng = np.random.RandomState(9)
    for sig, (h, w) in [(0.39, (700, 900)), (0.6, (333, 1001)), (1.3, (50, 2000)), (5.0, (100, 100)), (0.6, (2, 3))]:
        g = rng.randint(0, 256, (h, w)).astype(np.uint8)
        wts, radius = mrc.gaussian_weights(sig)
        out = np.empty_like(g)
        _lib.check(lib.mrchip_gaussian_u8(ctx.handle, _lib.ptr(g), _lib.ptr(out), w, h, sig, _lib.ptr(wts, _lib.f64p), radius))
        exp = O.gaussian_filter(g.astype(np.float32), sig, weights=wts).astype(np.uint8)
        assert np.array_equal(out, exp), (sig, h, w)
        # library-built table (libm exp)
        _lib.check(lib.mrchip_gaussian_u8(ctx.handle, _lib.ptr(g), _lib.ptr(out), w, h, sig, None, 0))
        wl, _ = O.gaussian_weights_libm(sig)
        assert np.array_equal(out, O.gaussian_filter(g.astype(np.float32), sig, weights=wl).astype(np.uint8))


@pytest.mark.parametrize('fast', ['0', '1'])
def test_gaussian_float32_form_and_float64_form_agree_with_the_oracle(fast, monkeypatch):
    """The blur runs in float32 with the pixels it cannot decide (result within 2^-11 of an integer: ~0.1 % on noise)
    listed and recomputed in float64, and tiles with more than 32 of them -- flat areas, where the result IS an integer --
    redone whole by the float64 tile code; MRCHIP_GAUSS_FAST=0 takes the float64 tile kernel everywhere.  Noise, flat
    pages (every tile goes to the tile list), two-level art, smooth ramps (long runs of near-integers), a flat page with
    a few specks (tiles with 1 .. 40 undecided pixels: both lists in one launch), every radius, ragged sizes."""
    monkeypatch.setenv('MRCHIP_GAUSS_FAST', fast)
    lib, ctx = lib_ctx()
    rng = np.random.RandomState(21)
    h, w = 203, 1021
    yy, xx = np.mgrid[0:h, 0:w]
    specks = np.full((h, w), 200, np.uint8)
    for k in range(60):
        specks[rng.randint(0, h), rng.randint(0, w)] = rng.randint(0, 256)
    specks[40:48, 100:340] = rng.randint(190, 211, (8, 240))
    imgs = [rng.randint(0, 256, (h, w)).astype(np.uint8), np.full((h, w), 255, np.uint8), np.zeros((h, w), np.uint8),
            np.full((h, w), 77, np.uint8), np.where((xx // 5 + yy // 3) % 2 == 0, 0, 255).astype(np.uint8),
            ((xx + 2 * yy) // 3 % 256).astype(np.uint8), (xx * 255 // (w - 1)).astype(np.uint8), specks,
            np.clip(rng.normal(225, 6, (h, w)), 0, 255).astype(np.uint8)]
    for sig in (0.3, 0.6, 0.9, 1.2, 1.9):
        wts, radius = mrc.gaussian_weights(sig)
        for i, g in enumerate(imgs):
            out = np.empty_like(g)
            _lib.check(lib.mrchip_gaussian_u8(ctx.handle, _lib.ptr(g), _lib.ptr(out), w, h, sig, _lib.ptr(wts, _lib.f64p), radius))
            exp = O.gaussian_filter(g.astype(np.float32), sig, weights=wts).astype(np.uint8)
            assert np.array_equal(out, exp), (fast, sig, i, int((out != exp).sum()), np.argwhere(out != exp)[:4].tolist())
    for (hh, ww) in [(16, 16), (17, 241), (33, 479), (16, 3000)]:
        g = rng.randint(0, 256, (hh, ww)).astype(np.uint8)
        wts, radius = mrc.gaussian_weights(0.6)
        out = np.empty_like(g)
        _lib.check(lib.mrchip_gaussian_u8(ctx.handle, _lib.ptr(g), _lib.ptr(out), ww, hh, 0.6, _lib.ptr(wts, _lib.f64p), radius))
        assert np.array_equal(out, O.gaussian_filter(g.astype(np.float32), 0.6, weights=wts).astype(np.uint8)), (fast, hh, ww)


@pytest.mark.parametrize('sig', [0.3, 0.45, 0.6])
def test_gaussian_float32_vertical_sum_stays_inside_its_bound_exhaustively(sig):
    """Device self-test: for scipy's table of radius 1 / 2 every (centre, pair sum, pair sum) -- 256 x 511 (x 511) -- through
    the kernel's float32 vertical sum and the reference's float64 one: the difference never exceeds e1 = (R + 2) 2^-17 +
    255 2^-24, the figure the 2^-11 undecided band of the float32 form is derived from (k_gauss.hip)."""
    import ctypes as C
    wts, radius = mrc.gaussian_weights(sig)
    assert radius in (1, 2)
    bad, worst = C.c_longlong(-1), C.c_double(-1.0)
    _lib.check(_lib.load().mrchip_selftest_gauss_fast(_lib.default_context().handle, _lib.ptr(wts, _lib.f64p), radius, C.byref(bad),
                                                      C.byref(worst)))
    assert bad.value == 0 and 0.0 < worst.value <= (radius + 2) * 2.0 ** -17 + 255 * 2.0 ** -24, (bad.value, worst.value)


def test_gaussian_tables_the_float32_form_is_not_proven_for_take_the_float64_form():
    """the float32 form's undecided band assumes taps >= 0 that add up to 1; mrchip_gaussian_u8 accepts any
    caller-supplied table of the right radius.  A table with negative outer taps and one that sums to 0.5 must come out as
    the reference's float64 accumulate gives them (they silently went through the float32 form before)."""
    lib, ctx = lib_ctx()
    rng = np.random.RandomState(31)
    h, w = 131, 977
    g = rng.randint(60, 201, (h, w)).astype(np.uint8)
    g[30:90, 200:500] = 128                                      # a flat block: integer results in the middle of it
    for sig, wts in ((0.6, np.array([-0.05, 0.25, 0.6, 0.25, -0.05])), (0.6, np.array([0.05, 0.1, 0.2, 0.1, 0.05])),
                     (0.3, np.array([0.2, 0.5, 0.2]))):
        wts = np.ascontiguousarray(wts, dtype=np.float64)
        radius = len(wts) // 2
        out = np.empty_like(g)
        _lib.check(lib.mrchip_gaussian_u8(ctx.handle, _lib.ptr(g), _lib.ptr(out), w, h, sig, _lib.ptr(wts, _lib.f64p), radius))
        exp = O.gaussian_filter(g.astype(np.float32), sig, weights=wts)
        assert exp.min() >= 0 and exp.max() < 256
        assert np.array_equal(out, exp.astype(np.uint8)), (wts.tolist(), int((out != exp.astype(np.uint8)).sum()))


def test_gaussian_page_with_saturated_margins_through_the_redo_launch():
    """a scan with clipped-white margins marks every tile of the margins for the float64 redo; the redo launch now
    sizes its grid from the tile count (32 count words per workgroup and round).  Result against the oracle on a page large
    enough for several hundred marked tiles next to noisy ones."""
    lib, ctx = lib_ctx()
    rng = np.random.RandomState(32)
    h, w = 1500, 2100
    g = np.clip(rng.normal(215, 7, (h, w)), 0, 255).astype(np.uint8)
    g[:260] = 255; g[-200:] = 255; g[:, :300] = 255; g[:, -280:] = 255        # scanner margins
    g[700:760, 600:1500] = 0                                                 # a solid bar
    for sig in (0.62, 1.1):
        wts, radius = mrc.gaussian_weights(sig)
        out = np.empty_like(g)
        _lib.check(lib.mrchip_gaussian_u8(ctx.handle, _lib.ptr(g), _lib.ptr(out), w, h, sig, _lib.ptr(wts, _lib.f64p), radius))
        exp = O.gaussian_filter(g.astype(np.float32), sig, weights=wts).astype(np.uint8)
        assert np.array_equal(out, exp), (sig, int((out != exp).sum()))


def test_thumbnail_golden_and_random():
    z, cases = thirdparty_cases('thumb')
    lib, ctx = lib_ctx()

    def run(im, rw, rh):
        h, w = im.shape[:2]
        c = 1 if im.ndim == 2 else 3
        ow, oh = C.c_int(), C.c_int()
        lib.mrchip_thumbnail_size(w, h, rw, rh, C.byref(ow), C.byref(oh))
        out = np.empty((oh.value, ow.value) if c == 1 else (oh.value, ow.value, 3), np.uint8)
        _lib.check(lib.mrchip_thumbnail(ctx.handle, _lib.ptr(np.ascontiguousarray(im)), w, h, c, rw, rh, _lib.ptr(out)))
        return out
    for _, i, h, w, f, ch in cases:
        got = run(z['thb_in_%d' % i], int(w / f), int(h / f))
        exp = z['thb_out_%d' % i]
        assert got.shape == exp.shape and np.array_equal(got, exp), (i, h, w, f, ch)
    rng = np.random.RandomState(11)
    for (h, w, f, c) in [(3000, 4000, 3, 3), (1500, 2000, 4, 3), (999, 1333, 3, 1), (600, 801, 5, 3), (700, 500, 7, 1)]:
        im = rng.randint(0, 256, (h, w) if c == 1 else (h, w, 3)).astype(np.uint8)
        got = run(im, int(w / f), int(h / f))
        exp = O.thumbnail(im, int(w / f), int(h / f))
        assert got.shape == exp.shape and np.array_equal(got, exp), (h, w, f, c)


@pytest.mark.parametrize('env', [{'MRCHIP_THUMB_NO_FUSE': '1'}, {'MRCHIP_FUSE_NB': '2'}, {'MRCHIP_FUSE_NB': '3'},
                                 {'MRCHIP_FUSE_NB': '16'}])
def test_thumbnail_other_schedules(env):
    """The same vectors through the two-kernel form (pass-to-pass image in memory) and through the one-kernel form with
    2, 3 and 16 blocks of output rows per workgroup (the LDS ring of lines wraps; small inputs take 1 by default).
    The switches are read once per process, hence the child."""
    import os
    import subprocess
    import sys
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-m', 'gpu', '-k',
                        'test_thumbnail_golden_and_random', '-p', 'no:cacheprovider'],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and '1 passed' in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


# ---- hOCR box mask ------------------------------------------------------------------------
def test_hocr_mask_vs_oracle():
    for seed, ns, dpi in [(0, 6.0, None), (1, 2.0, 150), (2, 12.0, None), (3, 0.0, 300)]:
        img, hocr = synth.synth_page(900, 700, 1, seed=seed, noise_sigma=ns, line_div=24)
        boxes = O.hocr_boxes(hocr, 900, 700)
        assert len(boxes) >= 5
        exp = np.zeros((700, 900), dtype=bool)
        dec = []
        O.create_hocr_mask(img, exp, boxes, dpi, dec)
        got = np.zeros((700, 900), dtype=bool)
        mrc.create_hocr_mask(img, got, hocr, dpi=dpi)
        assert np.array_equal(got, exp), (seed, int((got != exp).sum()), dec)
        # decisions through the C ABI
        lib, ctx = lib_ctx()
        d = np.zeros(len(boxes), np.int32)
        m2 = np.zeros((700, 900), np.uint8)
        _lib.check(lib.mrchip_hocr_mask(ctx.handle, _lib.ptr(img), _lib.ptr(m2), 900, 700, _lib.ptr(boxes, _lib.i32p),
                                        len(boxes), O.window_size(dpi), _lib.ptr(d, _lib.i32p)))
        assert d.tolist() == dec
    assert {0, 1, 2} <= set(dec) | {0, 1, 2} and len(dec) > 0


def test_optimise_quotient_equals_integer_division_exhaustively():
    """Device self-test: the fp32 reciprocal quotient of both optimise kernels == val // cnt for every count
    1..5120 and every val 0..255*cnt (3.3e9 pairs)."""
    import ctypes as C
    bad = C.c_longlong(-1)
    _lib.check(_lib.load().mrchip_selftest_optimise_quotients(_lib.default_context().handle, C.byref(bad)))
    assert bad.value == 0


def test_optimise_whole_rows_and_column_strips_agree_with_the_oracle(monkeypatch):
    """optimise runs a page-layer either as one workgroup walking whole rows or as column strips on several workgroups
    (the left boundary columns of every output row handed over through tagged write-through granules); small batches
    take the strips by default.  Both schedules, every strip width, against the oracle."""
    import mrc_oracle as O
    from mrchip import optimiser
    rng = np.random.RandomState(11)
    cases = [(180, 1300, 3, 3, 0.1), (150, 1300, 3, 10, 0.9), (90, 4000, 3, 10, 0.92), (60, 8100, 3, 3, 0.08), (77, 1001, 1, 10, 0.5),
             (64, 700, 3, 11, 0.5), (40, 5003, 1, 7, 0.3), (33, 129, 3, 3, 0.5), (12, 64, 1, 10, 1.0), (300, 2100, 3, 1, 0.0)]
    for (h, w, c, n, dens) in cases:
        img = rng.randint(0, 256, (h, w, c) if c == 3 else (h, w)).astype(np.uint8)
        mask = (rng.rand(h, w) < dens).astype(np.uint8)
        exp = (O.optimise_gray2 if c == 1 else O.optimise_rgb2)(mask, img, w, h, n)
        for mode in ('0', '128', '256', '512', '1024', None):
            if mode is None:
                monkeypatch.delenv('MRCHIP_OPT_STRIPS', raising=False)
            else:
                monkeypatch.setenv('MRCHIP_OPT_STRIPS', mode)
            got = (optimiser.optimise_gray2 if c == 1 else optimiser.optimise_rgb2)(mask, img, w, h, n)
            assert np.array_equal(got, exp), ((h, w, c, n, dens), mode, int((got != exp).sum()))


@pytest.mark.parametrize('h,w,c,n', [(40, 10007, 3, 3), (37, 10007, 3, 10), (30, 17001, 1, 10), (24, 9217, 3, 11), (20, 16385, 1, 3),
                                     (16, 20011, 3, 10)])
def test_optimise_large_format_rows_take_column_strips_whatever_their_width(h, w, c, n):
    """rows of more than 4096 columns always go in column strips of <= 1024 threads, so their width is
    not bounded by one workgroup's LDS (RGB > 9216, gray > 16384 were refused when the whole-row geometry was asked first)."""
    import mrc_oracle as O
    from mrchip import optimiser
    rng = np.random.RandomState(w + n)
    img = rng.randint(0, 256, (h, w, c) if c == 3 else (h, w)).astype(np.uint8)
    mask = (rng.rand(h, w) < 0.15).astype(np.uint8)
    mask[:, w - 40:] = 0                    # unselected pixels at the right edge of the last strip
    exp = (O.optimise_gray2 if c == 1 else O.optimise_rgb2)(mask, img, w, h, n)
    got = (optimiser.optimise_gray2 if c == 1 else optimiser.optimise_rgb2)(mask, img, w, h, n)
    assert np.array_equal(got, exp), (h, w, c, n, int((got != exp).sum()))


def _band_masks(rng, h, w, n):
    """Masks (1 = selected = copied) whose unselected rows come in runs separated by gaps of chosen lengths."""
    out = []
    def rows_to_mask(rows, dens=0.3):
        m = np.ones((h, w), np.uint8)
        for y in rows:
            if 0 <= y < h:
                m[y] = (rng.rand(w) >= dens).astype(np.uint8)
        return m
    for gap in (n - 1, n, n + 1, 2 * n + 3):               # rows 20.., then a gap of `gap` selected rows, then more
        rows = list(range(20, 26)) + list(range(26 + gap, 26 + gap + 5))
        out.append(('gap%d' % gap, rows_to_mask(rows)))
    out.append(('row0', rows_to_mask([0, 1, 2 + n + 5, h - 1])))
    out.append(('lastrow', rows_to_mask([h - 1])))
    out.append(('firstrow_only', rows_to_mask([0])))
    out.append(('single_pixel', rows_to_mask([])))
    out[-1][1][h // 2, w // 3] = 0
    out.append(('no_gaps', rows_to_mask(range(h), dens=0.02)))
    out.append(('all_selected', np.ones((h, w), np.uint8)))
    out.append(('none_selected', np.zeros((h, w), np.uint8)))
    out.append(('every_nth', rows_to_mask(range(3, h, n + 1))))          # gaps of exactly n rows all the way down
    out.append(('every_n', rows_to_mask(range(3, h, n))))                # gaps of n - 1: one band
    out.append(('random_rows', rows_to_mask([y for y in range(h) if rng.rand() < 0.06])))
    left = np.ones((h, w), np.uint8); left[40:45, :3] = 0; left[90:93, w - 2:] = 0
    out.append(('edges', left))
    return out


@pytest.mark.parametrize('c,n', [(3, 10), (3, 3), (1, 10), (3, 7), (1, 11), (3, 1), (1, 2)])
def test_optimise_band_walkers_against_the_oracle(c, n, monkeypatch):
    """Whole rows on one workgroup go through the band queue (optimise_band_kernel): rows without an unselected pixel are
    copies, runs of rows separated by >= n of them are independent jobs that rebuild their sums from the image.  Gap
    lengths n-1 / n / n+1, unselected pixels in the first / last row, no gaps, nothing / everything selected, bands
    touching the left / right edge -- against the oracle, through the mask as given and through the invert flag."""
    import ctypes as C
    import mrc_oracle as O
    from mrchip import optimiser
    monkeypatch.setenv('MRCHIP_OPT_STRIPS', '0')           # (single calls would otherwise take column strips)
    lib = _lib.load(); ctx = _lib.default_context()
    rng = np.random.RandomState(100 * c + n)
    for (h, w) in ((150, 1300), (97, 531)):
        for name, mask in _band_masks(rng, h, w, n):
            img = rng.randint(0, 256, (h, w, c) if c == 3 else (h, w)).astype(np.uint8)
            exp = (O.optimise_gray2 if c == 1 else O.optimise_rgb2)(mask, img, w, h, n)
            got = (optimiser.optimise_gray2 if c == 1 else optimiser.optimise_rgb2)(mask, img, w, h, n)
            assert np.array_equal(got, exp), (name, h, w, int((got != exp).sum()), np.argwhere((got != exp).reshape(h, -1))[:4].tolist())
            got2 = np.empty_like(img)
            _lib.check(lib.mrchip_optimise(ctx.handle, _lib.ptr(np.ascontiguousarray(1 - mask)), _lib.ptr(img), _lib.ptr(got2), w, h, c, n, 1))
            assert np.array_equal(got2, exp), (name, 'inverted', h, w, int((got2 != exp).sum()))


def test_pages_through_the_band_walkers(monkeypatch):
    """Full decomposition with whole-row workgroups forced (small batches take column strips by default): fg = one band,
    bg = a band per group of text lines; text pages, a page without ink-free rows, an empty page."""
    import mrc_oracle as O
    from mrchip import mrc, synth
    from PIL import Image
    monkeypatch.setenv('MRCHIP_OPT_STRIPS', '0')
    pages = [synth.synth_page(1000, 700, 3, seed=21, noise_sigma=6.0, line_div=20), synth.synth_page(640, 900, 1, seed=22, noise_sigma=3.0, line_div=40)]
    rng = np.random.RandomState(3)
    dense = rng.randint(0, 256, (300, 800, 3)).astype(np.uint8)             # ink in every row
    pages.append((dense, []))
    pages.append((np.full((200, 600, 3), 230, np.uint8), []))               # no ink at all
    for img, hocr in pages:
        pil = Image.fromarray(img)
        got = list(mrc.create_mrc_hocr_components(pil, hocr, bg_downsample=3, denoise_mask='fast'))
        exp = list(O.create_mrc_hocr_components(img, hocr, bg_downsample=3, denoise_mask='fast'))
        for g, e, nm in zip(got, exp, ('mask', 'fg', 'bg')):
            assert g.shape == e.shape and np.array_equal(g, e), (img.shape, nm, int((np.asarray(g) != np.asarray(e)).sum()))


NOBANDS_CHECK = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, 'archive-pdf-tools_amd')); sys.path.insert(0, os.path.join(%(root)r, 'oracle'))
import mrc_oracle as O
from mrchip import optimiser
rng = np.random.RandomState(5)
bad = []
for (h, w, c, n, dens) in [(120, 1300, 3, 10, 0.93), (90, 700, 1, 3, 0.07), (64, 2100, 3, 7, 0.5)]:
    img = rng.randint(0, 256, (h, w, c) if c == 3 else (h, w)).astype(np.uint8)
    mask = (rng.rand(h, w) < dens).astype(np.uint8)
    mask[20:45] = 1
    exp = (O.optimise_gray2 if c == 1 else O.optimise_rgb2)(mask, img, w, h, n)
    got = (optimiser.optimise_gray2 if c == 1 else optimiser.optimise_rgb2)(mask, img, w, h, n)
    if not np.array_equal(exp, got):
        bad.append((h, w, c, n, int((exp != got).sum())))
print('NOBANDS_BAD', bad)
'''


def test_optimise_whole_page_workgroups_without_the_band_queue():
    """MRCHIP_OPT_BANDS=0 (read once per process): one workgroup per page-layer walking every row (optimise_packed_kernel),
    the schedule the band walkers replaced; it stays the fallback for n_size 0 and pages of 65536 rows or more."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MRCHIP_OPT_BANDS='0', MRCHIP_OPT_STRIPS='0')
    r = subprocess.run([sys.executable, '-c', NOBANDS_CHECK % {'root': root}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'NOBANDS_BAD []' in r.stdout, r.stdout[-2000:]
