"""GPU parity: HIP Sauvola (through the C ABI) vs the oracle and the reference goldens."""
import numpy as np
import pytest

import mrc_oracle as O
from mrchip import sauvola, synth
from helpers import kernel_cases, unpack, load_digests, sha

pytestmark = pytest.mark.gpu


def run_gpu(img, ww, wh, k, R=128.0):
    h, w = img.shape
    out = np.empty(h * w, dtype=np.uint8)
    assert sauvola.binarise_sauvola(img.reshape(-1), out, w, h, ww, wh, k, R) == 0
    return out.reshape(h, w)


def run_cpu(img, ww, wh, k, R=128.0):
    h, w = img.shape
    out = np.empty(h * w, dtype=np.uint8)
    O.binarise_sauvola(img.reshape(-1), out, w, h, ww, wh, k, R)
    return out.reshape(h, w)


def test_golden_vectors():
    z, cases = kernel_cases('sauvola')
    for _, i, h, w, ww, wh, k in cases:
        got = run_gpu(z['sau_in_%d' % i], ww, wh, k)
        exp = unpack(z['sau_out_%d' % i], w)
        assert np.array_equal(got.astype(bool), exp), (i, h, w, ww, wh, k, int((got.astype(bool) != exp).sum()))


@pytest.mark.parametrize('h,w,ww,wh,k', [
    (300, 517, 51, 51, 0.34), (300, 517, 51, 51, 0.1), (257, 1031, 31, 31, 0.34), (100, 2000, 101, 101, 0.34),
    (400, 300, 91, 91, 0.34), (64, 64, 30, 30, -0.2), (37, 1200, 51, 51, 0.1), (3, 3, 51, 51, 0.34),
    (1, 700, 51, 51, 0.34), (700, 1, 51, 51, 0.34), (200, 900, 151, 51, 0.34), (150, 2100, 401, 101, 0.2),
    (90, 333, 2, 2, 0.34), (90, 333, 1, 1, 0.34), (128, 4000, 51, 51, 0.34), (513, 209, 51, 7, 0.5),
])
def test_random_vs_oracle(h, w, ww, wh, k):
    rng = np.random.RandomState(h * 7919 + w)
    for kind in range(3):
        if kind == 0:
            img = rng.randint(0, 256, (h, w)).astype(np.uint8)
        elif kind == 1:
            img = np.clip(rng.normal(140, 3, (h, w)), 0, 255).astype(np.uint8)    # near-flat: decision boundary
        else:
            img = synth.synth_page(w, h, 1, seed=h + w, noise_sigma=5.0, line_div=max(2, h // 30))[0]
        got, exp = run_gpu(img, ww, wh, k), run_cpu(img, ww, wh, k)
        assert np.array_equal(got, exp), (kind, int((got != exp).sum()))


def test_constant_images():
    for v in (0, 1, 127, 255):
        img = np.full((120, 300), v, dtype=np.uint8)
        assert np.array_equal(run_gpu(img, 51, 51, 0.34), run_cpu(img, 51, 51, 0.34)), v


def test_bool_out_array_and_return_value():
    img = synth.kat_pattern(128, 96)
    out = np.ndarray(96 * 128, dtype=bool)
    assert sauvola.binarise_sauvola(img.reshape(-1), out, 128, 96, 31, 31, 0.34, 128) == 0
    assert sha(out)[:16] == 'eea1434f2c20e76c' and int(out.sum()) == 10992


def test_unsupported_window_is_an_error_not_a_fallback():
    from mrchip._lib import MrchipError
    img = np.zeros((300, 300), np.uint8)
    out = np.empty(300 * 300, np.uint8)
    with pytest.raises(MrchipError):
        sauvola.binarise_sauvola(img.reshape(-1), out, 300, 300, 301, 301, 0.34, 128.0)


def test_config_sizes_digest_and_property():
    d = load_digests()
    # config 1: 1200x1600 gray, window 31 -- digest made by the reference
    img = synth.synth_page(1200, 1600, 1, seed=101, noise_sigma=6.0)[0]
    assert sha(img) == d['c1_threshold']['in']
    got = run_gpu(img, 31, 31, 0.34)
    assert sha(~got.astype(bool)) == d['c1_threshold']['out']
    # config 3 shape 3300x4600, window 51: digest + full comparison with the oracle
    img = synth.synth_page(3300, 4600, 1, seed=303, noise_sigma=6.0)[0]
    assert sha(img) == d['c3_threshold']['in']
    got = run_gpu(img, 51, 51, 0.34)
    assert sha(~got.astype(bool)) == d['c3_threshold']['out']
    # size-independent property: the image 255-p under k2-symmetric formula is NOT the complement,
    # but translating the page by a multiple of the tile leaves interior results unchanged
    sub = run_gpu(np.ascontiguousarray(img[1000:2000, 500:2500]), 51, 51, 0.34)
    assert np.array_equal(sub[26:-26, 26:-26], got[1026:1974, 526:2474])


def test_near_tie_found_by_the_fuzzer():
    """tests/fuzz_parity.py case: at pixel (31, 67) Q/count = 52231.9987 and the decision's two sides differ by
    4e-4 relative, so a quotient off by one flips the pixel.  (The hardware fp64 reciprocal alone is a ~2^-26
    seed and did exactly that; the kernel divides with a correctly rounded 1/count.)"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'sauvola_neartie.npz'))
    img = np.ascontiguousarray(z['img'])
    ww, wh, k, R = z['params']
    h, w = img.shape
    out = np.empty(h * w, np.uint8)
    sauvola.binarise_sauvola(img.reshape(-1), out, w, h, int(ww), int(wh), float(k), float(R))
    assert np.array_equal(out.reshape(h, w), z['expected'])


def test_fp64_quotients_equal_integer_division_for_every_divisor():
    """Device self-test: floor(fma(N, r, r/2)) with the Newton-refined reciprocal r == N // c for c = 1..65792 and
    the dividends k*c-1, k*c, k*c+1 over the 32-bit range (~8e8 pairs)."""
    import ctypes as C
    from mrchip import _lib
    bad = C.c_longlong(-1)
    _lib.check(_lib.load().mrchip_selftest_sauvola_quotients(_lib.default_context().handle, C.byref(bad)))
    assert bad.value == 0


@pytest.mark.parametrize('k,R', [(0.34, 128.0), (0.1, 128.0), (0.0, 128.0), (0.2, 128.0), (0.5, 128.0), (0.34, 64.0)])
def test_decision_table_equals_the_fp64_sequence_for_every_mean_pixel_variance(k, R):
    """Device self-test of the table-driven decision: for every integer mean, pixel and reachable variance (7.1e8
    triples per (k, R)) `var + mean^2 >= T2[mean][px]`, read with the kernel's index arithmetic from the table the
    kernel stages into LDS, equals the reference's fp64 sequence (sauvola.pyx:143-153) as the general path runs it."""
    import ctypes as C
    from mrchip import _lib
    bad, tested, nbytes = C.c_longlong(-1), C.c_longlong(0), C.c_int(0)
    _lib.check(_lib.load().mrchip_selftest_sauvola_table(_lib.default_context().handle, k, R, C.byref(bad), C.byref(tested),
                                                         C.byref(nbytes)))
    assert bad.value == 0, (k, R, bad.value)
    assert tested.value == 256 * sum(min(65025, 255 * m + 254) - m * m + 1 for m in range(256))
    assert 0 < nbytes.value <= 128 * 1024


@pytest.mark.parametrize('mode', ['0', '1'])
def test_decision_paths_agree_with_the_oracle(mode, monkeypatch):
    """The decision has two forms: the reference's fp64 sequence on exact quotients (any k, any count;
    MRCHIP_SAUVOLA_FAST=0 everywhere) and, by default for k >= 0, `Q >= count * T2[mean][px]` with the integer mean from
    a magic-number multiply -- scalar operands where a strip sees the full window width, per-column records at the left
    / right border, the fp64 sequence where the window is clipped vertically on border strips or the count has no
    magic number.  Both settings must equal the oracle on pages (borders = per-lane counts, top / bottom = changing row
    counts), on images narrower / shorter than the window, and on the box masks (two polarities)."""
    from mrchip import mrc
    monkeypatch.setenv('MRCHIP_SAUVOLA_FAST', mode)
    rng = np.random.RandomState(5 + int(mode))
    for (h, w, ww, wh, k) in [(260, 1300, 51, 51, 0.34), (90, 520, 51, 51, 0.1), (300, 700, 31, 75, 0.34), (64, 2000, 91, 91, 0.5),
                              (40, 300, 51, 51, 0.0), (300, 1100, 101, 101, 0.34), (70, 40, 51, 51, 0.34), (600, 1500, 51, 51, 0.2),
                              (120, 1030, 101, 101, 0.1), (33, 2100, 25, 25, 0.34)]:
        for img in (rng.randint(0, 256, (h, w)).astype(np.uint8), np.clip(rng.normal(120, 2.5, (h, w)), 0, 255).astype(np.uint8),
                    synth.synth_page(w, h, 1, seed=h + w + 1, noise_sigma=4.0, line_div=max(2, h // 30))[0],
                    (rng.randint(0, 2, (h, w)) * 255).astype(np.uint8)):
            got, exp = run_gpu(img, ww, wh, k), run_cpu(img, ww, wh, k)
            assert np.array_equal(got, exp), (mode, h, w, ww, wh, k, int((got != exp).sum()))
    # the two-polarity launch: create_hocr_mask against the oracle
    for (pw, ph, seed) in [(900, 700, 77), (2300, 900, 78)]:
        img, hocr = synth.synth_page(pw, ph, 1, seed=seed, noise_sigma=5.0, line_div=14)
        m_gpu = np.zeros(img.shape, dtype=np.bool_)
        m_cpu = np.zeros(img.shape, dtype=np.bool_)
        mrc.create_hocr_mask(img, m_gpu, hocr)
        O.create_hocr_mask(img, m_cpu, mrc.hocr_boxes(hocr, pw, ph), None)
        assert np.array_equal(m_gpu, m_cpu), (mode, int((m_gpu != m_cpu).sum()))


@pytest.mark.parametrize('counted', ['0', '1'])
def test_counted_stores_and_plain_stores_agree_with_the_oracle(counted, monkeypatch):
    """The 8-column page kernel issues its stores as asm under an exec mask -- exactly one vector store (and one byte of the
    1-bpp row) per row -- so that the hand-counted `s_waitcnt vmcnt(N)` of the row queues can count them; with
    MRCHIP_SAUVOLA_COUNTED_STORES=0 the compiler's stores run instead.  Widths that are not multiples of 8 (the lane at
    the right edge stores zeros past column w, into the row's padding), pages through the batch (1-bpp rows fused) and
    through threshold_image (bytes only), tiles shorter than the queue depth."""
    from mrchip import mrc
    monkeypatch.setenv('MRCHIP_SAUVOLA_COUNTED_STORES', counted)
    rng = np.random.RandomState(17)
    for (h, w) in [(300, 1027), (257, 1500), (1, 1100), (2, 1300), (3, 2049), (700, 1029)]:
        img = synth.synth_page(w, max(h, 40), 1, seed=h + w, noise_sigma=5.0, line_div=6)[0][:h]
        img = np.ascontiguousarray(img)
        got, exp = run_gpu(img, 51, 51, 0.34), run_cpu(img, 51, 51, 0.34)
        assert np.array_equal(got, exp), (counted, h, w, int((got != exp).sum()))
        assert np.array_equal(mrc.threshold_image(img, None), O.threshold_image(img, None))
    for (pw, ph, seed) in [(1301, 333, 5), (2051, 270, 6)]:
        img, hocr = synth.synth_page(pw, ph, 3, seed=seed, noise_sigma=5.0, line_div=8)
        g = mrc.create_mrc_hocr_components(img, hocr, denoise_mask='fast', bg_downsample=3)
        e = O.create_mrc_hocr_components(img, hocr, denoise_mask='fast', bg_downsample=3)
        for k, (a, b) in enumerate(zip(g, e)):
            assert a.shape == b.shape and np.array_equal(a, b), (counted, pw, k)


def _headline_page(w, h, seed):
    """a page whose hOCR has page-like boxes: the whole page, a tall headline, long lines, a light-on-dark banner"""
    img, _ = synth.synth_page(w, h, 1, seed=seed, noise_sigma=5.0, line_div=18)
    img[h // 2:h // 2 + 300, 40:w - 60] = 255 - img[h // 2:h // 2 + 300, 40:w - 60]       # inverted-polarity decisions
    word = [{'text': 'w', 'confidence': 88}]
    lines = [{'bbox': [0, 0, w, h], 'words': word},                                   # the whole page as one box
             {'bbox': [30, 20, min(w, 1530), 420], 'words': word},                     # 1500 x 400 headline
             {'bbox': [10, 440, w - 3, 500], 'words': word},                           # long thin lines beside it
             {'bbox': [7, 505, w - 11, 566], 'words': word},
             {'bbox': [40, h // 2, w - 60, h // 2 + 300], 'words': word},              # the light-on-dark banner: 300 rows
             {'bbox': [w - 1100, h - 290, w, h], 'words': word},                       # touches the right / bottom border
             {'bbox': [100, 600, 400, 640], 'words': word}]                            # an ordinary small box in the same launch
    return img, [{'lines': lines}]


@pytest.mark.parametrize('mode', ['0', '1'])
@pytest.mark.parametrize('dpi', [None, 500, 1000])
def test_page_sized_hocr_boxes_take_the_wide_two_polarity_kernels(mode, dpi, monkeypatch):
    """a box launch whose boxes reach 1024 columns AND 256 rows switches to 8 columns per lane (16 for
    windows > 360) for a PAGE launch; a two-polarity (box) launch stays on 4 columns per lane for table-sized windows
    (<= 120) and takes the 8-column fp64 kernel beyond (dpi 500 -> window 125, dpi 1000 -> 251; square windows end at
    256: the 16-column kernel is out of a box launch's reach).  Table-driven by default, the fp64 sequence under
    MRCHIP_SAUVOLA_FAST=0.  create_hocr_mask (mrc.py:222-238: both thresholds, ratios, decisions) and the full page."""
    from mrchip import mrc
    monkeypatch.setenv('MRCHIP_SAUVOLA_FAST', mode)
    for (pw, ph, seed) in [(1700, 1100, 91), (3001, 1203, 92)]:
        img, hocr = _headline_page(pw, ph, seed)
        boxes = mrc.hocr_boxes(hocr, pw, ph)
        assert (boxes[:, 2] - boxes[:, 0]).max() >= 1024 and (boxes[:, 3] - boxes[:, 1]).max() >= 256
        m_gpu = np.zeros(img.shape, dtype=np.bool_)
        m_cpu = np.zeros(img.shape, dtype=np.bool_)
        mrc.create_hocr_mask(img, m_gpu, hocr, dpi=dpi)
        dec = []
        O.create_hocr_mask(img, m_cpu, boxes, dpi, dec)
        assert np.array_equal(m_gpu, m_cpu), (mode, dpi, pw, int((m_gpu != m_cpu).sum()))
        assert len(set(dec)) > 1, dec          # the boxes do not all decide the same way
        g = mrc.create_mrc_hocr_components(img, hocr, dpi=dpi, denoise_mask='fast', bg_downsample=3)
        e = O.create_mrc_hocr_components(img, hocr, dpi=dpi, denoise_mask='fast', bg_downsample=3)
        for k, (a, b) in enumerate(zip(g, e)):
            assert a.shape == b.shape and np.array_equal(a, b), (mode, dpi, pw, k)


def test_saturated_and_two_level_images():
    """variance 0 / tmp <= 0 boundary and the extreme variances: constant, two-level (0 / 255 halves, stripes, checker),
    saturated borders -- the corners of the decision table."""
    h, w = 200, 1200
    yy, xx = np.mgrid[0:h, 0:w]
    imgs = [np.full((h, w), v, np.uint8) for v in (0, 1, 2, 127, 128, 254, 255)]
    imgs += [np.where(xx < w // 2, 0, 255).astype(np.uint8), np.where((xx // 7 + yy // 5) % 2 == 0, 0, 255).astype(np.uint8),
             np.where(xx % 2 == 0, 0, 255).astype(np.uint8), np.where(yy % 3 == 0, 255, 0).astype(np.uint8),
             np.where((xx < 40) | (xx >= w - 40) | (yy < 40) | (yy >= h - 40), 0, 230).astype(np.uint8),
             np.where((xx - 600) ** 2 + (yy - 100) ** 2 < 70 ** 2, 255, 3).astype(np.uint8)]
    for k in (0.34, 0.1):
        for i, img in enumerate(imgs):
            got, exp = run_gpu(img, 51, 51, k), run_cpu(img, 51, 51, k)
            assert np.array_equal(got, exp), (k, i, int((got != exp).sum()))
