"""GPU parity on edge cases: tiny pages, pages smaller than the window, widths that are not a
multiple of any vector width, empty / degenerate hOCR, non-contiguous and PIL inputs, API misuse."""
import numpy as np
import pytest

import mrc_oracle as O
from mrchip import _lib, mrc, optimiser, sauvola, synth

pytestmark = pytest.mark.gpu


def both(img, hocr, **kw):
    g = mrc.create_mrc_hocr_components(img, hocr, **kw)
    got = [next(g).copy(), next(g), next(g)]
    e = O.create_mrc_hocr_components(img, hocr, **kw)
    exp = [next(e).copy(), next(e), next(e)]
    return got, exp


@pytest.mark.parametrize('w,h,ch', [(1, 1, 1), (1, 1, 3), (2, 3, 3), (5, 5, 1), (7, 4, 3), (13, 31, 3), (51, 51, 1),
                                    (64, 3, 3), (3, 64, 1), (255, 17, 3), (257, 65, 3), (1021, 9, 1)])
def test_tiny_and_odd_pages(w, h, ch):
    rng = np.random.RandomState(w * 131 + h)
    img = rng.randint(0, 256, (h, w) if ch == 1 else (h, w, 3)).astype(np.uint8)
    got, exp = both(img, [], denoise_mask='fast', bg_downsample=3, fg_downsample=2)
    for a, b in zip(got, exp):
        assert a.shape == b.shape and np.array_equal(a, b), (w, h, ch)


def test_boxes_touching_borders_and_degenerate():
    img, _ = synth.synth_page(300, 200, 1, seed=3, noise_sigma=4.0, line_div=10)
    hocr = [{'lines': [
        {'bbox': [0, 0, 300, 200], 'words': [{'text': 'a', 'confidence': 90}]},          # the whole page
        {'bbox': [0, 0, 1, 1], 'words': [{'text': 'a', 'confidence': 90}]},              # 1x1
        {'bbox': [299, 199, 300, 200], 'words': [{'text': 'a', 'confidence': 90}]},      # last pixel
        {'bbox': [10, 10, 10, 50], 'words': [{'text': 'a', 'confidence': 90}]},          # zero width: skipped
        {'bbox': [50, 60, 40, 70], 'words': [{'text': 'a', 'confidence': 90}]},          # inverted: skipped + message
        {'bbox': [-5, 10, 40, 30], 'words': [{'text': 'a', 'confidence': 90}]},          # outside: skipped + message
        {'bbox': [10.9, 20.2, 290.7, 45.9], 'words': [{'text': 'a', 'confidence': 19.9}]},   # low confidence
        {'bbox': [10.9, 50.2, 290.7, 75.9], 'words': []},                                 # no words -> conf 0
        {'bbox': [10.9, 80.2, 290.7, 120.9], 'words': [{'text': 'x', 'confidence': 20}]},  # float coords truncated
        {'bbox': [100, 90, 250, 140], 'words': [{'text': 'x', 'confidence': 50}]},        # overlaps the previous
    ]}]
    got, exp = both(img, hocr, denoise_mask='fast', bg_downsample=2)
    for a, b in zip(got, exp):
        assert a.shape == b.shape and np.array_equal(a, b)
    got, exp = both(np.ascontiguousarray(img[::2, ::2]), hocr, downsample=2, denoise_mask='none')
    for a, b in zip(got, exp):
        assert np.array_equal(a, b)


def test_non_contiguous_and_pil_modes():
    from PIL import Image
    base, hocr = synth.synth_page(260, 180, 3, seed=9, line_div=9)
    view = base[::-1, ::-1]                                    # negative strides
    a = [x for x in mrc.create_mrc_hocr_components(view, [], denoise_mask='fast')]
    b = [x for x in O.create_mrc_hocr_components(np.ascontiguousarray(view), [], denoise_mask='fast')]
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    pal = Image.fromarray(base).convert('P')                   # mapped image -> RGB (mrc.py:401-404)
    rgb = np.array(pal.convert('RGB'))
    a = [x for x in mrc.create_mrc_hocr_components(pal, [], denoise_mask='fast')]
    b = [x for x in O.create_mrc_hocr_components(rgb, [], denoise_mask='fast')]
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    # threshold_image on a strided crop, like mrc.py:223-230
    gray = O.luma601(base)
    crop = gray[20:90, 33:201]
    assert np.array_equal(mrc.threshold_image(crop, 150, 0.1), O.threshold_image(np.ascontiguousarray(crop), 150, 0.1))


def test_kernel_entry_points_reject_bad_arguments():
    with pytest.raises(ValueError):
        sauvola.binarise_sauvola(np.zeros((4, 4), np.uint8), np.zeros(16, np.uint8), 4, 4, 3, 3, 0.3, 128)   # 2-D in_arr
    with pytest.raises(ValueError):
        sauvola.binarise_sauvola(np.zeros(16, np.float32), np.zeros(16, np.uint8), 4, 4, 3, 3, 0.3, 128)     # dtype
    with pytest.raises(ValueError):
        optimiser.optimise_rgb2(np.zeros((4, 4), bool), np.zeros((4, 4), np.uint8), 4, 4, 3)                 # ndim
    with pytest.raises(_lib.MrchipError):
        optimiser.optimise_gray2(np.zeros((4, 4), bool), np.zeros((4, 4), np.uint8), 4, 4, 99)               # n > 32
    lib, ctx = _lib.load(), _lib.default_context()
    pg = lib.mrchip_page_create(ctx.handle, 16, 16, 1)
    assert pg
    import ctypes as C
    assert lib.mrchip_page_sigma(pg, C.byref(C.c_double())) == -5                 # MRCHIP_E_STATE: nothing uploaded
    assert lib.mrchip_page_layer(pg, 0, 0.0, None, None, None) == -5
    assert b'before' in lib.mrchip_last_error()
    lib.mrchip_page_destroy(pg)
    assert not lib.mrchip_page_create(ctx.handle, 0, 16, 1)
    assert not lib.mrchip_page_create(ctx.handle, 16, 16, 4)


def test_optimise_generic_path_and_large_n():
    rng = np.random.RandomState(4)
    for (h, w, n) in [(60, 300, 15), (40, 257, 32), (30, 100, 12)]:     # n > 11: the unpacked kernel
        m = rng.rand(h, w) < 0.2
        c = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        g = rng.randint(0, 256, (h, w)).astype(np.uint8)
        assert np.array_equal(optimiser.optimise_rgb2(m, c, w, h, n), O.optimise_rgb2(m, c, w, h, n)), n
        assert np.array_equal(optimiser.optimise_gray2(m, g, w, h, n), O.optimise_gray2(m, g, w, h, n)), n


@pytest.mark.parametrize('radius', [1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_gaussian_every_fused_radius_and_ragged_widths(radius):
    """scipy radius = int(4 sigma + 0.5); radii 1..8 take the fused tile kernel (halo of one or two float4s per
    lane), 9 the two-pass fallback.  Widths straddle the 256-column tile and the 4-column lane group."""
    lib, ctx = _lib.load(), _lib.default_context()
    sig = (radius - 0.5) / 4.0 + 0.06
    wts, r = mrc.gaussian_weights(sig)
    assert r == radius
    rng = np.random.RandomState(radius)
    for (h, w) in [(33, 16), (40, 255), (65, 257), (31, 259), (70, 513), (16, 1001)]:
        g = rng.randint(0, 256, (h, w)).astype(np.uint8)
        out = np.empty_like(g)
        _lib.check(lib.mrchip_gaussian_u8(ctx.handle, _lib.ptr(g), _lib.ptr(out), w, h, sig, _lib.ptr(wts, _lib.f64p), r))
        exp = O.gaussian_filter(g.astype(np.float32), sig, weights=wts).astype(np.uint8)
        assert np.array_equal(out, exp), (radius, h, w)


@pytest.mark.parametrize('no_mfma', [False, True])
def test_thumbnail_geometries_matrix_core_and_fallback(no_mfma, monkeypatch):
    """The bicubic passes run on the MFMA units when both directions resize and a 16-output tile spans at most
    128 input bytes (k_resample.hip resize_mm_kernel), otherwise -- or with MRCHIP_THUMB_NO_MFMA -- on the integer
    VALU kernels.  Scale factors 1.3 .. 4 (+ Image.reduce beyond), gray and RGB, sizes that leave partial tiles
    (outputs % 16), partial line quads (lines % 64) and odd row lengths."""
    import ctypes as C
    if no_mfma:
        monkeypatch.setenv('MRCHIP_THUMB_NO_MFMA', '1')
    lib, ctx = _lib.load(), _lib.default_context()
    rng = np.random.RandomState(5)
    cases = [((131, 257, 3), (100, 60)), ((700, 1000), (333, 233)), ((389, 515, 3), (129, 97)), ((70, 90, 3), (45, 35)),
             ((1000, 64), (21, 333)), ((33, 2050, 3), (683, 11)), ((640, 480, 3), (369, 492)), ((1203, 901, 3), (112, 150)),
             ((256, 256), (64, 64)), ((257, 255, 3), (85, 86))]
    for shape, (rw, rh) in cases:
        im = rng.randint(0, 256, shape).astype(np.uint8)
        h, w = shape[:2]
        c = 1 if im.ndim == 2 else 3
        ow, oh = C.c_int(), C.c_int()
        lib.mrchip_thumbnail_size(w, h, rw, rh, C.byref(ow), C.byref(oh))
        out = np.empty((oh.value, ow.value) if c == 1 else (oh.value, ow.value, 3), np.uint8)
        _lib.check(lib.mrchip_thumbnail(ctx.handle, _lib.ptr(np.ascontiguousarray(im)), w, h, c, rw, rh, _lib.ptr(out)))
        exp = O.thumbnail(im, rw, rh)
        assert out.shape == exp.shape and np.array_equal(out, exp), (shape, rw, rh, no_mfma)


def test_lanczos_ingest_downsample_matches_pillow_vectors_and_oracle():
    """recode.py:368-372: image.thumbnail((w/ds, h/ds), resample=Image.LANCZOS, reducing_gap=None) -- the same
    resample machinery with the Lanczos3 table (matrix-core path up to scale ~3, integer kernels beyond)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'lanczos.npz'))
    for m in z['meta']:
        i, ds, flt, gap, rw, rh = str(m).split('|')
        a = z['in_' + i]
        h, w = a.shape[:2]
        got = mrc.thumbnail(a, (w / float(ds), h / float(ds)), resample=flt, reducing_gap=None if gap == 'None' else float(gap))
        assert got.shape == z['out_' + i].shape and np.array_equal(got, z['out_' + i]), m
    rng = np.random.RandomState(8)
    for shape, ds in [((1000, 1500, 3), 2), ((999, 1333), 3), ((800, 1200, 3), 4), ((517, 333, 3), 5.5), ((100, 3000, 3), 2)]:
        a = rng.randint(0, 256, shape).astype(np.uint8)
        h, w = shape[:2]
        got = mrc.thumbnail(a, (w / ds, h / ds), resample='lanczos', reducing_gap=None)
        exp = O.thumbnail_ex(a, int(w / ds), int(h / ds), 'lanczos', None)
        assert got.shape == exp.shape and np.array_equal(got, exp), (shape, ds)
    # fuzzer case: Image.reduce(2) then Lanczos on a 231-pixel-wide intermediate -- the line panel of the last
    # workgroup used to be loaded past the end of the (small) scratch image
    for _ in range(20):
        a = rng.randint(0, 256, (695, 461, 3)).astype(np.uint8)
        got = mrc.thumbnail(a, (461 / 5.0, 695 / 5.0), resample='lanczos', reducing_gap=2.0)
        exp = O.thumbnail_ex(a, 92, 139, 'lanczos', 2.0)
        assert np.array_equal(got, exp)


@pytest.mark.parametrize('seed,w,h', [(1, 640, 420), (2, 640, 420), (3, 640, 420), (4, 1531, 530), (5, 1290, 777)])
def test_many_overlapping_boxes_later_box_wins_under_the_page_threshold(seed, w, h):
    """mask[t:b, l:r] = th box after box (mrc.py:266), then mask |= page threshold (mrc.py:329).  The device
    stores the page threshold first and ORs each pixel's LAST deciding box on top; random heavily overlapping
    boxes (both polarities, some undecided) must give the same mask.  The pages of 1024 columns and more take the
    schedule in which Sauvola and the commit also write the denoiser's 1-bpp rows (widths not multiples of 8 / 32:
    partial bytes and words at the right edge, box edges inside 16-pixel groups)."""
    rng = np.random.RandomState(seed)
    img, _ = synth.synth_page(w, h, 3, seed=40 + seed, noise_sigma=5.0, line_div=14)
    img[200:330, 60:600] = 255 - img[200:330, 60:600]            # a light-on-dark block: inverted-polarity decisions
    lines = []
    for _ in range(14 if w < 1024 else 22):
        l, t = int(rng.randint(0, w - 80)), int(rng.randint(0, h - 40))
        r, b = min(w, l + int(rng.randint(40, 400 if w < 1024 else 900))), min(h, t + int(rng.randint(12, 120)))
        lines.append({'bbox': [l, t, r, b], 'words': [{'text': 'w', 'confidence': 80}]})
    hocr = [{'lines': lines}]
    got, exp = both(img, hocr, denoise_mask='fast', bg_downsample=3)
    for a, b in zip(got, exp):
        assert a.shape == b.shape and np.array_equal(a, b)
    got, exp = both(img, hocr, denoise_mask='none')
    assert np.array_equal(got[0], exp[0])


@pytest.mark.parametrize('w,c,n', [(4097, 3, 3), (5000, 3, 10), (8000, 3, 10), (8160, 1, 3), (8163, 3, 10), (8200, 1, 10), (6001, 3, 11), (5003, 3, 14)])
def test_optimise_rows_wider_than_4096_columns(w, c, n):
    """4097..8160 columns take the packed kernel with two column groups per thread; beyond that (or n > 11) the
    unpacked one.  Both polarities of the mask, ragged right edges."""
    rng = np.random.RandomState(w + n)
    h = 37
    img = rng.randint(0, 256, (h, w) if c == 1 else (h, w, 3)).astype(np.uint8)
    for density in (0.08, 0.92):
        mask = (rng.rand(h, w) < density).astype(np.uint8)
        f, e = (optimiser.optimise_gray2, O.optimise_gray2) if c == 1 else (optimiser.optimise_rgb2, O.optimise_rgb2)
        assert np.array_equal(f(mask, img, w, h, n), e(mask, img, w, h, n)), (w, c, n, density)


@pytest.mark.gpu
def test_guard_bands_of_the_device_allocator_see_a_stray_write_and_stay_clean_under_the_entry_points():
    """MRCHIP_CANARY pads every block of the caching allocator with a pattern beyond the slack the
    kernels may touch (they read and write row padding by design).  In a child process (the switch is read once): the
    deliberate one-byte writes of the self-test are seen (2), and a round of host-buffer entry points and a page batch on
    ragged shapes leaves every guard band intact."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, 'archive-pdf-tools_amd'))
import numpy as np
from mrchip import _lib, mrc, sauvola, optimiser, synth
ctx = _lib.default_context()
assert ctx.canary_selftest() == 2
rng = np.random.RandomState(5)
for (h, w) in ((1, 1), (33, 65), (257, 511), (64, 1027), (301, 299)):
    g = rng.randint(0, 256, (h, w)).astype(np.uint8)
    out = np.empty(h * w, np.uint8)
    for win in (3, 51, 101):
        sauvola.binarise_sauvola(g.ravel(), out, w, h, win, win, 0.34, 128.0)
    m = (rng.rand(h, w) < 0.3).astype(np.uint8)
    optimiser.optimise_gray2(m, g, w, h, 3); optimiser.optimise_gray2(m, g, w, h, 10)
    rgb = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    optimiser.optimise_rgb2(m, rgb, w, h, 10)
    optimiser.fast_mask_denoise(m.copy(), w, h, 4, 2)
    mrc.estimate_noise(g)
    mm = np.zeros((h, w), bool); mrc.create_threshold_mask(mm, g.astype(np.float32), dpi=None)
    if h > 8 and w > 8:
        mrc.thumbnail(rgb, (w / 3, h / 3))
    assert ctx.canary_check() == 0, (h, w)
img, hocr = synth.synth_page(703, 517, 3, seed=11, noise_sigma=6.0, line_div=14)
for _ in range(2):
    list(mrc.create_mrc_hocr_components(img, hocr, bg_downsample=3, denoise_mask='fast'))
res = mrc.decompose_pages([img, img], [hocr, hocr], denoise_mask='fast', bg_downsample=3, fg_downsample=2)
assert ctx.canary_check() == 0
print('canary ok')
''' % root
    env = dict(os.environ, MRCHIP_CANARY='64')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'canary ok' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    # and the switch off: no guards, the self-test reports nothing
    from mrchip import _lib
    if not os.environ.get('MRCHIP_CANARY'):
        assert _lib.default_context().canary_selftest() == 0


@pytest.mark.gpu
def test_sixteen_processes_on_one_gpu_return_exact_results():
    """Round 6's parity finding (profiles/r06_README.md): with 16 or more processes on one GPU the runtime's transfers to and
    from PAGEABLE host memory lost 4 KiB pages and delivered incomplete tables -- hundreds of wrong results per minute on
    large host-buffer calls, none with 8 processes, none once every such transfer goes through the library's own page-locked
    staging buffers (ctx.hip).  Sixteen processes of tests/fuzz_parity.py (large-window Sauvola calls against the oracle,
    diagnosis mode: mismatches are counted, the run goes on) for 20 s: not one mismatch.  (The direct path,
    MRCHIP_DIRECT_PAGEABLE=1, showed ~6 per second in this very setting.)"""
    import subprocess, sys, os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FUZZ_DIAG='1', FUZZ_FAMILIES='8')
    env.pop('MRCHIP_DIRECT_PAGEABLE', None)
    procs = [subprocess.Popen([sys.executable, os.path.join(root, 'tests', 'fuzz_parity.py'), '20', str(8800 + i)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for i in range(16)]
    outs = [p.communicate(timeout=400)[0] for p in procs]
    cases = 0
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and 'fuzz ok' in o, o[-1500:]
        assert 'DIAG' not in o, [l for l in o.split('\n') if l.startswith('DIAG')][:3]
        cases += int(re.search(r'(\d+) cases', o).group(1))
    assert cases > 50, cases
