"""GPU parity of the full page pipeline: mrchip.mrc.create_mrc_hocr_components against the
reference goldens, the oracle, and the reference's digests at BASELINE.json config sizes."""
import numpy as np
import pytest

import mrc_oracle as O
from mrchip import mrc, synth
from helpers import load_npz, load_digests, unpack, sha

pytestmark = pytest.mark.gpu


def run_page(img, hocr, **kw):
    td, er = [], set()
    g = mrc.create_mrc_hocr_components(img, hocr, timing_data=td, errors=er, **kw)
    mask = next(g)
    fg = next(g)
    bg = next(g)
    with pytest.raises(StopIteration):
        next(g)
    return mask, fg, bg, [k for k, _ in td], sorted(er)


def test_pages_golden():
    z, meta = load_npz('pages.npz')
    for i, m in enumerate(meta):
        w, h, ch, seed, ns, dpi, ds, bgd, fgd, dn = m['case']
        img, hocr = synth.synth_page(w * (ds or 1), h * (ds or 1), ch, seed=seed, noise_sigma=ns, line_div=16)
        if ds:
            img = np.ascontiguousarray(img[::ds, ::ds])
        assert sha(img) == str(z['pg_img_sha_%d' % i])
        mask, fg, bg, keys, errs = run_page(img, hocr, dpi=dpi, downsample=ds, bg_downsample=bgd, fg_downsample=fgd,
                                            denoise_mask=dn)
        assert mask.dtype == np.bool_ and mask.shape == (h, w)
        exp = unpack(z['pg_mask_%d' % i], w)
        assert np.array_equal(mask, exp), (i, int((mask != exp).sum()))
        assert fg.shape == z['pg_fg_%d' % i].shape and np.array_equal(fg, z['pg_fg_%d' % i]), i
        assert bg.shape == z['pg_bg_%d' % i].shape and np.array_equal(bg, z['pg_bg_%d' % i]), i
        assert keys == m['keys'], (keys, m['keys'])
        assert errs == m['errors']


@pytest.mark.parametrize('w,h,ch,seed,ns,dpi,bgd,fgd', [
    (1000, 800, 3, 11, 6.0, None, 3, None), (1111, 777, 1, 12, 3.0, 200, 2, 2), (640, 480, 3, 13, 0.0, None, None, None),
    (2000, 1500, 3, 14, 12.0, 300, 3, 4), (517, 389, 3, 15, 40.0, None, 3, None),
])
def test_pages_vs_oracle(w, h, ch, seed, ns, dpi, bgd, fgd):
    img, hocr = synth.synth_page(w, h, ch, seed=seed, noise_sigma=ns, line_div=24)
    mask, fg, bg, keys, errs = run_page(img, hocr, dpi=dpi, bg_downsample=bgd, fg_downsample=fgd, denoise_mask='fast')
    g = O.create_mrc_hocr_components(img, hocr, dpi=dpi, bg_downsample=bgd, fg_downsample=fgd, denoise_mask='fast')
    em, ef, eb = next(g).copy(), next(g), next(g)
    assert np.array_equal(mask, em), int((mask != em).sum())
    assert fg.shape == ef.shape and np.array_equal(fg, ef)
    assert bg.shape == eb.shape and np.array_equal(bg, eb)


def test_generator_is_lazy_and_validates_options():
    img, hocr = synth.synth_page(320, 240, 3, seed=1, line_div=12)
    g = mrc.create_mrc_hocr_components(img, hocr, denoise_mask='fast', bg_downsample=3)
    m = next(g)            # --bw-pdf pulls only the mask (recode.py:400-408)
    assert m.dtype == np.bool_
    del g
    with pytest.raises(ValueError):
        next(mrc.create_mrc_hocr_components(img, hocr, denoise_mask=None))     # mrc.py:396
    errors = set()
    g = mrc.create_mrc_hocr_components(img, [], denoise_mask='none', bg_downsample=1000, errors=errors)
    next(g), next(g)
    bg = next(g)
    assert bg.shape == (240, 320, 3) and errors == {'too-small-to-downsample'}


def test_pil_image_input():
    from PIL import Image
    img, hocr = synth.synth_page(400, 300, 3, seed=2, line_div=12)
    a = run_page(Image.fromarray(img), hocr, denoise_mask='fast', bg_downsample=3)
    b = run_page(img, hocr, denoise_mask='fast', bg_downsample=3)
    for x, y in zip(a[:3], b[:3]):
        assert np.array_equal(x, y)


def test_config2_digest():
    """BASELINE.json configs[1]: 4000x3000 RGB + hOCR, bg/3 -- outputs must hash to what the
    reference itself produced (tests/golden/digests.json)."""
    d = load_digests()
    for tag, dpi in (('c2_dpiNone', None), ('c2_dpi400', 400)):
        img, hocr = synth.synth_page(4000, 3000, 3, seed=202, noise_sigma=6.0, line_div=60)
        assert sha(img) == d[tag]['in']
        mask, fg, bg, keys, errs = run_page(img, hocr, dpi=dpi, bg_downsample=3, denoise_mask='fast')
        assert int(mask.sum()) == d[tag]['mask_sum']
        assert sha(mask) == d[tag]['mask']
        assert sha(fg) == d[tag]['fg']
        assert list(bg.shape) == d[tag]['bg_shape'] and sha(bg) == d[tag]['bg']
        assert keys == d[tag]['keys']


def test_config5_digest():
    """BASELINE.json configs[4] shape: 8000x6000 RGB, dpi 364 (window 91), fg and bg /4."""
    d = load_digests()['c5']
    img, hocr = synth.synth_page(8000, 6000, 3, seed=505, noise_sigma=6.0, line_div=60)
    assert sha(img) == d['in']
    mask, fg, bg, keys, errs = run_page(img, hocr, dpi=364, bg_downsample=4, fg_downsample=4, denoise_mask='fast')
    assert sha(mask) == d['mask'] and int(mask.sum()) == d['mask_sum']
    assert list(fg.shape) == d['fg_shape'] and sha(fg) == d['fg']
    assert list(bg.shape) == d['bg_shape'] and sha(bg) == d['bg']


def test_config5_batch_of_pages_streamed():
    """configs[4] as a BATCH (only one 8000x6000 page went through pytest): six pages, both reference-digested
    seeds, through decompose_stream in batches of four -- the launch geometry of `bench.py --config c5` (optimise on rows
    of 8000 columns, the box `reduce` + bicubic thumbnails of both layers) -- every output of every page digest-checked."""
    dg = load_digests()
    want = {505: dg['c5'], 506: dg['c5_506']}
    made = {s: synth.synth_page(8000, 6000, 3, seed=s, noise_sigma=6.0, line_div=60) for s in (505, 506)}
    for s in want:
        assert sha(made[s][0]) == want[s]['in']
    order = [505, 506, 506, 505, 506, 505]
    n = 0
    for s, (m, fg, bg) in zip(order, mrc.decompose_stream(((made[s][0], made[s][1]) for s in order), dpi=364, bg_downsample=4,
                                                          fg_downsample=4, batch_pages=4, copy=True)):
        d = want[s]
        assert sha(m) == d['mask'] and int(m.sum()) == d['mask_sum'], (n, s)
        assert list(fg.shape) == d['fg_shape'] and sha(fg) == d['fg'], (n, s)
        assert list(bg.shape) == d['bg_shape'] and sha(bg) == d['bg'], (n, s)
        n += 1
    assert n == len(order)
