"""Scan-like and adversarial pages (tests/golden/scans.npz: inputs stored, outputs made by the REAL reference through
tests/golden/make_golden.py scans): JPEG-decoded text rendered from a bitmap font, a photo-like region, a black scanner
border, a white-on-black block under hOCR boxes, ink in every row, line pitch below the bg radius, a gray scan, constant
and two-level pages.  The oracle is checked against them on the CPU, the HIP path on the GPU (default schedule and the
band walkers of whole-row workgroups forced), plus the extremes at config-2 size against reference digests."""
import numpy as np
import pytest

import mrc_oracle as O
from helpers import load_npz, load_digests, unpack, sha


def _cases():
    z, meta = load_npz('scans.npz')
    for i, m in enumerate(meta):
        img = z['sc_img_%d' % i]
        yield m, img, unpack(z['sc_mask_%d' % i], img.shape[1]), z['sc_fg_%d' % i], z['sc_bg_%d' % i]


def test_fixture_covers_the_families():
    names = [m['name'] for m, *_ in _cases()]
    assert names == ['jpeg_text', 'jpeg_photo', 'scanner_border', 'white_on_black', 'ink_every_row', 'tight_pitch', 'jpeg_gray',
                     'all0', 'all255', 'two_level']
    by = {m['name']: m for m, *_ in _cases()}
    assert by['ink_every_row']['ink_rows'] == by['ink_every_row']['h']           # no ink-free row at all
    assert by['all255']['mask_sum'] == 0 and by['all0']['mask_sum'] == 300 * 400


def test_oracle_equals_the_reference_on_scan_like_pages():
    for m, img, mask, fg, bg in _cases():
        g = O.create_mrc_hocr_components(img, m['hocr'], bg_downsample=3, denoise_mask='fast')
        em, ef, eb = next(g).copy(), next(g), next(g)
        assert np.array_equal(em, mask), (m['name'], int((em != mask).sum()))
        assert ef.shape == fg.shape and np.array_equal(ef, fg), m['name']
        assert eb.shape == bg.shape and np.array_equal(eb, bg), m['name']


@pytest.mark.gpu
@pytest.mark.parametrize('strips', [None, '0'])
def test_hip_path_equals_the_reference_on_scan_like_pages(strips, monkeypatch):
    from PIL import Image
    from mrchip import mrc
    if strips is not None:
        monkeypatch.setenv('MRCHIP_OPT_STRIPS', strips)        # whole rows: the band walkers
    for m, img, mask, fg, bg in _cases():
        td, er = [], set()
        g = mrc.create_mrc_hocr_components(Image.fromarray(img), m['hocr'], bg_downsample=3, denoise_mask='fast', timing_data=td,
                                           errors=er)
        gm, gf, gb = next(g), next(g), next(g)
        assert np.array_equal(gm, mask), (m['name'], int((gm != mask).sum()))
        assert gf.shape == fg.shape and np.array_equal(gf, fg), (m['name'], int((gf != fg).sum()))
        assert gb.shape == bg.shape and np.array_equal(gb, bg), (m['name'], int((gb != bg).sum()))
        assert [k for k, _ in td] == m['keys'] and sorted(er) == m['errors'], m['name']


@pytest.mark.gpu
def test_scan_like_pages_as_one_batch_and_as_a_stream():
    """The same pages through the batch object (every stage one launch over all pages of one size) and through
    decompose_stream."""
    from mrchip import mrc
    cases = [c for c in _cases() if c[1].shape == (600, 440, 3)]
    imgs = [c[1] for c in cases]
    hocrs = [c[0]['hocr'] for c in cases]
    res = list(mrc.decompose_stream(zip(imgs, hocrs), bg_downsample=3, denoise_mask='fast', copy=True))
    assert len(res) == len(cases)
    for (m, img, mask, fg, bg), r in zip(cases, res):
        assert np.array_equal(np.asarray(r[0]).astype(bool), mask), m['name']
        assert np.array_equal(r[1], fg), m['name']
        assert np.array_equal(r[2], bg), m['name']


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['c2_all0', 'c2_all255', 'c2_two_level'])
def test_extreme_pages_at_config2_size(name):
    from mrchip import mrc, synth
    d = load_digests()[name]
    arr = {'c2_all0': lambda: np.zeros((3000, 4000, 3), np.uint8), 'c2_all255': lambda: np.full((3000, 4000, 3), 255, np.uint8),
           'c2_two_level': lambda: synth.two_level_page(4000, 3000)}[name]()
    assert sha(arr) == d['in']
    hocr = [{'lines': [{'bbox': [200, 300, 3800, 420], 'words': [{'text': 'x', 'confidence': 95}]}]}]
    g = mrc.create_mrc_hocr_components(arr, hocr, dpi=None, bg_downsample=3, denoise_mask='fast')
    m, fg, bg = next(g), next(g), next(g)
    assert int(m.sum()) == d['mask_sum'] and sha(m) == d['mask']
    assert sha(fg) == d['fg']
    assert list(bg.shape) == d['bg_shape'] and sha(bg) == d['bg']
