"""GPU tests of the streaming page pipeline (mrc.decompose_stream): batches rotate through upload / compute /
download on separate HIP streams; every yielded array is compared with the oracle."""
import numpy as np
import pytest

import mrc_oracle as O
from mrchip import _lib, mrc, synth

pytestmark = pytest.mark.gpu


def _expect(img, hocr, **kw):
    g = O.create_mrc_hocr_components(img, hocr, denoise_mask='fast', **kw)
    return next(g).copy(), next(g), next(g)


def test_stream_equals_oracle_in_order_with_short_last_batch():
    pages = [synth.synth_page(420, 300, 3, seed=300 + i, noise_sigma=[6.0, 0.0, 11.0][i % 3], line_div=12)
             for i in range(11)]
    got = []
    for res in mrc.decompose_stream(iter(pages), bg_downsample=3, batch_pages=4, copy=True):   # batches of 4, 4, 3
        got.append(res)
    assert len(got) == len(pages)
    for (img, hocr), (m, fg, bg) in zip(pages, got):
        em, ef, eb = _expect(img, hocr, bg_downsample=3)
        assert m.dtype == np.bool_ and np.array_equal(m, em), int((m != em).sum())
        assert fg.shape == ef.shape and np.array_equal(fg, ef)
        assert bg.shape == eb.shape and np.array_equal(bg, eb)


def test_stream_views_stay_valid_for_a_batch_and_packed_masks():
    pages = [synth.synth_page(333, 211, 1, seed=330 + i, noise_sigma=5.0, line_div=9) for i in range(9)]
    n = 0
    window = []          # (page index, views) of the most recent `batch_pages` results: still valid by contract
    for res in mrc.decompose_stream(iter(pages), dpi=200, bg_downsample=2, fg_downsample=2, batch_pages=3,
                                    mask_format='packed'):
        window.append((n, res))
        if len(window) > 3:
            window.pop(0)
        for i, (m, fg, bg) in window:
            em, ef, eb = _expect(pages[i][0], pages[i][1], dpi=200, bg_downsample=2, fg_downsample=2)
            assert m.shape == (211, (333 + 7) // 8) and np.array_equal(m, np.packbits(em, axis=1)), i
            assert np.array_equal(fg, ef) and np.array_equal(bg, eb), i
        n += 1
    assert n == 9


def test_stream_mixed_sizes_modes_and_pinned_inputs():
    ctx = _lib.default_context()
    specs = [(320, 200, 3), (320, 200, 3), (200, 320, 1), (320, 200, 3), (64, 48, 3), (64, 48, 3), (64, 48, 3),
             (200, 320, 1)]
    pages = []
    for i, (w, h, c) in enumerate(specs):
        img, hocr = synth.synth_page(w, h, c, seed=390 + i, noise_sigma=4.0, line_div=9)
        if i % 2 == 0:       # every other page lives in page-locked memory: asynchronous upload path
            pin = ctx.pinned_empty(img.shape)
            pin[...] = img
            assert _lib.is_pinned(pin) and not _lib.is_pinned(img)
            img = pin
        pages.append((img, hocr))
    got = list(mrc.decompose_stream(iter(pages), bg_downsample=2, batch_pages=2, copy=True, ctx=ctx))
    assert len(got) == len(pages)
    for (img, hocr), (m, fg, bg) in zip(pages, got):
        em, ef, eb = _expect(np.array(img), hocr, bg_downsample=2)
        assert np.array_equal(m, em) and np.array_equal(fg, ef) and np.array_equal(bg, eb)
    # decompose_pages (sorted by geometry, results back in input order) gives the same arrays
    res = mrc.decompose_pages([p[0] for p in pages], [p[1] for p in pages], bg_downsample=2, batch_pages=3)
    for a, b in zip(res, got):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_stream_rejects_bad_options_and_closes_on_abandon():
    img, hocr = synth.synth_page(100, 80, 1, seed=1, line_div=6)
    with pytest.raises(ValueError):
        next(mrc.decompose_stream([(img, hocr)], denoise_mask='bregman'))
    with pytest.raises(ValueError):
        next(mrc.decompose_stream([(img, hocr)], mask_format='bits'))
    g = mrc.decompose_stream([(img, hocr)] * 7, batch_pages=2)
    next(g)
    g.close()            # generator abandoned mid-stream: slots are released, no error
    m, fg, bg = next(mrc.decompose_stream([(img, hocr)], batch_pages=8))      # one page in a batch sized for 8
    em, ef, eb = _expect(img, hocr)
    assert np.array_equal(m, em) and np.array_equal(fg, ef) and np.array_equal(bg, eb)


def test_one_batch_object_serves_growing_page_counts():
    """regression (found by tests/fuzz_parity.py): a batch sized for 3 pages first used with 1 page, then with 2, then 3:
    the layer planes are allocated for the capacity, not for the pages in use at first call"""
    ctx = _lib.default_context()
    pages = [synth.synth_page(658, 222, 3, seed=500 + i, noise_sigma=[0.0, 11.0, 5.0][i], line_div=12) for i in range(3)]
    bt = mrc.Batch(ctx, 3, 658, 222, 3)
    for sel in ([0], [1, 2], [0, 1, 2], [2]):
        for j, i in enumerate(sel):
            bt.upload(j, pages[i][0])
            bt.set_boxes(j, mrc.hocr_boxes(pages[i][1], 658, 222))
        bt.set_count(len(sel))
        bt.mask_begin(51)
        bt.mask_finish(bt.sigmas(), True)
        fgs, bgs, _ = bt.layers(None, 3)
        for j, i in enumerate(sel):
            em, ef, eb = _expect(pages[i][0], pages[i][1], bg_downsample=3)
            assert np.array_equal(bt.download_mask(j), em)
            assert np.array_equal(bt.download_layer(j, 0, fgs), ef) and np.array_equal(bt.download_layer(j, 1, bgs), eb)
    bt.close()


def test_book_of_many_page_sizes_stays_inside_the_pool_budget():
    """one idle 8-page slot per distinct geometry does not scale to scanned books whose pages all differ in size.
    Slots are sized to the run they serve and idle slots of other sizes are closed under a byte budget -- results unchanged."""
    pages = [synth.synth_page(300 + 7 * i, 220 + 3 * (i % 5), 3 if i % 3 else 1, seed=500 + i, noise_sigma=5.0, line_div=10)
             for i in range(24)]
    pages += [pages[3]] * 5                                 # and a run of equal pages at the end
    budget = 6 * mrc._slot_bytes(1, 480, 240, 3)
    with mrc.StreamPool(max_bytes=budget) as pool:
        peak = 0
        got = []
        for res in mrc.decompose_stream(iter(pages), bg_downsample=2, batch_pages=4, copy=True, pool=pool):
            got.append(res)
            peak = max(peak, pool.bytes_held())
        assert len(got) == len(pages)
        # four slots are in flight at any time (they cannot be closed); nothing beyond them piles up
        assert len(pool.every) <= 8 and peak <= budget + 4 * mrc._slot_bytes(4, 480, 240, 3), (len(pool.every), peak, budget)
        assert all(sl.key[0] <= 4 for sl in pool.every)
    for (img, hocr), (m, fg, bg) in zip(pages, got):
        em, ef, eb = _expect(img, hocr, bg_downsample=2)
        assert np.array_equal(m, em) and np.array_equal(fg, ef) and np.array_equal(bg, eb)
