"""GPU parity of the batch API: N pages in one batch == N single-page runs == oracle."""
import numpy as np
import pytest

import mrc_oracle as O
from mrchip import _lib, mrc, synth

pytestmark = pytest.mark.gpu


def test_batch_equals_oracle_per_page():
    pages = [synth.synth_page(700, 500, 3, seed=50 + i, noise_sigma=ns, line_div=20)
             for i, ns in enumerate([6.0, 0.0, 12.0, 2.0, 30.0])]       # blur / no blur / different radii
    res = mrc.decompose_pages([p[0] for p in pages], [p[1] for p in pages], bg_downsample=3, fg_downsample=2)
    for (img, hocr), (mask, fg, bg) in zip(pages, res):
        g = O.create_mrc_hocr_components(img, hocr, bg_downsample=3, fg_downsample=2, denoise_mask='fast')
        em, ef, eb = next(g).copy(), next(g), next(g)
        assert np.array_equal(mask, em), int((mask != em).sum())
        assert fg.shape == ef.shape and np.array_equal(fg, ef)
        assert bg.shape == eb.shape and np.array_equal(bg, eb)


def test_batch_gray_and_repeatable():
    pages = [synth.synth_page(513, 301, 1, seed=70 + i, noise_sigma=5.0, line_div=14) for i in range(3)]
    a = mrc.decompose_pages([p[0] for p in pages], [p[1] for p in pages], dpi=200, bg_downsample=4)
    b = mrc.decompose_pages([p[0] for p in pages], [p[1] for p in pages], dpi=200, bg_downsample=4)
    for (img, hocr), x, y in zip(pages, a, b):
        for u, v in zip(x, y):
            assert np.array_equal(u, v)
        g = O.create_mrc_hocr_components(img, hocr, dpi=200, bg_downsample=4, denoise_mask='fast')
        em, ef, eb = next(g).copy(), next(g), next(g)
        assert np.array_equal(x[0], em) and np.array_equal(x[1], ef) and np.array_equal(x[2], eb)


def test_packed_mask_matches_numpy_packbits_and_pil():
    """the mask leaves the device at 1 bpp in the layout mrc.encode_mrc_mask builds
    (PIL mode '1' == numpy.packbits rows, MSB first); widths that are not multiples of 8 / 32."""
    ctx = _lib.default_context()
    for (w, h, c) in [(403, 301, 3), (64, 40, 1), (1001, 77, 1), (7, 5, 3)]:
        pages = []
        for i in range(2):
            img, hocr = synth.synth_page(w, h, c, seed=50 + i, noise_sigma=5.0, line_div=12)
            pages.append((img, mrc.hocr_boxes(hocr, w, h)))
        bt = mrc.Batch(ctx, 2, w, h, c)
        for i, (img, boxes) in enumerate(pages):
            bt.upload(i, img)
            bt.set_boxes(i, boxes)
        bt.mask_begin(51)
        bt.mask_finish(bt.sigmas(), True)
        for i in range(2):
            mask = bt.download_mask(i)
            packed = bt.download_mask_packed(i)
            assert packed.shape == (h, (w + 7) // 8)
            assert np.array_equal(packed, np.packbits(mask, axis=1))
            pbm = mrc.packed_mask_to_pbm(packed, w, h)
            assert pbm.startswith(b'P4\n%d %d\n' % (w, h)) and len(pbm) == len(b'P4\n%d %d\n' % (w, h)) + packed.size
            try:
                from PIL import Image
            except ImportError:
                continue
            a = Image.frombytes('1', (w, h), packed.tobytes())
            b = Image.fromarray(mask)
            assert a.mode == b.mode == '1' and a.tobytes() == b.tobytes()
        bt.close()


def test_async_layer_downloads_into_pinned_memory_and_pnm_bytes():
    """layers leave through pinned buffers with enqueue-only copies (encode page i while
    page i+1 is in flight); layer_to_pnm == what PIL writes for the JPEG2000 encoders."""
    import io
    ctx = _lib.default_context()
    w, h = 333, 217
    for c in (1, 3):
        pages = [synth.synth_page(w, h, c, seed=70 + i, noise_sigma=5.0, line_div=10) for i in range(3)]
        bt = mrc.Batch(ctx, 3, w, h, c)
        for i, (img, hocr) in enumerate(pages):
            bt.upload(i, img)
            bt.set_boxes(i, mrc.hocr_boxes(hocr, w, h))
        bt.mask_begin(51)
        bt.mask_finish(bt.sigmas(), True)
        fgs, bgs, _ = bt.layers(None, 3)
        pinned = [(ctx.pinned_empty((fgs[1], fgs[0]) if c == 1 else (fgs[1], fgs[0], 3)),
                   ctx.pinned_empty((bgs[1], bgs[0]) if c == 1 else (bgs[1], bgs[0], 3))) for _ in range(3)]
        for i in range(3):                                   # all six copies enqueued, one wait
            bt.download_layer(i, 0, fgs, out=pinned[i][0], wait=False)
            bt.download_layer(i, 1, bgs, out=pinned[i][1], wait=False)
        bt.sync()
        for i in range(3):
            assert np.array_equal(pinned[i][0], bt.download_layer(i, 0, fgs))
            assert np.array_equal(pinned[i][1], bt.download_layer(i, 1, bgs))
            e = O.create_mrc_hocr_components(pages[i][0], pages[i][1], denoise_mask='fast', bg_downsample=3)
            next(e)
            assert np.array_equal(pinned[i][0], next(e)) and np.array_equal(pinned[i][1], next(e))
        pnm = mrc.layer_to_pnm(pinned[0][1])
        try:
            from PIL import Image
        except ImportError:
            Image = None
        if Image is not None:
            f = io.BytesIO()
            Image.fromarray(np.array(pinned[0][1])).save(f, format='PPM')
            assert f.getvalue() == pnm
        bt.close()


def test_decompose_pages_groups_mixed_sizes_and_modes():
    """a book's pages differ in size and mode: decompose_pages groups them into same-size batches and returns the
    results in input order"""
    specs = [(320, 200, 3), (200, 320, 1), (320, 200, 3), (64, 48, 3), (200, 320, 1), (320, 200, 1)]
    pages = [synth.synth_page(w, h, c, seed=90 + i, noise_sigma=4.0, line_div=9) for i, (w, h, c) in enumerate(specs)]
    res = mrc.decompose_pages([p[0] for p in pages], [p[1] for p in pages], bg_downsample=2, max_batch_bytes=3 << 20)
    for (img, hocr), got in zip(pages, res):
        e = O.create_mrc_hocr_components(img, hocr, denoise_mask='fast', bg_downsample=2)
        for a in got:
            b = next(e)
            assert a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize('c,bg_ds,fg_ds', [(3, 3, None), (1, None, None), (3, 2, 2)])
def test_large_batch_takes_the_band_queue(c, bg_ds, fg_ds):
    """More page-layers than half the CUs: optimise runs on the band walkers (a persistent workgroup per CU on a
    device-built queue) instead of column strips -- the schedule of the 128-page bench batches, here with 132 small pages
    (six distinct ones, with and without ink-free gaps) against the oracle.  bg_downsample set: the bg layer holds its
    bands only and the one-kernel thumbnail reads the other rows from the image."""
    ctx = _lib.default_context()
    w, h, n = 320, 240, 132
    distinct = [synth.synth_page(w, h, c, seed=900 + i, noise_sigma=ns, line_div=ld)
                for i, (ns, ld) in enumerate([(6.0, 12), (0.0, 20), (12.0, 8), (3.0, 30)])]
    dense = distinct[0][0].copy()
    dense[:, 40:43] = 25                                                                   # a rule down the page: ink in every row, one band
    blank = np.full((h, w, 3) if c == 3 else (h, w), 231, np.uint8)                        # no ink: a layer of copies
    distinct += [(dense, distinct[0][1]), (blank, [])]
    bt = mrc.Batch(ctx, n, w, h, c)
    for i in range(n):
        img, hocr = distinct[i % len(distinct)]
        bt.upload(i, img)
        bt.set_boxes(i, mrc.hocr_boxes(hocr, w, h))
    bt.mask_begin(51)
    bt.mask_finish(bt.sigmas(), True)
    fg_size, bg_size, _ = bt.layers(fg_ds, bg_ds)
    exp = []
    for img, hocr in distinct:
        g = O.create_mrc_hocr_components(img, hocr, bg_downsample=bg_ds, fg_downsample=fg_ds, denoise_mask='fast')
        exp.append((next(g).copy(), next(g), next(g)))
    for i in list(range(12)) + [63, 64, 65, 126, 127, 128, 131]:
        em, ef, eb = exp[i % len(distinct)]
        assert np.array_equal(bt.download_mask(i), em), i
        fg = bt.download_layer(i, 0, fg_size)
        bg = bt.download_layer(i, 1, bg_size)
        assert fg.shape == ef.shape and np.array_equal(fg, ef), (i, int((fg != ef).sum()))
        assert bg.shape == eb.shape and np.array_equal(bg, eb), (i, int((bg != eb).sum()))
    bt.close()
