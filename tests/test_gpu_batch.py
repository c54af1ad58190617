"""GPU parity of the batch API: N pages in one batch == N single-page runs == oracle."""
import numpy as np
import pytest

import mrc_oracle as O
from mrchip import mrc, synth

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
