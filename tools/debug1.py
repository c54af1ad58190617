import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'archive-pdf-tools_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import mrc_oracle as O
from mrchip import _lib, mrc, optimiser, synth
from helpers import kernel_cases, unpack
z, cases = kernel_cases('optimise')
for _, i, h, w, n, _a, _b in cases:
    m = unpack(z['opt_mask_%d' % i], w); c = z['opt_c_%d' % i]
    got = optimiser.optimise_rgb2(m, c, w, h, n); exp = z['opt_c2_%d' % i]
    bad = np.argwhere(got != exp)
    print('case', i, h, w, n, 'mismatch', len(bad), bad[:12].tolist(), flush=True)
    if len(bad):
        y, x, ch = bad[0]; print('  got', got[y, x], 'exp', exp[y, x], 'mask', m[y, x], 'img', c[y, x])
# stage-by-stage on the failing page
w, h, ch, seed, ns = 517, 389, 3, 15, 40.0
img, hocr = synth.synth_page(w, h, ch, seed=seed, noise_sigma=ns, line_div=24)
lib = _lib.load(); ctx = _lib.default_context()
gray = np.empty((h, w), np.uint8)
_lib.check(lib.mrchip_luma601(ctx.handle, _lib.ptr(img), _lib.ptr(gray), w, h)); print('luma ok', np.array_equal(gray, O.luma601(img)), flush=True)
s = mrc.estimate_noise(gray); print('sigma', s, O.estimate_noise(gray.astype(np.float32)), flush=True)
boxes = mrc.hocr_boxes(hocr, w, h); print('boxes', len(boxes), flush=True)
m = np.zeros((h, w), bool); mrc.create_hocr_mask(gray, m, hocr); e = np.zeros((h, w), bool); dec = []; O.create_hocr_mask(gray, e, boxes, None, dec)
print('hocr ok', np.array_equal(m, e), dec, flush=True)
wts, radius = mrc.gaussian_weights(s * 0.1); print('radius', radius, flush=True)
bl = np.empty_like(gray)
_lib.check(lib.mrchip_gaussian_u8(ctx.handle, _lib.ptr(gray), _lib.ptr(bl), w, h, s * 0.1, _lib.ptr(wts, _lib.f64p), radius)); print('gauss ok', flush=True)
t = mrc.threshold_image(bl, None); print('thr ok', flush=True)
m |= t
optimiser.fast_mask_denoise(m, w, h, 4, 2); print('denoise ok', flush=True)
fg = optimiser.optimise_rgb2(m, img, w, h, 3); print('fg ok', np.array_equal(fg, O.optimise_rgb2(m, img, w, h, 3)), flush=True)
bgm = ~m
bg = optimiser.optimise_rgb2(bgm, img, w, h, 10); print('bg ok', np.array_equal(bg, O.optimise_rgb2(bgm, img, w, h, 10)), flush=True)
g = mrc.create_mrc_hocr_components(img, hocr, bg_downsample=3, denoise_mask='fast')
mm = next(g); print('page mask ok', flush=True); f = next(g); print('page fg ok', flush=True); b = next(g); print('page bg ok', b.shape, flush=True)
