#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep (kernel entry points and whole-page decompositions over random
shapes / windows / k / n_size / downsample factors / hOCR boxes).  Not part of the pytest suite (minutes);
run on a GPU box:  python tests/fuzz_parity.py [seconds] [seed]

Environment:
  FUZZ_MODE=self      GPU-vs-GPU determinism mode (round 6): the checker of every case is the GPU path run AGAIN -- under
                      another draw of the kernel-form switches (fp64 / table Sauvola decision, counted / compiler stores,
                      float32 / float64 Gaussian, strips / whole rows, MFMA / VALU and fused / two-pass thumbnails), a
                      third of the time under the very same forms -- instead of the oracle.  A result that depends on the
                      run or on the form is a defect whatever the oracle says.  (Measured: about the same cases per
                      second as the oracle mode -- a case's time is host work, not the C oracle -- so a hunt runs many
                      processes side by side: tools/runs/fuzz_hunt.sh.)
  FUZZ_FAMILIES=0,4,13  only these families;  FUZZ_SCALE=0.5  shapes of the Sauvola / Gaussian / threshold-mask / sigma
                      families scaled down (more launches per second);  FUZZ_BIAS=n  half the cases from family n.
  FUZZ_INJECT=n       the checker's n-th result is corrupted on purpose: proves that a mismatch is caught and that the
                      arrays of the failing case land in gpurun_out/fuzz_fail_<seed>.npz.
  MRCHIP_CANARY=64    (library switch) guard bands around every device block; verified after every case here."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'archive-pdf-tools_amd'))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import mrc_oracle as O
from mrchip import _lib, mrc, sauvola, optimiser, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
rng = np.random.RandomState(seed)
lib, ctx = _lib.load(), _lib.default_context()
t0 = time.time()
counts = {}
SELF = os.environ.get('FUZZ_MODE') == 'self'
FAMILIES = [int(x) for x in os.environ['FUZZ_FAMILIES'].split(',')] if os.environ.get('FUZZ_FAMILIES') else list(range(17))
SCALE = float(os.environ.get('FUZZ_SCALE', '1'))
INJECT = int(os.environ.get('FUZZ_INJECT', '0'))
CANARY = bool(os.environ.get('MRCHIP_CANARY'))
FORM_SWITCHES = (('MRCHIP_GAUSS_FAST', '0'), ('MRCHIP_SAUVOLA_COUNTED_STORES', '0'), ('MRCHIP_SAUVOLA_FAST', '0'),
                 ('MRCHIP_OPT_STRIPS', '0'), ('MRCHIP_THUMB_NO_MFMA', '1'), ('MRCHIP_THUMB_NO_FUSE', '1'))


def sc(n, lo=1):
    """a drawn dimension under FUZZ_SCALE"""
    return max(lo, int(n * SCALE))


class GpuTwin:
    """FUZZ_MODE=self: the oracle's functions, answered by the GPU path itself under another draw of the form switches
    (all of them are read per launch).  `last_env` is what the second run saw; it goes into the dump of a failing case."""

    def __init__(self):
        self.calls = 0
        self.last_env = {}

    def _alt(self):
        keep = {k: os.environ.get(k) for k, _v in FORM_SWITCHES}
        same = rng.rand() < 0.34
        for k, v in FORM_SWITCHES:
            if same:
                continue                      # the very same forms again: pure run-to-run determinism
            if rng.rand() < 0.5: os.environ[k] = v
            else: os.environ.pop(k, None)
        self.last_env = {k: os.environ.get(k) for k, _v in FORM_SWITCHES}
        return keep

    def _restore(self, keep):
        for k, v in keep.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v

    def _run(self, fn):
        keep = self._alt()
        try:
            r = fn()
        finally:
            self._restore(keep)
        self.calls += 1
        if INJECT and self.calls == INJECT:
            a = r[0] if isinstance(r, list) else r
            if isinstance(a, np.ndarray) and a.size:
                a.reshape(-1)[0] ^= 1
                print('FUZZ_INJECT: corrupted one byte of checker result %d' % self.calls, flush=True)
        return r

    def binarise_sauvola(self, a, out, w, h, ww, wh, k, R):
        self._run(lambda: (sauvola.binarise_sauvola(a, out, w, h, ww, wh, k, R), out)[1])

    def optimise_gray2(self, mask, img, w, h, n): return self._run(lambda: optimiser.optimise_gray2(mask, img, w, h, n))
    def optimise_rgb2(self, mask, img, w, h, n): return self._run(lambda: optimiser.optimise_rgb2(mask, img, w, h, n))
    def fast_mask_denoise(self, m, w, h, mincnt, n): return self._run(lambda: optimiser.fast_mask_denoise(m, w, h, mincnt, n))
    def thumbnail_ex(self, img, rw, rh, flt, gap): return self._run(lambda: mrc.thumbnail(img, (rw, rh), resample=flt, reducing_gap=gap))

    def gaussian_filter(self, imgf, sig, weights=None):
        g = np.ascontiguousarray(imgf, dtype=np.uint8)
        h, w = g.shape
        wts, radius = (weights, len(weights) // 2) if weights is not None else mrc.gaussian_weights(sig)
        out = np.empty_like(g)
        return self._run(lambda: (_lib.check(lib.mrchip_gaussian_u8(ctx.handle, _lib.ptr(g), _lib.ptr(out), w, h, sig,
                                                                     _lib.ptr(wts, _lib.f64p), radius)), out)[1])

    def estimate_sigma(self, a): return self._run(lambda: mrc.mean_estimate_sigma(a))
    def estimate_noise(self, a): return self._run(lambda: mrc.estimate_noise(a))
    def threshold_image(self, src, dpi, k=0.34): return self._run(lambda: mrc.threshold_image(src, dpi, k))

    def luma601(self, rgb):
        h, w = rgb.shape[:2]
        out = np.empty((h, w), np.uint8)
        src = np.ascontiguousarray(rgb)
        return self._run(lambda: (_lib.check(lib.mrchip_luma601(ctx.handle, _lib.ptr(src), _lib.ptr(out), w, h)), out)[1])

    def create_mrc_hocr_components(self, img, hocr, **kw):
        res = self._run(lambda: list(mrc.create_mrc_hocr_components(img, hocr, **kw)))
        for a in res:
            yield a


if SELF:
    O = GpuTwin()


def tick(name):
    counts[name] = counts.get(name, 0) + 1


def rnd_img(h, w, c=1):
    kind = rng.randint(4)
    if kind == 0:
        a = rng.randint(0, 256, (h, w) if c == 1 else (h, w, c))
    elif kind == 1:
        a = np.full((h, w) if c == 1 else (h, w, c), rng.randint(0, 256))
        a = a + rng.randint(-3, 4, a.shape)
    else:
        img, _ = synth.synth_page(max(w, 64), max(h, 64), c, seed=int(rng.randint(1 << 30)), noise_sigma=float(rng.choice([0, 2, 6, 14])),
                                  line_div=int(rng.choice([8, 14, 30])))
        a = img[:h, :w]
    return np.clip(a, 0, 255).astype(np.uint8)


if os.environ.get('FUZZ_TRACE'):        # wait for the stream after every batch call and say so: a device fault then names its stage
    for _name in ('upload', 'set_boxes', 'set_count', 'mask_begin', 'sigmas', 'mask_finish', 'layers', 'download_mask',
                  'download_mask_packed', 'download_layer'):
        def _wrap(fn, name=_name):
            def f(self, *a, **k):
                r = fn(self, *a, **k)
                _lib.check(self.lib.mrchip_batch_sync(self._h), 'sync')
                print('ok', name, self.n, self.w, self.h, self.c, a[0] if a and isinstance(a[0], int) else '', flush=True)
                return r
            return f
        setattr(mrc.Batch, _name, _wrap(getattr(mrc.Batch, _name)))

os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
_last = open(os.path.join(ROOT, 'gpurun_out', 'fuzz_last_%d.txt' % seed), 'w')


def note(*a):
    """the case about to run, flushed: a GPU fault kills the process, this line says which case it was"""
    _last.seek(0); _last.truncate(); _last.write(repr(a) + '\n'); _last.flush(); os.fsync(_last.fileno())


def _dump_on_failure(tp, val, tb):
    """a failing case leaves its arrays (inputs, got, expected) and the kernel-form switches in gpurun_out/fuzz_fail_<seed>.npz"""
    try:
        keep = {k: v for k, v in globals().items() if isinstance(v, np.ndarray) and v.size < 8_000_000 and not k.startswith('_')}
        keep['env_switches'] = np.array([os.environ.get('MRCHIP_GAUSS_FAST', ''), os.environ.get('MRCHIP_SAUVOLA_COUNTED_STORES', ''),
                                         os.environ.get('MRCHIP_OPT_STRIPS', '')])
        keep['failure'] = np.array([repr(val)])
        if SELF: keep['twin_env'] = np.array([repr(O.last_env)])
        np.savez_compressed(os.path.join(ROOT, 'gpurun_out', 'fuzz_fail_%d.npz' % seed), **keep)
    finally:
        sys.__excepthook__(tp, val, tb)


sys.excepthook = _dump_on_failure

_ncase = 0
while time.time() - t0 < budget:
    if CANARY and _ncase:
        _bad = ctx.canary_check()
        assert _bad == 0, ('canary: %d guard byte(s) overwritten by the case before this one' % _bad, open(_last.name).read())
    _ncase += 1
    what = int(FAMILIES[rng.randint(len(FAMILIES))])
    if os.environ.get('FUZZ_BIAS') and rng.rand() < 0.5: what = int(os.environ['FUZZ_BIAS'])      # half the cases from one family
    # round 5: the two forms of the Gaussian (float32 + float64 fix-up / float64) and of the page kernel's stores
    # (counted asm stores / the compiler's) are read per launch: a case in five runs the other form
    for _var in ('MRCHIP_GAUSS_FAST', 'MRCHIP_SAUVOLA_COUNTED_STORES') + (('MRCHIP_SAUVOLA_FAST',) if SELF else ()):
        if rng.rand() < 0.2: os.environ[_var] = '0'
        else: os.environ.pop(_var, None)
    if what == 0:       # sauvola
        h, w = sc(rng.randint(1, 700)), sc(rng.randint(1, 1500))
        ww, wh = int(rng.randint(1, 140)), int(rng.randint(1, 140))
        if rng.rand() < 0.5: wh = ww
        k = float(rng.choice([0.34, 0.1, 0.5, -0.2, 0.0, 1.0])); R = float(rng.choice([128.0, 64.0, 200.0]))
        g = rnd_img(h, w)
        note('sauvola', h, w, ww, wh, k, R)
        out = np.empty(h * w, np.uint8)
        sauvola.binarise_sauvola(g.reshape(-1), out, w, h, ww, wh, k, R)
        exp = np.empty(h * w, np.uint8)
        O.binarise_sauvola(g.reshape(-1), exp, w, h, ww, wh, k, R)
        if not np.array_equal(out, exp):
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            np.savez_compressed(os.path.join(ROOT, 'gpurun_out', 'fuzz_fail.npz'), g=g, out=out, exp=exp, params=np.array([h, w, ww, wh, k, R]))
            bad = np.argwhere(out.reshape(h, w) != exp.reshape(h, w))
            print('MISMATCH sauvola', (h, w, ww, wh, k, R), len(bad), bad[:10].tolist())
            raise SystemExit(1)
        tick('sauvola')
    elif what == 1:     # optimise
        h, w = int(rng.randint(1, 400)), int(rng.randint(1, 1300))
        c = int(rng.choice([1, 3])); n = int(rng.choice([0, 1, 2, 3, 5, 8, 10, 11, 12, 20]))
        img = rnd_img(h, w, c)
        mask = (rng.rand(h, w) < rng.choice([0.02, 0.1, 0.5, 0.9, 1.0, 0.0])).astype(np.uint8)
        note('optimise', h, w, c, n, float(mask.mean()))
        f = optimiser.optimise_gray2 if c == 1 else optimiser.optimise_rgb2
        got = f(mask, img, w, h, n)
        exp = (O.optimise_gray2 if c == 1 else O.optimise_rgb2)(mask, img, w, h, n)
        assert np.array_equal(got, exp), ('optimise', h, w, c, n)
        tick('optimise')
    elif what == 2:     # denoise
        h, w = int(rng.randint(1, 500)), int(rng.randint(1, 1400))
        mask = (rng.rand(h, w) < rng.choice([0.01, 0.05, 0.2, 0.5])).astype(np.uint8)
        note('denoise', h, w)
        got = mask.copy(); optimiser.fast_mask_denoise(got, w, h, 4, 2)
        exp = O.fast_mask_denoise(mask.copy(), w, h, 4, 2)
        assert np.array_equal(got, exp), ('denoise', h, w)
        tick('denoise')
    elif what == 3:     # thumbnails
        h, w = int(rng.randint(8, 900)), int(rng.randint(8, 1400)); c = int(rng.choice([1, 3]))
        img = rnd_img(h, w, c)
        ds = float(rng.choice([1.3, 2, 2.5, 3, 3.7, 4, 5, 6.5, 9]))
        flt = str(rng.choice(['bicubic', 'lanczos'])); gap = None if rng.rand() < 0.5 else 2.0
        rw, rh = int(w / ds), int(h / ds)
        if rw < 1 or rh < 1: continue
        note('thumbnail', h, w, c, ds, flt, gap)
        got = mrc.thumbnail(img, (w / ds, h / ds), resample=flt, reducing_gap=gap)
        exp = O.thumbnail_ex(img, rw, rh, flt, gap)
        assert got.shape == exp.shape and np.array_equal(got, exp), ('thumbnail', h, w, c, ds, flt, gap)
        tick('thumbnail')
    elif what == 4:     # gaussian
        h, w = sc(rng.randint(2, 500), 2), sc(rng.randint(2, 1300), 2)
        g = rnd_img(h, w)
        sig = float(rng.choice([0.15, 0.3, 0.45, 0.63, 0.9, 1.3, 2.1, 3.3]))
        wts, radius = mrc.gaussian_weights(sig)
        note('gauss', h, w, sig)
        out = np.empty_like(g)
        _lib.check(lib.mrchip_gaussian_u8(ctx.handle, _lib.ptr(g), _lib.ptr(out), w, h, sig, _lib.ptr(wts, _lib.f64p), radius))
        exp = O.gaussian_filter(g.astype(np.float32), sig, weights=wts).astype(np.uint8)
        assert np.array_equal(out, exp), ('gauss', h, w, sig)
        tick('gauss')
    elif what == 6:     # noise estimate entry points (float32 path on uint8 values, float64 path on bool arrays)
        h, w = sc(rng.randint(1, 400)), sc(rng.randint(1, 900))
        if rng.rand() < 0.5:
            a = rnd_img(h, w)
            note('sigma_u8', h, w)
            got = mrc.mean_estimate_sigma(a.astype(np.float32))
            exp = O.estimate_sigma(a.astype(np.float32))
        else:
            a = rng.rand(h, w) < rng.choice([0.0, 0.03, 0.3, 0.5, 1.0])
            note('sigma_bool', h, w)
            got = mrc.mean_estimate_sigma(a)
            exp = O.estimate_sigma(a)
        assert (np.isnan(got) and np.isnan(exp)) or got == exp, ('sigma', h, w, got, exp)
        tick('sigma')
    elif what == 7:     # luma
        h, w = int(rng.randint(1, 500)), int(rng.randint(1, 1500))
        rgb = rnd_img(h, w, 3)
        note('luma', h, w)
        out = np.empty((h, w), np.uint8)
        _lib.check(lib.mrchip_luma601(ctx.handle, _lib.ptr(np.ascontiguousarray(rgb)), _lib.ptr(out), w, h))
        assert np.array_equal(out, O.luma601(rgb)), ('luma', h, w)
        tick('luma')
    elif what == 8:     # sauvola: large images and windows (8 / 16 columns per lane)
        h, w = int(rng.randint(200, 1800)), int(rng.randint(900, 3000))
        ww = int(rng.choice([31, 51, 91, 101, 121, 135, 200, 255, 301])); wh = ww if rng.rand() < 0.7 else int(rng.randint(3, 250))
        if ww * wh > 65792: wh = 65792 // ww
        k = float(rng.choice([0.34, 0.1, -0.2])); R = 128.0
        g = rnd_img(h, w)
        note('sauvola_big', h, w, ww, wh, k)
        out = np.empty(h * w, np.uint8); exp = np.empty(h * w, np.uint8)
        if os.environ.get('FUZZ_DIAG'):
            out[:] = 0xEE                      # (a byte that is still 0xEE afterwards was never written on the host)
        sauvola.binarise_sauvola(g.reshape(-1), out, w, h, ww, wh, k, R)
        O.binarise_sauvola(g.reshape(-1), exp, w, h, ww, wh, k, R)
        if os.environ.get('FUZZ_DIAG') and not np.array_equal(out, exp):
            # diagnosis mode (round 6): describe the mismatch, run the same call again, go on
            bad = np.argwhere(out.reshape(h, w) != exp.reshape(h, w))
            vals, cnts = np.unique(out[out != exp], return_counts=True)
            again = np.full(h * w, 0xEE, np.uint8)
            sauvola.binarise_sauvola(g.reshape(-1), again, w, h, ww, wh, k, R)
            counts['DIAG_mismatch'] = counts.get('DIAG_mismatch', 0) + 1
            print('DIAG sauvola_big', (h, w, ww, wh, k), 'bad px', len(bad), 'rows', int(bad[:, 0].min()), int(bad[:, 0].max()), 'cols',
                  int(bad[:, 1].min()), int(bad[:, 1].max()), 'distinct rows', len(np.unique(bad[:, 0])),
                  'values', dict(zip(vals.tolist()[:6], cnts.tolist()[:6])), 'n_values', len(vals),
                  'same call again wrong px', int((again != exp).sum()), flush=True)
        else:
            assert np.array_equal(out, exp), ('sauvola_big', h, w, ww, wh, k)
        tick('sauvola_big')
    elif what == 9:     # optimise on wide rows (more than 4096 columns: the unpacked kernel)
        h, w = int(rng.randint(1, 60)), int(rng.randint(3000, 9000))
        c = int(rng.choice([1, 3])); n = int(rng.choice([1, 3, 10, 14]))
        img = rnd_img(h, w, c)
        mask = (rng.rand(h, w) < rng.choice([0.05, 0.5, 0.95])).astype(np.uint8)
        note('optimise_wide', h, w, c, n)
        got = (optimiser.optimise_gray2 if c == 1 else optimiser.optimise_rgb2)(mask, img, w, h, n)
        exp = (O.optimise_gray2 if c == 1 else O.optimise_rgb2)(mask, img, w, h, n)
        assert np.array_equal(got, exp), ('optimise_wide', h, w, c, n)
        tick('optimise_wide')
    elif what == 10:    # create_hocr_mask on a caller's mask (assignment semantics), random boxes
        h, w = int(rng.randint(30, 500)), int(rng.randint(30, 900))
        gimg, _ = synth.synth_page(max(w, 64), max(h, 64), 1, seed=int(rng.randint(1 << 30)), noise_sigma=4.0, line_div=int(rng.choice([8, 16])))
        gimg = np.ascontiguousarray(gimg[:h, :w])
        lines = []
        for _ in range(int(rng.randint(0, 12))):
            l, t = int(rng.randint(0, w - 4)), int(rng.randint(0, h - 4))
            r, b = min(w, l + int(rng.randint(2, w))), min(h, t + int(rng.randint(2, 140)))
            lines.append({'bbox': [l, t, r, b], 'words': [{'text': 'w', 'confidence': int(rng.choice([90, 90, 10]))}]})
        hocr = [{'lines': lines}]
        m0 = (rng.rand(h, w) < 0.1)
        note('hocr_mask', h, w, len(lines))
        got = m0.copy(); mrc.create_hocr_mask(gimg, got, hocr, dpi=int(rng.choice([100, 200, 300])) if rng.rand() < 0.5 else None)
        tick('hocr_mask')   # compared through the page path below as well; here: must not fault and only touch box pixels
        touched = np.zeros((h, w), bool)
        for bx in mrc.hocr_boxes(hocr, w, h): touched[bx[1]:bx[3], bx[0]:bx[2]] = True
        assert np.array_equal(got[~touched], m0[~touched]), ('hocr_mask outside boxes', h, w)
    elif what == 11:    # batch of several same-sized pages == page by page
        h, w = int(rng.randint(40, 500)), int(rng.randint(40, 900)); c = int(rng.choice([1, 3])); npg = int(rng.randint(2, 6))
        pages = [synth.synth_page(w, h, c, seed=int(rng.randint(1 << 30)), noise_sigma=float(rng.choice([0, 4, 9, 20])),
                                  line_div=int(rng.choice([6, 12, 24]))) for _ in range(npg)]
        kw = dict(dpi=rng.choice([None, 150, 300]), bg_downsample=rng.choice([None, 2, 3]), fg_downsample=rng.choice([None, 2]))
        kw = {k: (None if v is None else int(v)) for k, v in kw.items()}
        note('batch', h, w, c, npg, kw)
        res = mrc.decompose_pages([p[0] for p in pages], [p[1] for p in pages], denoise_mask='fast', **kw)
        for (img, hocr), (mask, fg, bg) in zip(pages, res):
            e = O.create_mrc_hocr_components(img, hocr, denoise_mask='fast', **kw)
            for a in (mask, fg, bg):
                b = next(e)
                assert a.shape == b.shape and np.array_equal(a, b), ('batch', h, w, c, kw)
        tick('batch')
    elif what == 12:    # stream of pages of mixed sizes / modes, short last batches, both mask formats
        shapes = [(int(rng.randint(40, 400)), int(rng.randint(40, 700)), int(rng.choice([1, 3]))) for _ in range(int(rng.randint(1, 3)))]
        npg = int(rng.randint(1, 9))
        pages, spec = [], []
        for i in range(npg):
            h, w, c = shapes[int(rng.randint(len(shapes)))]
            spec.append(dict(w=w, h=h, channels=c, seed=int(rng.randint(1 << 30)), noise_sigma=float(rng.choice([0, 5, 11])),
                             line_div=int(rng.choice([6, 12]))))
            pages.append(synth.synth_page(**spec[-1]))
        kw = dict(dpi=rng.choice([None, 200]), bg_downsample=rng.choice([None, 3]), fg_downsample=rng.choice([None, 2]))
        kw = {k: (None if v is None else int(v)) for k, v in kw.items()}
        fmt = str(rng.choice(['bool', 'packed'])); bp = int(rng.randint(1, 5))
        note('stream', spec, kw, fmt, bp)
        n = 0
        for (img, hocr), (mask, fg, bg) in zip(pages, mrc.decompose_stream(iter(pages), batch_pages=bp, mask_format=fmt, copy=True, **kw)):
            e = O.create_mrc_hocr_components(img, hocr, denoise_mask='fast', **kw)
            em = next(e)
            if fmt == 'packed': em = np.packbits(em, axis=1)
            assert np.array_equal(mask, em), ('stream mask', n, kw, fmt)
            for a in (fg, bg):
                b = next(e)
                assert a.shape == b.shape and np.array_equal(a, b), ('stream layer', n, kw)
            n += 1
        assert n == npg
        tick('stream')
    elif what == 13:    # create_threshold_mask: estimate + blur + Sauvola + in-place OR
        h, w = sc(rng.randint(8, 500), 8), sc(rng.randint(8, 900), 8)
        gimg, _ = synth.synth_page(max(w, 64), max(h, 64), 1, seed=int(rng.randint(1 << 30)), noise_sigma=float(rng.choice([0, 3, 8, 20])),
                                   line_div=int(rng.choice([8, 16])))
        gimg = np.ascontiguousarray(gimg[:h, :w])
        dpi = None if rng.rand() < 0.5 else int(rng.choice([100, 200, 400]))
        m0 = rng.rand(h, w) < 0.05
        note('threshold_mask', h, w, dpi)
        got = m0.copy()
        mrc.create_threshold_mask(got, gimg.astype(np.float32), dpi=dpi)
        sig = O.estimate_noise(gimg.astype(np.float32))
        src = gimg
        if sig > 1.0:
            wts, _r = mrc.gaussian_weights(sig * 0.1)
            src = O.gaussian_filter(gimg.astype(np.float32), sig * 0.1, weights=wts).astype(np.uint8)
        exp = m0 | O.threshold_image(src, dpi)
        if not np.array_equal(got, exp):
            # which stage?  the same inputs through each stage again (a second full run tells a race from a wrong result)
            gs = mrc.estimate_noise(gimg)
            print('threshold_mask mismatch: %d px; sigma gpu %r oracle %r' % (int((got != exp).sum()), gs, sig), flush=True)
            if sig > 1.0:
                gb = np.empty_like(gimg)
                _lib.check(lib.mrchip_gaussian_u8(ctx.handle, _lib.ptr(gimg), _lib.ptr(gb), w, h, sig * 0.1, _lib.ptr(wts, _lib.f64p), _r))
                print('  gaussian again: %d px differ from the oracle' % int((gb != src).sum()), np.argwhere(gb != src)[:6].tolist(), flush=True)
            tb = mrc.threshold_image(src, dpi)
            print('  threshold of the oracle-blurred image again: %d px differ' % int((tb != O.threshold_image(src, dpi)).sum()), flush=True)
            again = m0.copy()
            mrc.create_threshold_mask(again, gimg.astype(np.float32), dpi=dpi)
            print('  whole call again: %d px differ; first differing px of the failing call' % int((again != exp).sum()),
                  np.argwhere(got != exp)[:8].tolist(), flush=True)
        assert np.array_equal(got, exp), ('threshold_mask', h, w, dpi)
        tick('threshold_mask')
    elif what == 14:    # optimise, band walkers (whole rows forced): unselected pixels in runs of rows with gaps around n_size
        h, w = int(rng.randint(30, 400)), int(rng.randint(8, 1300))
        c = int(rng.choice([1, 3])); n = int(rng.choice([1, 2, 3, 5, 7, 10, 11]))
        img = rnd_img(h, w, c)
        mask = np.ones((h, w), np.uint8)
        y = int(rng.randint(0, 3 * n + 2)) if rng.rand() < 0.8 else 0
        while y < h:
            run = int(rng.randint(1, 12))
            for yy in range(y, min(h, y + run)):
                if rng.rand() < 0.7:
                    mask[yy] = (rng.rand(w) >= rng.choice([0.01, 0.2, 0.8])).astype(np.uint8)
            y += run + int(rng.choice([n - 1, n, n + 1, 2 * n + 1, 0, 40]))
        if rng.rand() < 0.2: mask[h - 1, int(rng.randint(w))] = 0
        inv = bool(rng.rand() < 0.5)
        note('optimise_bands', h, w, c, n, inv)
        os.environ['MRCHIP_OPT_STRIPS'] = '0'
        try:
            got = np.empty_like(img)
            m_arg = np.ascontiguousarray(1 - mask) if inv else mask
            _lib.check(lib.mrchip_optimise(ctx.handle, _lib.ptr(m_arg), _lib.ptr(img), _lib.ptr(got), w, h, c, n, 1 if inv else 0))
        finally:
            del os.environ['MRCHIP_OPT_STRIPS']
        exp = (O.optimise_gray2 if c == 1 else O.optimise_rgb2)(mask, img, w, h, n)
        assert np.array_equal(got, exp), ('optimise_bands', h, w, c, n, inv)
        tick('optimise_bands')
    elif what == 15:    # adversarial pages: black scanner border, saturated / constant regions, an inverted block, two-level areas
        h, w = int(rng.randint(60, 700)), int(rng.randint(60, 1100)); c = int(rng.choice([1, 3]))
        img, hocr = synth.synth_page(w, h, c, seed=int(rng.randint(1 << 30)), noise_sigma=float(rng.choice([0, 3, 6, 12])),
                                     line_div=int(rng.choice([8, 14, 30])))
        img = img.copy()
        ops = rng.rand(6)
        if ops[0] < 0.5:
            bw = int(rng.randint(1, 41)); img[:bw] = 3; img[-bw:] = 0; img[:, :bw] = 5; img[:, -bw:] = 2
        if ops[1] < 0.5:
            y0, x0 = int(rng.randint(h // 2)), int(rng.randint(w // 2)); img[y0:y0 + h // 3, x0:x0 + w // 3] = 255 - img[y0:y0 + h // 3, x0:x0 + w // 3]
        if ops[2] < 0.4:
            y0 = int(rng.randint(h - 8)); img[y0:y0 + int(rng.randint(1, 60))] = int(rng.choice([0, 255, 128]))
        if ops[3] < 0.4:
            yy, xx = np.mgrid[0:h, 0:w]; chk = ((yy // 5 + xx // 3) % 2 == 0); sel = (yy > h // 2) & (xx > w // 2) & chk
            img[sel] = 0; img[(yy > h // 2) & (xx > w // 2) & ~chk] = 255
        if ops[4] < 0.3:
            img[:, int(rng.randint(w))] = 20                   # a rule down the page: ink in every row
        kw = dict(dpi=rng.choice([None, 200, 400]), bg_downsample=rng.choice([None, 3]), fg_downsample=rng.choice([None, 2]))
        kw = {k: (None if v is None else int(v)) for k, v in kw.items()}
        whole = bool(rng.rand() < 0.5)
        note('adversarial_page', h, w, c, kw, whole, ops.tolist())
        if whole: os.environ['MRCHIP_OPT_STRIPS'] = '0'
        try:
            g = mrc.create_mrc_hocr_components(img, hocr, denoise_mask='fast', **kw)
            e = O.create_mrc_hocr_components(img, hocr, denoise_mask='fast', **kw)
            for i in range(3):
                a, b = next(g), next(e)
                assert a.shape == b.shape and np.array_equal(a, b), ('adversarial_page', h, w, c, kw, i)
        finally:
            os.environ.pop('MRCHIP_OPT_STRIPS', None)
        tick('adversarial_page')
    elif what == 16:    # batches through the band walkers + the thumbnail that reads gap rows from the image
        h, w = int(rng.randint(60, 420)), int(rng.randint(120, 900)); c = int(rng.choice([1, 3])); npg = int(rng.randint(1, 5))
        pages = [synth.synth_page(w, h, c, seed=int(rng.randint(1 << 30)), noise_sigma=float(rng.choice([0, 4, 9])),
                                  line_div=int(rng.choice([6, 12, 24]))) for _ in range(npg)]
        kw = dict(dpi=rng.choice([None, 150]), bg_downsample=rng.choice([None, 2, 3, 4]), fg_downsample=rng.choice([None, 2, 3]))
        kw = {k: (None if v is None else int(v)) for k, v in kw.items()}
        note('batch_bands', h, w, c, npg, kw)
        os.environ['MRCHIP_OPT_STRIPS'] = '0'
        try:
            res = mrc.decompose_pages([p[0] for p in pages], [p[1] for p in pages], denoise_mask='fast', **kw)
        finally:
            del os.environ['MRCHIP_OPT_STRIPS']
        for (img, hocr), (mask, fg, bg) in zip(pages, res):
            e = O.create_mrc_hocr_components(img, hocr, denoise_mask='fast', **kw)
            for a in (mask, fg, bg):
                b = next(e)
                assert a.shape == b.shape and np.array_equal(a, b), ('batch_bands', h, w, c, kw)
        tick('batch_bands')
    else:               # whole pages (what == 5)
        h, w = int(rng.randint(20, 900)), int(rng.randint(20, 1300)); c = int(rng.choice([1, 3]))
        img, hocr = synth.synth_page(w, h, c, seed=int(rng.randint(1 << 30)), noise_sigma=float(rng.choice([0, 3, 6, 12, 25])),
                                     line_div=int(rng.choice([6, 10, 20, 40])))
        kw = dict(dpi=rng.choice([None, 100, 200, 364, 400]), bg_downsample=rng.choice([None, 2, 3, 4]),
                  fg_downsample=rng.choice([None, 2, 3]), denoise_mask=str(rng.choice(['fast', 'none'])))
        kw = {k: (None if v is None else (v if isinstance(v, str) else int(v))) for k, v in kw.items()}
        note('page', h, w, c, kw)
        g = mrc.create_mrc_hocr_components(img, hocr, **kw)
        e = O.create_mrc_hocr_components(img, hocr, **kw)
        for i in range(3):
            a, b = next(g), next(e)
            assert a.shape == b.shape and np.array_equal(a, b), ('page', h, w, c, kw, i)
        tick('page')
if CANARY:
    assert ctx.canary_selftest() == 2, 'the guard bands did not see a deliberate stray write'
    assert ctx.canary_check() == 0
print('fuzz ok%s%s: %.0f s, seed %d, %d cases %s' % (' (GPU-vs-GPU, forms redrawn)' if SELF else '', ' (guard bands on)' if CANARY else '',
                                                  time.time() - t0, seed, sum(counts.values()), counts))
