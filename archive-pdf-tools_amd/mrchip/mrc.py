"""Drop-in for the hot path of internetarchivepdf/mrc.py.

Same names, arguments, defaults, yielded arrays, timing keys and error behaviour
as the reference (file:line cited per function); the pixel work runs in
libmrchip.so on the GPU.  `create_mrc_hocr_components` keeps the page resident
on the device between its three yields.
"""
import ctypes as C
import sys
from time import time

import numpy as np

from . import _lib
from .sauvola import binarise_sauvola  # noqa: F401  (mrc.py:37 imports the name)
from .optimiser import (optimise_gray, optimise_rgb, optimise_gray2, optimise_rgb2,  # noqa: F401
                        fast_mask_denoise)

# internetarchivepdf/const.py:38,41-43
RECODE_RUNTIME_WARNING_TOO_SMALL_TO_DOWNSAMPLE = 'too-small-to-downsample'
DENOISE_NONE = 'none'
DENOISE_FAST = 'fast'
DENOISE_BREGMAN = 'bregman'
# create_mrc_hocr_components: honour edits a caller makes to the yielded mask between next() calls, as the reference's
# shared array does; False saves a copy and two compares of the mask per page
SHARED_MASK = True


def _window_size(dpi):
    """mrc.py:68-75"""
    window_size = 51
    if dpi is not None:
        window_size = int(dpi / 4)
        if window_size % 2 == 0:
            window_size += 1
    return window_size


def threshold_image(img, dpi, k=0.34, ctx=None):
    """mrc.threshold_image (mrc.py:58-87): Sauvola binarisation, True = dark/foreground."""
    window_size = _window_size(dpi)
    h, w = img.shape
    src = _lib.as_u8(img, 'img')
    out_img = np.empty((h, w), dtype=np.uint8)
    ctx = ctx or _lib.default_context()
    # np.invert (mrc.py:85) is fused into the kernel's store
    _lib.check(_lib.load().mrchip_sauvola_u8(ctx.handle, _lib.ptr(src), _lib.ptr(out_img), w, h, window_size,
                                             window_size, float(k), 128.0, 1), 'mrchip_sauvola_u8')
    return out_img.view(np.bool_)


def _u8_valued(imgf, what):
    """The float32 images of the production path are `np.array(grayimg, dtype=np.float32)` (mrc.py:372): whole numbers
    0..255, for which the device works on the uint8 plane (the fast kernels).  Returns (uint8 plane, None) for those and
    (None, float32 array) for a float32 image with other values (round 6: the general form of mrc.py:273-329, slower
    kernels, same results as the reference).  Other dtypes with non-uint8 values are refused: the reference would run
    PyWavelets / scipy in float64 for them, which this path does not restate."""
    a = np.asarray(imgf)
    if a.ndim != 2:
        raise ValueError('%s: a 2-D image is expected, got shape %r' % (what, a.shape))
    if a.dtype == np.uint8:
        return np.ascontiguousarray(a), None
    with np.errstate(all='ignore'):
        src = np.ascontiguousarray(a, dtype=np.uint8)
    if np.array_equal(src, a):
        return src, None
    if a.dtype == np.float32:
        return None, np.ascontiguousarray(a)
    raise _lib.MrchipError('%s: a uint8-valued image or a float32 image is expected, got %s with other values' % (what, a.dtype))


def mean_estimate_sigma(arr, ctx=None):
    """mrc.mean_estimate_sigma (mrc.py:52-55) for float32 images and bool arrays."""
    a = np.asarray(arr)
    ctx = ctx or _lib.default_context()
    sigma = C.c_double()
    if a.dtype == np.bool_:
        src = np.ascontiguousarray(a).view(np.uint8)
        kind = 1
    else:
        src, f32 = _u8_valued(a, 'mean_estimate_sigma')
        if f32 is not None:
            h, w = f32.shape
            _lib.check(_lib.load().mrchip_estimate_sigma_f32(ctx.handle, _lib.ptr(f32, _lib.f32p), w, w, h, C.byref(sigma)),
                       'mrchip_estimate_sigma_f32')
            return sigma.value
        kind = 0
    h, w = src.shape
    _lib.check(_lib.load().mrchip_estimate_sigma(ctx.handle, _lib.ptr(src), w, w, h, kind, C.byref(sigma)),
               'mrchip_estimate_sigma')
    return sigma.value


def estimate_noise(imgf, ctx=None):
    """mrc.estimate_noise (mrc.py:273-296) on float32(gray) -- or on any float32 image (the general kernels)."""
    src, f32 = _u8_valued(imgf, 'estimate_noise')
    ctx = ctx or _lib.default_context()
    sigma = C.c_double()
    if f32 is not None:
        h, w = f32.shape
        _lib.check(_lib.load().mrchip_estimate_noise_f32(ctx.handle, _lib.ptr(f32, _lib.f32p), w, h, C.byref(sigma)),
                   'mrchip_estimate_noise_f32')
        return sigma.value
    h, w = src.shape
    _lib.check(_lib.load().mrchip_estimate_noise_u8(ctx.handle, _lib.ptr(src), w, h, C.byref(sigma)),
               'mrchip_estimate_noise_u8')
    return sigma.value


def gaussian_weights(sigma):
    """The table scipy.ndimage.gaussian_filter builds on the host (filters.py
    _gaussian_kernel1d, order 0, truncate 4.0) -- numpy expression for expression,
    so that it equals what the reference's scipy computes in the same environment."""
    sd = float(sigma)
    lw = int(4.0 * sd + 0.5)
    sigma2 = sd * sd
    x = np.arange(-lw, lw + 1)
    phi_x = np.exp(-0.5 / sigma2 * x ** 2)
    phi_x = phi_x / phi_x.sum()
    return np.ascontiguousarray(phi_x[::-1], dtype=np.float64), lw


def _line_box(line, scale):
    """One hOCR line -> (left, top, right, bottom) in page pixels, or None when create_hocr_mask skips the
    line without a message (mrc.py:198-213): no text, mean word confidence under 20, or an empty box."""
    words = line['words']
    if not ' '.join(w['text'] for w in words).strip():
        return None
    conf = [w['confidence'] for w in words]
    if (sum(conf) / len(conf) if conf else 0) < 20:
        return None
    box = tuple(int(v / scale) if scale is not None else int(v) for v in line['bbox'])
    if box[0] == box[2] or box[1] == box[3]:
        return None
    return box


def hocr_boxes(hocr_word_data, image_width, image_height, downsample=None):
    """The line filter of mrc.create_hocr_mask (mrc.py:194-221) as host logic: int32[nb,4] boxes in list
    order.  Inverted boxes and boxes that leave the page are reported on stderr with the reference's
    messages (mrc.py:216, 220) and dropped."""
    kept = []
    for paragraph in hocr_word_data:
        for line in paragraph['lines']:
            box = _line_box(line, downsample)
            if box is None:
                continue
            l, t, r, b = box
            if l >= r or t >= b:
                print('Invalid bounding box: (%d, %d, %d, %d)' % box, file=sys.stderr)
            elif l < 0 or t < 0 or r > image_width or b > image_height:
                print('Invalid bounding box outside image: (%d, %d, %d, %d)' % box, file=sys.stderr)
            else:
                kept.append(box)
    return np.ascontiguousarray(np.asarray(kept, dtype=np.int32).reshape(-1, 4))


def create_hocr_mask(img, mask_arr, hocr_word_data, downsample=None, dpi=None, timing_data=None, ctx=None):
    """mrc.create_hocr_mask (mrc.py:188-270): img is a PIL 'L' image or uint8[h,w]; mask_arr modified in place."""
    np_img = _lib.as_u8(np.array(img), 'img')
    image_height, image_width = np_img.shape
    t = time()
    boxes = hocr_boxes(hocr_word_data, image_width, image_height, downsample)
    m = np.asarray(mask_arr)
    if m.shape != np_img.shape:
        raise ValueError('create_hocr_mask: mask_arr shape %r does not match the image %r' % (m.shape, np_img.shape))
    if len(boxes):
        tmp = _lib.as_u8(m, 'mask_arr')
        ctx = ctx or _lib.default_context()
        _lib.check(_lib.load().mrchip_hocr_mask(ctx.handle, _lib.ptr(np_img), _lib.ptr(tmp), image_width, image_height,
                                                _lib.ptr(boxes, _lib.i32p), len(boxes), _window_size(dpi), None),
                   'mrchip_hocr_mask')
        if not np.shares_memory(tmp, m):
            m[...] = tmp.view(m.dtype) if m.dtype == np.bool_ else tmp
    if timing_data is not None:
        timing_data.append(('hocr_mask_gen', time() - t))


def create_threshold_mask(mask_arr, imgf, dpi=None, denoise_mask=None, timing_data=None, ctx=None):
    """mrc.create_threshold_mask (mrc.py:300-329): noise estimate, Gaussian blur when sigma_est > 1, Sauvola
    k=0.34 of the (truncated) result, OR-ed into mask_arr in place.  `imgf` is float32(gray) as in
    mrc.py:372; `denoise_mask` is accepted and unused like in the reference.  Timing keys est_1 / blur_1 /
    threshold as the reference appends them."""
    gray, f32 = _u8_valued(imgf, 'create_threshold_mask')
    m = np.asarray(mask_arr)
    if m.shape != (gray if f32 is None else f32).shape:
        raise ValueError('create_threshold_mask: mask_arr shape %r does not match the image %r' % (m.shape, np.asarray(imgf).shape))
    if f32 is not None and not (np.isfinite(f32).all() and f32.min() >= 0 and f32.max() < 256):
        # imgf.astype(np.uint8) (mrc.py:325) of such values is what the C compiler of the reference's numpy makes of an
        # out-of-range float -> uint8 cast: platform-defined, so there is nothing to be identical to
        raise _lib.MrchipError('create_threshold_mask: float32 image values must be finite and in [0, 256)')
    ctx = ctx or _lib.default_context()
    t = time()
    sigma_est = estimate_noise(gray if f32 is None else f32, ctx=ctx)
    if timing_data is not None:
        timing_data.append(('est_1', time() - t))
    if f32 is not None:
        # the general float32 form: scipy's float32 gaussian_filter (when sigma_est > 1) and the uint8 truncation, one call
        t = time()
        h, w = f32.shape
        gray = np.empty((h, w), np.uint8)
        if sigma_est > 1.0:                                               # mrc.py:309-313
            wts, radius = gaussian_weights(sigma_est * 0.1)
            _lib.check(_lib.load().mrchip_gaussian_f32(ctx.handle, _lib.ptr(f32, _lib.f32p), _lib.ptr(gray), w, h,
                                                       float(sigma_est * 0.1), _lib.ptr(wts, _lib.f64p), radius), 'mrchip_gaussian_f32')
            if timing_data is not None:
                timing_data.append(('blur_1', time() - t))
        else:
            _lib.check(_lib.load().mrchip_gaussian_f32(ctx.handle, _lib.ptr(f32, _lib.f32p), _lib.ptr(gray), w, h, 0.0, None, 0),
                       'mrchip_gaussian_f32')
    elif sigma_est > 1.0:                                                 # mrc.py:309-313
        t = time()
        wts, radius = gaussian_weights(sigma_est * 0.1)
        h, w = gray.shape
        blurred = np.empty_like(gray)
        _lib.check(_lib.load().mrchip_gaussian_u8(ctx.handle, _lib.ptr(gray), _lib.ptr(blurred), w, h,
                                                  float(sigma_est * 0.1), _lib.ptr(wts, _lib.f64p), radius),
                   'mrchip_gaussian_u8')
        gray = blurred                                                    # imgf.astype(np.uint8), mrc.py:325
        if timing_data is not None:
            timing_data.append(('blur_1', time() - t))
    t = time()
    thres_arr = threshold_image(gray, dpi, ctx=ctx)
    if timing_data is not None:
        timing_data.append(('threshold', time() - t))
    if m.dtype == np.bool_:
        m |= thres_arr                                                    # mrc.py:329
    else:
        m |= thres_arr.view(np.uint8)


def _require_skimage_for_bregman():
    """denoise_mask='bregman' is a host pass through scikit-image, the dependency the reference itself uses for it
    (mrc.py:34, 101).  Asked for where it is not installed, the call fails HERE -- when the option is given, before a
    page is uploaded or a mask computed -- not minutes into a book."""
    import importlib.util
    if importlib.util.find_spec('skimage') is None:
        raise ImportError("denoise_mask='bregman' needs scikit-image (skimage.restoration.denoise_tv_bregman), the same "
                          "dependency the reference uses for it (mrc.py:34, 101); it is not installed in this environment. "
                          "Use denoise_mask='fast' or 'none', or install scikit-image.")


def denoise_bregman(binary_img):
    """mrc.denoise_bregman (mrc.py:90-108): host passthrough to scikit-image's iterative TV solver, which is
    third-party code the reference calls as is (SURVEY.md 8f rank 4 keeps it on the CPU).  Raises
    ImportError where scikit-image is not installed -- there is no stand-in."""
    _require_skimage_for_bregman()
    from skimage.restoration import denoise_tv_bregman
    thresf = np.array(binary_img, dtype=np.float32)
    return np.array(denoise_tv_bregman(thresf, weight=1.) > 0.4, dtype=bool)


def _same_bytes(a, b):
    """np.array_equal for two 1-byte-per-element arrays of one shape, compared 8 bytes at a time where the layout allows
    (half the time of the element-wise form on a 12 Mpx mask)"""
    if a.shape != b.shape:
        return False
    if a.flags.c_contiguous and b.flags.c_contiguous and a.itemsize == 1 and b.itemsize == 1:
        n8 = a.size & ~7
        fa, fb = a.reshape(-1).view(np.uint8), b.reshape(-1).view(np.uint8)
        return bool(np.array_equal(fa[:n8].view(np.uint64), fb[:n8].view(np.uint64)) and np.array_equal(fa[n8:], fb[n8:]))
    return bool(np.array_equal(a, b))


def _checked_page(arr, w, h, c, what):
    """C-contiguous uint8 array of exactly the handle's geometry: the library copies w*c*h bytes from the
    pointer, so a smaller or differently shaped array must be refused here (ValueError like a Cython buffer
    mismatch in the reference)."""
    a = _lib.as_u8(arr, what)
    want = (h, w) if c == 1 else (h, w, c)
    if a.shape != want:
        raise ValueError('%s: array of shape %r expected, got %r' % (what, want, a.shape))
    return a


class _Page:
    """mrchip_page handle (device-resident page)."""

    def __init__(self, ctx, w, h, c):
        self.lib = _lib.load()
        self.ctx = ctx
        self.w, self.h, self.c = w, h, c
        self._h = self.lib.mrchip_page_create(ctx.handle, w, h, c)
        if not self._h:
            raise _lib.MrchipError('mrchip_page_create: %s' % _lib.last_error())

    def close(self):
        if self._h:
            self.lib.mrchip_page_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass

    def upload(self, arr):
        arr = _checked_page(arr, self.w, self.h, self.c, 'page')
        _lib.check(self.lib.mrchip_page_upload(self._h, _lib.ptr(arr)), 'mrchip_page_upload')

    def upload_gray(self, gray):
        gray = _checked_page(gray, self.w, self.h, 1, 'gray plane')
        _lib.check(self.lib.mrchip_page_upload_gray(self._h, _lib.ptr(gray)), 'mrchip_page_upload_gray')

    def upload_mask(self, mask):
        mask = _checked_page(mask, self.w, self.h, 1, 'mask')
        _lib.check(self.lib.mrchip_page_upload_mask(self._h, _lib.ptr(mask)), 'mrchip_page_upload_mask')

    def mask_begin(self, boxes, window):
        _lib.check(self.lib.mrchip_page_mask_begin(self._h, _lib.ptr(boxes, _lib.i32p) if len(boxes) else None,
                                                   len(boxes), window), 'mrchip_page_mask_begin')

    def sigma(self):
        s = C.c_double()
        _lib.check(self.lib.mrchip_page_sigma(self._h, C.byref(s)), 'mrchip_page_sigma')
        return s.value

    def mask_finish(self, sigma_est, denoise_fast, weights='numpy'):
        wts, radius = None, 0
        if sigma_est > 1.0 and weights == 'numpy':
            wts, radius = gaussian_weights(sigma_est * 0.1)
        _lib.check(self.lib.mrchip_page_mask_finish(self._h, _lib.ptr(wts, _lib.f64p) if wts is not None else None,
                                                    radius, 1 if denoise_fast else 0), 'mrchip_page_mask_finish')

    def download_mask(self):
        m = np.empty((self.h, self.w), dtype=np.uint8)
        _lib.check(self.lib.mrchip_page_download_mask(self._h, _lib.ptr(m)), 'mrchip_page_download_mask')
        return m.view(np.bool_)

    def layer(self, is_bg, downsample):
        ow, oh, small = C.c_int(), C.c_int(), C.c_int()
        _lib.check(self.lib.mrchip_page_layer(self._h, 1 if is_bg else 0, float(downsample or 0.0), C.byref(ow),
                                              C.byref(oh), C.byref(small)), 'mrchip_page_layer')
        return ow.value, oh.value, bool(small.value)

    def layers(self, fg_downsample, bg_downsample):
        """fg and bg in one launch: ((fg_w, fg_h), (bg_w, bg_h), too_small bits)"""
        fw, fh, bw, bh, ts = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _lib.check(self.lib.mrchip_page_layers(self._h, float(fg_downsample or 0.0), float(bg_downsample or 0.0), C.byref(fw),
                                               C.byref(fh), C.byref(bw), C.byref(bh), C.byref(ts)), 'mrchip_page_layers')
        return (fw.value, fh.value), (bw.value, bh.value), ts.value

    def download_layer(self, is_bg, ow, oh):
        shape = (oh, ow) if self.c == 1 else (oh, ow, 3)
        out = np.empty(shape, dtype=np.uint8)
        _lib.check(self.lib.mrchip_page_download_layer(self._h, 1 if is_bg else 0, _lib.ptr(out)),
                   'mrchip_page_download_layer')
        return out

    def sync(self):
        _lib.check(self.lib.mrchip_page_sync(self._h), 'mrchip_page_sync')

    def box_decisions(self, nb):
        d = np.zeros(max(nb, 1), dtype=np.int32)
        _lib.check(self.lib.mrchip_page_box_decisions(self._h, _lib.ptr(d, _lib.i32p), nb))
        return d[:nb].tolist()


class Batch:
    """mrchip_batch handle: N same-sized pages resident on the device, every stage of
    create_mrc_hocr_components one launch over the whole batch."""

    def __init__(self, ctx, npages, w, h, c):
        self.lib = _lib.load()
        self.ctx = ctx
        self.n, self.w, self.h, self.c = npages, w, h, c
        self._h = self.lib.mrchip_batch_create(ctx.handle, npages, w, h, c)
        if not self._h:
            raise _lib.MrchipError('mrchip_batch_create: %s' % _lib.last_error())
        self._wtab = np.zeros((npages, _lib.MAX_TAPS), dtype=np.float64)
        self._radius = np.zeros(npages, dtype=np.int32)
        self._sources = []          # host arrays of uploads that may still be in flight

    def close(self):
        if self._h:
            self.lib.mrchip_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass

    def upload(self, page, arr):
        """Enqueue the copy of one page.  The source array is referenced until the batch's stream is next waited
        for: the HIP runtime may copy from page-locked AND from ordinary host memory after this call returns."""
        arr = _checked_page(arr, self.w, self.h, self.c, 'page')
        self._sources.append(arr)
        _lib.check(self.lib.mrchip_batch_upload(self._h, page, _lib.ptr(arr)), 'mrchip_batch_upload')

    def upload_gray(self, page, gray):
        """Gray plane of an RGB page from the caller (PIL's convert('L') of a mode other than L / RGB)."""
        gray = _checked_page(gray, self.w, self.h, 1, 'gray plane')
        _lib.check(self.lib.mrchip_batch_upload_gray(self._h, page, _lib.ptr(gray)), 'mrchip_batch_upload_gray')

    def upload_mask(self, page, mask):
        mask = _checked_page(mask, self.w, self.h, 1, 'mask')
        _lib.check(self.lib.mrchip_batch_upload_mask(self._h, page, _lib.ptr(mask)), 'mrchip_batch_upload_mask')

    def set_count(self, count):
        """Pages in use (1..n): the stages then skip pages >= count (a short last batch of a stream)."""
        _lib.check(self.lib.mrchip_batch_set_count(self._h, count), 'mrchip_batch_set_count')

    def set_boxes(self, page, boxes):
        boxes = np.ascontiguousarray(boxes, dtype=np.int32).reshape(-1, 4)
        _lib.check(self.lib.mrchip_batch_set_boxes(self._h, page, _lib.ptr(boxes, _lib.i32p) if len(boxes) else None,
                                                   len(boxes)), 'mrchip_batch_set_boxes')

    def mask_begin(self, window):
        _lib.check(self.lib.mrchip_batch_mask_begin(self._h, window), 'mrchip_batch_mask_begin')

    def threshold(self, dpi=None, k=0.34):
        """mrc.threshold_image (mrc.py:58-87) of every page in one launch; the result is the batch's mask."""
        _lib.check(self.lib.mrchip_batch_threshold(self._h, _window_size(dpi), float(k)), 'mrchip_batch_threshold')

    def sigmas(self):
        s = np.zeros(self.n, dtype=np.float64)
        _lib.check(self.lib.mrchip_batch_sigmas(self._h, _lib.ptr(s, _lib.f64p)), 'mrchip_batch_sigmas')
        self._sources = []          # the call waited for the stream: every upload has landed
        return s

    def mask_finish(self, sigmas, denoise_fast=True):
        """Builds the per-page Gaussian tables on the host exactly like scipy and enqueues phase B."""
        for i, s in enumerate(sigmas):
            if s > 1.0:
                wts, r = gaussian_weights(s * 0.1)
                self._wtab[i, :len(wts)] = wts
                self._radius[i] = r
            else:
                self._radius[i] = 0
        _lib.check(self.lib.mrchip_batch_mask_finish(self._h, _lib.ptr(self._wtab, _lib.f64p),
                                                     _lib.ptr(self._radius, _lib.intp), 1 if denoise_fast else 0),
                   'mrchip_batch_mask_finish')

    def layers(self, fg_downsample=None, bg_downsample=None, which=3):
        fw, fh, bw, bh, ts = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _lib.check(self.lib.mrchip_batch_layers(self._h, which, float(fg_downsample or 0.0), float(bg_downsample or 0.0),
                                                C.byref(fw), C.byref(fh), C.byref(bw), C.byref(bh), C.byref(ts)),
                   'mrchip_batch_layers')
        return (fw.value, fh.value), (bw.value, bh.value), ts.value

    @staticmethod
    def _dest(out, shape, what):
        if out is None:
            return np.empty(shape, dtype=np.uint8)
        if out.shape != shape or out.dtype not in (np.uint8, np.bool_) or not out.flags.c_contiguous:
            raise ValueError('%s: out must be a C-contiguous 1-byte array of shape %r' % (what, shape))
        return out

    def download_mask(self, page, out=None, wait=True):
        """bool[h,w] mask of one page; out / wait as in download_layer."""
        m = self._dest(out, (self.h, self.w), 'download_mask')
        fn = self.lib.mrchip_batch_download_mask if wait else self.lib.mrchip_batch_download_mask_async
        _lib.check(fn(self._h, page, _lib.ptr(m.view(np.uint8))), 'mrchip_batch_download_mask')
        return m.view(np.bool_)

    def download_mask_packed(self, page, out=None, wait=True):
        """The finished mask at 1 bit per pixel (MSB first, rows of ceil(w/8) bytes): what
        mrc.encode_mrc_mask (mrc.py:474-520) feeds to jbig2 / PNG, an eighth of the bytes over PCIe.
        `PIL.Image.frombytes('1', (w, h), packed.tobytes())` equals `Image.fromarray(mask)`."""
        m = self._dest(out, (self.h, (self.w + 7) // 8), 'download_mask_packed')
        fn = self.lib.mrchip_batch_download_mask_packed if wait else self.lib.mrchip_batch_download_mask_packed_async
        _lib.check(fn(self._h, page, _lib.ptr(m)), 'mrchip_batch_download_mask_packed')
        return m

    def done(self):
        """True when nothing is pending on the batch's stream (never blocks)."""
        rc = self.lib.mrchip_batch_done(self._h)
        if rc < 0:
            _lib.check(rc, 'mrchip_batch_done')
        return bool(rc)

    def download_layer(self, page, is_bg, size, out=None, wait=True):
        """fg (is_bg=0) / bg layer of one page.  `out`: destination array (e.g. Context.pinned_empty); with
        wait=False the copy is only enqueued (call sync() before reading) so the host can encode page i while
        page i+1 is still being copied / decomposed."""
        ow, oh = size
        shape = (oh, ow) if self.c == 1 else (oh, ow, 3)
        if out is None:
            out = np.empty(shape, dtype=np.uint8)
        elif out.shape != shape or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise ValueError('download_layer: out must be a C-contiguous uint8 array of shape %r' % (shape,))
        fn = self.lib.mrchip_batch_download_layer if wait else self.lib.mrchip_batch_download_layer_async
        _lib.check(fn(self._h, page, 1 if is_bg else 0, _lib.ptr(out)), 'mrchip_batch_download_layer')
        return out

    def box_decisions(self, page, nb):
        d = np.zeros(max(nb, 1), dtype=np.int32)
        _lib.check(self.lib.mrchip_batch_box_decisions(self._h, page, _lib.ptr(d, _lib.i32p), nb))
        return d[:nb].tolist()

    def sync(self):
        _lib.check(self.lib.mrchip_batch_sync(self._h), 'mrchip_batch_sync')


class _StreamJob:
    """One device batch travelling through decompose_stream: fill (upload + phase A), mid (phase B + layers +
    queued downloads), drain (wait, hand out the arrays)."""

    def __init__(self, slot, items):
        self.slot = slot            # _StreamSlot
        self.items = items          # [(pixels, gray, hocr)]
        self.sizes = None


class _StreamSlot:
    """A Batch plus the pinned host arrays its results are copied into; reused batch after batch."""

    def __init__(self, ctx, n, w, h, c):
        self.batch = Batch(ctx, n, w, h, c)
        self.ctx = ctx
        self.key = (n, w, h, c)
        self.stamp = 0              # StreamPool's clock at the last take / give
        self.out = {}               # name -> pinned array [n, ...]

    def pinned(self, name, shape):
        a = self.out.get(name)
        if a is None or a.shape != shape:
            a = self.ctx.pinned_empty(shape)
            self.out[name] = a
        return a

    def close(self):
        self.batch.close()
        self.out.clear()


def _slot_bytes(n, w, h, c):
    """Device bytes a _StreamSlot of n pages holds, roughly (planes of a batch) -- the page-locked result
    arrays on the host side are about a third of it."""
    return int(n) * (int(w) * int(h) * (10 + 6 * int(c)) + (4 << 20))


class StreamPool:
    """The device batches and page-locked result arrays of decompose_stream, kept between calls: creating them
    (tens of GB of hipMalloc / hipHostMalloc for 4000x3000 pages) costs more than decomposing a few hundred pages,
    so a long-running caller makes one pool and passes it to every decompose_stream(..., pool=pool).

    A slot serves batches of its geometry (w, h, c) with up to its `n` pages.  Books whose pages all differ in size
    would otherwise leave one idle slot per size behind: the pool keeps at most `max_bytes` of device memory
    (default: a quarter of the device) and closes idle slots of OTHER geometries, least recently used first, before
    it makes a new one; slots are sized to the run of pages they are made for, not to `batch_pages`."""

    def __init__(self, ctx=None, max_bytes=None):
        self.ctx = ctx or _lib.default_context()
        self.free = {}             # (w, h, c) -> [idle _StreamSlot], most recently used last
        self.every = []
        self.clock = 0
        if max_bytes is None:
            try:
                max_bytes = self.ctx.info()['hbm_bytes'] // 4
            except Exception:       # noqa: BLE001 - no device query: a fixed, modest cap
                max_bytes = 32 << 30
        self.max_bytes = int(max_bytes)

    def bytes_held(self):
        return sum(_slot_bytes(*sl.key) for sl in self.every)

    def _evict(self, need, keep_geo):
        """close idle slots (never one of geometry keep_geo unless nothing else is left) until `need` more bytes fit"""
        idle = sorted((sl for lst in self.free.values() for sl in lst), key=lambda sl: (sl.key[1:] == keep_geo, sl.stamp))
        for sl in idle:
            if self.bytes_held() + need <= self.max_bytes:
                break
            self.free[sl.key[1:]].remove(sl)
            self.every.remove(sl)
            sl.close()              # idle = drained: nothing of it is pending on its stream

    def take(self, n, w, h, c, capacity=None):
        """a slot for a batch of n pages of w x h x c: an idle one of that geometry with room for n, else a new one of
        `capacity` (>= n, default n) pages"""
        geo = (w, h, c)
        self.clock += 1
        idle = self.free.setdefault(geo, [])
        fit = [sl for sl in idle if sl.key[0] >= n]
        if fit:
            sl = min(fit, key=lambda sl: sl.key[0])
            idle.remove(sl)
        else:
            cap = max(n, capacity or n)
            self._evict(_slot_bytes(cap, w, h, c), geo)
            sl = _StreamSlot(self.ctx, cap, w, h, c)
            self.every.append(sl)
        sl.stamp = self.clock
        return sl

    def give(self, slot):
        self.clock += 1
        slot.stamp = self.clock
        self.free.setdefault(slot.key[1:], []).append(slot)

    def close(self):
        for sl in self.every:
            sl.close()
        self.every, self.free = [], {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def decompose_stream(pages, dpi=None, downsample=None, bg_downsample=None, fg_downsample=None,
                     denoise_mask=DENOISE_FAST, ctx=None, batch_pages=8, slots=4, mask_format='bool', copy=False,
                     pool=None, stats=None):
    """The page loop of recode.py:291-492 as a pipeline: `pages` is an iterable of (image, hocr_word_data);
    yields (mask, fg, bg) per page, in input order, equal to what create_mrc_hocr_components yields for it.

    Consecutive pages of one size and mode are collected into device batches of up to `batch_pages`; `slots`
    (>= 3) batches rotate, each on its own HIP stream: while batch i is being decomposed, batch i+1 crosses
    PCIe to the device and the results of batch i-1 cross back into page-locked host arrays (enqueue-only
    copies), so both directions of the link and the GPU are busy at once.  The defaults (8 pages, 4 slots) are what
    keeps the link busiest for 4000x3000 RGB pages: 1130 pages/s against 980 with batches of 32 on 3 slots -- with
    short batches two slots are usually downloading at once and fill each other's gaps between copies; the GPU
    still has 1.5x the link's capacity at 8 pages per launch (column strips of `optimise`).  Images that live in
    Context.pinned_empty arrays are uploaded by asynchronous DMA; any other array makes the HIP runtime
    stage the copy while this thread waits (the downloads and the kernels of the other batches still overlap).

    mask_format: 'bool' -- bool[h,w] like the reference's first yield; 'packed' -- uint8[h,(w+7)//8], 1 bit per
    pixel MSB first (PIL mode '1' / PBM rows, what mrc.encode_mrc_mask builds for jbig2; an eighth of the bytes).
    copy=False hands out views of the pinned result arrays, valid until `batch_pages` further pages have been
    taken from the generator; copy=True returns arrays the caller owns.  pool: a StreamPool to take the device
    batches and result arrays from (and leave them in); without one they are made and released by this call.
    stats: a dict that receives the host wall time spent in each phase (upload, boxes, phase A enqueue, the sigma
    wait, phase B + layers + download enqueue, the final wait) -- where a slow stream loses its time."""
    if denoise_mask not in (DENOISE_NONE, DENOISE_FAST):
        raise ValueError('Invalid denoise option:', denoise_mask)     # bregman is a host pass: use the generator
    if mask_format not in ('bool', 'packed'):
        raise ValueError("mask_format must be 'bool' or 'packed'")
    if slots < 3 or batch_pages < 1:
        raise ValueError('decompose_stream: slots >= 3 and batch_pages >= 1 expected')
    own_pool = pool is None
    if own_pool:
        pool = StreamPool(ctx)
    ctx = pool.ctx
    window = _window_size(dpi)
    take_slot = pool.take

    def tick(key, seconds):
        if stats is not None:
            stats[key] = stats.get(key, 0.0) + seconds

    def fill(items):
        h, w = items[0][0].shape[:2]
        c = 1 if items[0][0].ndim == 2 else 3
        # a full run gets (or makes) a slot of batch_pages pages; a short one -- a page size that occurs once or twice
        # in a row -- only what it needs, so a book of many sizes does not hold batch_pages pages per size
        job = _StreamJob(take_slot(len(items), w, h, c, capacity=batch_pages if len(items) == batch_pages else None), items)
        bt = job.slot.batch
        t0 = time()
        for j, (arr, gray, hocr) in enumerate(items):
            bt.upload(j, arr)
            if gray is not None:
                bt.upload_gray(j, gray)
        t1 = time()
        for j, (arr, gray, hocr) in enumerate(items):
            bt.set_boxes(j, hocr_boxes(hocr, w, h, downsample))
        bt.set_count(len(items))                     # a short last batch leaves the other pages untouched
        t2 = time()
        bt.mask_begin(window)
        tick('upload_s', t1 - t0)
        tick('boxes_s', t2 - t1)
        tick('phase_a_enqueue_s', time() - t2)
        return job

    def mid(job):
        bt = job.slot.batch
        n, w, h, c = job.slot.key
        t0 = time()
        sig = bt.sigmas()
        t1 = time()
        tick('sigma_wait_s', t1 - t0)
        bt.mask_finish(sig, denoise_mask == DENOISE_FAST)
        fgs, bgs, _ = bt.layers(fg_downsample, bg_downsample)
        job.sizes = (fgs, bgs)
        mshape = (n, h, (w + 7) // 8) if mask_format == 'packed' else (n, h, w)
        m = job.slot.pinned('mask', mshape)
        fg = job.slot.pinned('fg', (n, fgs[1], fgs[0]) + ((3,) if c == 3 else ()))
        bg = job.slot.pinned('bg', (n, bgs[1], bgs[0]) + ((3,) if c == 3 else ()))
        for j in range(len(job.items)):
            if mask_format == 'packed':
                bt.download_mask_packed(j, out=m[j], wait=False)
            else:
                bt.download_mask(j, out=m[j], wait=False)
            bt.download_layer(j, 0, fgs, out=fg[j], wait=False)
            bt.download_layer(j, 1, bgs, out=bg[j], wait=False)
        tick('phase_b_enqueue_s', time() - t1)

    def drain(job):
        bt = job.slot.batch
        t0 = time()
        bt.sync()
        tick('final_wait_s', time() - t0)
        m, fg, bg = job.slot.out['mask'], job.slot.out['fg'], job.slot.out['bg']
        for j in range(len(job.items)):
            mj = m[j] if mask_format == 'packed' else m[j].view(np.bool_)
            if copy:
                yield mj.copy(), fg[j].copy(), bg[j].copy()
            else:
                yield mj, fg[j], bg[j]
        job.items = None
        pool.give(job.slot)

    inflight = []                # jobs in pipeline order; [-1] filled, [-2] ready for mid, [0] drains next

    def advance(job):
        inflight.append(job)
        if len(inflight) >= 2 and inflight[-2].sizes is None:
            mid(inflight[-2])
        if len(inflight) >= slots:
            yield from drain(inflight.pop(0))

    try:
        run, key = [], None
        for image, hocr in pages:
            arr, gray = _image_to_array(image)
            k = (arr.shape[0], arr.shape[1], arr.ndim)
            if run and (k != key or len(run) == batch_pages):
                yield from advance(fill(run))
                run = []
            key = k
            run.append((arr, gray, hocr))
        if run:
            yield from advance(fill(run))
        for job in inflight:
            if job.sizes is None:
                mid(job)
        while inflight:
            yield from drain(inflight.pop(0))
    finally:
        for job in inflight:           # abandoned mid-stream: wait for what is queued, return the slots
            try:
                job.slot.batch.sync()
            except Exception:
                pass
            pool.give(job.slot)
        if own_pool:
            pool.close()


def decompose_pages(images, hocr_list, dpi=None, downsample=None, bg_downsample=None, fg_downsample=None,
                    denoise_mask=DENOISE_FAST, ctx=None, batch_pages=8, max_batch_bytes=None):
    """Batch form of create_mrc_hocr_components: the list of (mask, fg, bg) tuples the generator would yield
    page by page, in input order (arrays owned by the caller).  Pages of different sizes / modes may be mixed:
    consecutive pages of one geometry share a device batch (decompose_stream does the work)."""
    images = list(images)
    hocr_list = list(hocr_list)
    if len(images) != len(hocr_list):
        raise ValueError('decompose_pages: one hOCR page per image expected')
    # pages grouped by geometry so that every device batch is as full as it can be; results go back in input order
    geo = [np.asarray(im).shape if not hasattr(im, 'mode') else (im.size[1], im.size[0], im.mode) for im in images]
    order = sorted(range(len(images)), key=lambda i: (str(geo[i]), i))
    out = [None] * len(images)
    if max_batch_bytes is not None and images:       # cap on the device memory of one batch
        per_page = max(int(g[0]) * int(g[1]) * 28 + (4 << 20) for g in geo)
        batch_pages = max(1, min(batch_pages, int(max_batch_bytes // per_page)))
    gen = decompose_stream(((images[i], hocr_list[i]) for i in order), dpi=dpi, downsample=downsample,
                           bg_downsample=bg_downsample, fg_downsample=fg_downsample, denoise_mask=denoise_mask,
                           ctx=ctx, batch_pages=max(1, min(batch_pages, len(images))), copy=True)
    for i, res in zip(order, gen):
        out[i] = res
    return out


def _image_to_array(image):
    """(pixels, gray) from a PIL image or ndarray: pixels uint8 [h,w] or [h,w,3]; gray is None unless the
    image is a PIL image of a mode other than L / RGB.  For those the reference thresholds
    `image.convert('L')` of the ORIGINAL image (mrc.py:359-361) and converts to RGB only for the layers
    (mrc.py:401-404) -- Pillow's L conversion of e.g. YCbCr, CMYK or P is not the luma of the RGB conversion
    -- so both planes are made on the host by Pillow's own mode rules and uploaded."""
    gray = None
    if hasattr(image, 'mode'):
        if image.mode not in ('L', 'RGB'):
            gray = np.ascontiguousarray(np.array(image.convert('L')))
            image = image.convert('RGB')
        arr = np.array(image)
    else:
        arr = np.asarray(image)
    if arr.dtype != np.uint8 or arr.ndim not in (2, 3) or (arr.ndim == 3 and arr.shape[2] != 3):
        raise ValueError('expected an 8-bit L or RGB image, got dtype %s shape %s' % (arr.dtype, arr.shape))
    return np.ascontiguousarray(arr), gray


# kernels (mrchip_prof_* names) behind each of the reference's timing keys (mrc.py:363, 270, 308, 313, 327, 390, 418, 434,
# 452, 468).  The radix-select passes of the noise estimate serve the page estimate and the hOCR boxes' estimates alike;
# they are booked on est_1.
_STAGE_KERNELS = {
    'grey_conversion': ('luma601',),
    'hocr_mask_gen': ('sauvola_boxes', 'hocr_commit', 'dwt_dd_f64'),
    'est_1': ('dwt_dd_f32', 'median_reset', 'median_hist', 'median_scan'),
    'blur_1': ('gauss_fused', 'gauss_v', 'gauss_h'),
    'threshold': ('sauvola',),
    'fast_denoise': ('denoise_pack', 'denoise_solve', 'denoise_reconcile', 'denoise_unpack', 'denoise_jacobi', 'mask_pack_bits'),
    'partial_blur': ('optimise_rgb', 'optimise_gray'),
    'downsample': ('thumb_reduce', 'thumb_resize', 'thumb_resize_h', 'thumb_resize_v', 'thumb_resize_mm'),
}


class _StageClock:
    """Per-stage seconds for timing_data: the GPU time of the kernels behind each key, from the library's HIP-event
    profile (mrchip_prof_*), taken as differences of the running totals so that a caller's own profiling is left alone.

    The profile is per context: the figures are this page's only while nothing else runs on the context (two
    interleaved generators with timing_data, or a decompose_stream on another thread of the same context, would book each
    other's kernels).  Enabling is reference-counted per context, so the first generator to finish does not switch the
    profile off under a second one, and a caller who enabled profiling himself keeps it."""

    def __init__(self, ctx, on):
        self.ctx, self.on = ctx, on
        if on:
            users = getattr(ctx, '_stage_clock_users', 0)
            if users == 0:
                ctx._stage_clock_was_on = bool(getattr(ctx, 'prof_on', False))
                if not ctx._stage_clock_was_on:
                    ctx.prof_enable(True)
            ctx._stage_clock_users = users + 1
            self.last = ctx.prof_report()

    def lap(self):
        """seconds of GPU time per stage since the previous lap"""
        if not self.on:
            return {}
        now = self.ctx.prof_report()
        out = {}
        for key, names in _STAGE_KERNELS.items():
            out[key] = sum(now[k]['ms'] - self.last.get(k, {'ms': 0.0})['ms'] for k in names if k in now) * 1e-3
        self.last = now
        return out

    def close(self):
        if self.on:
            self.on = False
            self.ctx._stage_clock_users -= 1
            if self.ctx._stage_clock_users == 0 and not self.ctx._stage_clock_was_on:
                self.ctx.prof_enable(False)


def _book(timing_data, wall, main_key, entries):
    """append (key, seconds) in the reference's order: every key gets the GPU time of its kernels, the key that stands
    for the phase also the rest of the phase's wall time (host work, PCIe, launch gaps), so the keys add up to `wall`"""
    rest = max(0.0, wall - sum(sec for _, sec in entries))
    for key, sec in entries:
        timing_data.append((key, sec + (rest if key == main_key else 0.0)))


def create_mrc_hocr_components(image, hocr_word_data,
                               dpi=None,
                               downsample=None,
                               bg_downsample=None,
                               fg_downsample=None,
                               denoise_mask=None, timing_data=None,
                               errors=None, ctx=None):
    """mrc.create_mrc_hocr_components (mrc.py:334-471): generator yielding mask (bool[h,w]),
    foreground and background (uint8 arrays), lazily, with the reference's timing keys.
    One check runs when the function is CALLED rather than at the first next(): denoise_mask='bregman' without
    scikit-image raises ImportError at once (every other argument error keeps the reference's timing)."""
    if denoise_mask == DENOISE_BREGMAN:
        _require_skimage_for_bregman()
    return _mrc_hocr_components(image, hocr_word_data, dpi, downsample, bg_downsample, fg_downsample, denoise_mask,
                                timing_data, errors, ctx)


def _mrc_hocr_components(image, hocr_word_data, dpi, downsample, bg_downsample, fg_downsample, denoise_mask, timing_data,
                         errors, ctx):
    image_arr, gray_arr = _image_to_array(image)
    height_, width_ = image_arr.shape[:2]
    channels = 1 if image_arr.ndim == 2 else 3
    ctx = ctx or _lib.default_context()
    page = _Page(ctx, width_, height_, channels)
    clock = _StageClock(ctx, timing_data is not None)
    try:
        t = time()
        page.upload(image_arr)
        if gray_arr is not None:
            page.upload_gray(gray_arr)
        boxes = hocr_boxes(hocr_word_data, width_, height_, downsample)
        page.mask_begin(boxes, _window_size(dpi))
        sigma_est = page.sigma()
        now = time()
        if timing_data is not None:
            # the reference's keys and order (mrc.py:363, 270, 308); each key = the GPU time of its kernels, the upload, the
            # box parsing and the wait for the estimate go to hocr_mask_gen
            g = clock.lap()
            _book(timing_data, now - t, 'hocr_mask_gen',
                  ([('grey_conversion', g['grey_conversion'])] if channels == 3 else []) +
                  [('hocr_mask_gen', g['hocr_mask_gen']), ('est_1', g['est_1'])])
        t = time()
        page.mask_finish(sigma_est, denoise_mask == DENOISE_FAST)
        mask_arr = page.download_mask()
        now = time()
        if timing_data is not None:
            g = clock.lap()
            # (the commit of the hOCR-box decisions runs in this phase: its kernels stay on the threshold key's remainder)
            _book(timing_data, now - t, 'threshold',
                  ([('blur_1', g['blur_1'])] if sigma_est > 1.0 else []) +                      # mrc.py:313
                  [('threshold', g['threshold'])] +                                             # mrc.py:327
                  ([('fast_denoise', g['fast_denoise'])] if denoise_mask == DENOISE_FAST else []))   # mrc.py:390
        if denoise_mask == DENOISE_BREGMAN:                                   # mrc.py:391-394, on the host
            t = time()
            mask_arr = denoise_bregman(mask_arr)
            page.upload_mask(mask_arr)
            if timing_data is not None:
                timing_data.append(('denoise', time() - t))
        elif denoise_mask not in (DENOISE_NONE, DENOISE_FAST):
            raise ValueError('Invalid denoise option:', denoise_mask)         # mrc.py:396
        # The reference yields the very array its fg / bg stages read again (mrc.py:399, 413, 439): a caller that edits the
        # mask between two next() calls gets layers of the edited mask.  The device keeps its own copy, so the yielded
        # array is compared with what was yielded (one 1-byte-per-pixel copy and compare per stage, SHARED_MASK) and
        # uploaded again where it changed; a change after the joint fg / bg launch recomputes the layers for bg.
        mask_seen = mask_arr.copy() if SHARED_MASK else None

        def mask_edited():
            nonlocal mask_seen
            if mask_seen is None or _same_bytes(mask_arr, mask_seen):
                return False
            page.upload_mask(np.ascontiguousarray(mask_arr, dtype=np.bool_).view(np.uint8))
            mask_seen = mask_arr.copy()
            return True
        yield mask_arr

        sizes = None
        glay = {}
        for is_bg, ds, key in ((0, fg_downsample, 'fg'), (1, bg_downsample, 'bg')):
            t = time()
            if sizes is None:
                # both layers are made when the first one is asked for, in one launch: a single page leaves the chip
                # nearly idle, the two page-layers run side by side (a caller that stops after the mask, recode.py:400-408,
                # never gets here).  The launch goes out BEFORE the yielded mask is compared with what was yielded: the
                # compare runs on the host while the kernels run, and only a caller that did edit the mask pays a second launch
                sizes = page.layers(fg_downsample, bg_downsample)
            if mask_edited():
                sizes = page.layers(fg_downsample, bg_downsample)
            (ow, oh), too_small = sizes[is_bg], bool(sizes[2] & (1 << is_bg))
            arr = page.download_layer(is_bg, ow, oh)
            now = time()
            if timing_data is not None:
                if not glay:
                    # one optimise launch holds both page-layers (equal pixel counts: half each); the thumbnail kernels of
                    # the layers that are downsampled likewise
                    g = clock.lap()
                    nds = (fg_downsample is not None) + (bg_downsample is not None)
                    glay = {'blur': g['partial_blur'] / 2, 'ds': g['downsample'] / max(nds, 1)}
                # mrc.py:418, 434, 452, 468; the wall time of this yield (fg: the joint launch and its download, bg: its
                # download) beyond this layer's kernels stays on the partial_blur key
                _book(timing_data, max(now - t, glay['blur'] + (glay['ds'] if ds is not None else 0.0)),
                      '%s_partial_blur' % key,
                      [('%s_partial_blur' % key, glay['blur'])] + ([('%s_downsample' % key, glay['ds'])] if ds is not None else []))
            if ds is not None:
                if too_small and errors is not None:
                    errors.add(RECODE_RUNTIME_WARNING_TOO_SMALL_TO_DOWNSAMPLE)  # mrc.py:429-431
            yield arr
    finally:
        clock.close()
        page.close()
    return


def packed_mask_to_pbm(packed, w, h):
    """Raw PBM (P4) file bytes of a packed mask (Batch.download_mask_packed).  PBM's 1 = black is the
    mask's True = foreground, the polarity jbig2 and mrc.encode_mrc_mask's PNG use."""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    if packed.shape != (h, (w + 7) // 8):
        raise ValueError('packed mask shape %r does not match %dx%d' % (packed.shape, w, h))
    return b'P4\n%d %d\n' % (w, h) + packed.tobytes()


_FILTERS = {'bicubic': 0, 'lanczos': 1}


def thumbnail(arr, size, resample='bicubic', reducing_gap=2.0, ctx=None):
    """`im = Image.fromarray(arr); im.thumbnail(size, resample=..., reducing_gap=...); np.array(im)` on the
    device.  `size` may hold floats like PIL's (floored).  The layer downsample of mrc.py:422-428 is the
    default (BICUBIC, 2.0); the page-ingest downsample of recode.py:368-372 is
    `thumbnail(arr, (w / downsample, h / downsample), resample='lanczos', reducing_gap=None)`."""
    import math
    a = np.ascontiguousarray(arr, dtype=np.uint8)
    if a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
        raise ValueError('thumbnail: uint8[H,W] or uint8[H,W,3] expected')
    if resample not in _FILTERS:
        raise ValueError('thumbnail: resample must be one of %s' % sorted(_FILTERS))
    h, w = a.shape[:2]
    c = 1 if a.ndim == 2 else 3
    rw, rh = math.floor(size[0]), math.floor(size[1])
    if rw <= 0 or rh <= 0:
        raise ValueError('thumbnail: requested size must be positive')
    lib = _lib.load()
    ctx = ctx or _lib.default_context()
    ow, oh = C.c_int(), C.c_int()
    lib.mrchip_thumbnail_size(w, h, rw, rh, C.byref(ow), C.byref(oh))
    out = np.empty((oh.value, ow.value) if c == 1 else (oh.value, ow.value, 3), np.uint8)
    _lib.check(lib.mrchip_thumbnail_ex(ctx.handle, _lib.ptr(a), w, h, c, rw, rh, _FILTERS[resample],
                                       float(reducing_gap) if reducing_gap else 0.0, _lib.ptr(out)), 'mrchip_thumbnail_ex')
    return out


def layer_to_pnm(arr):
    """Binary PGM (P5) / PPM (P6) file bytes of a uint8 layer -- what `Image.fromarray(arr).save('x.pnm')`
    writes and what the JPEG2000 encoders of mrc.encode_mrc_images read (mrc.py:523-580, jpeg2000.py:44-84)
    -- without the PIL round trip."""
    a = np.ascontiguousarray(arr, dtype=np.uint8)
    if a.ndim == 2:
        head = b'P5\n%d %d\n255\n' % (a.shape[1], a.shape[0])
    elif a.ndim == 3 and a.shape[2] == 3:
        head = b'P6\n%d %d\n255\n' % (a.shape[1], a.shape[0])
    else:
        raise ValueError('layer_to_pnm: uint8[H,W] or uint8[H,W,3] expected')
    return head + a.tobytes()
