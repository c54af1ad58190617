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


def mean_estimate_sigma(arr, ctx=None):
    """mrc.mean_estimate_sigma (mrc.py:52-55) for float32 images holding uint8 values or bool arrays."""
    a = np.asarray(arr)
    ctx = ctx or _lib.default_context()
    sigma = C.c_double()
    if a.dtype == np.bool_:
        src = np.ascontiguousarray(a).view(np.uint8)
        kind = 1
    else:
        src = np.ascontiguousarray(a, dtype=np.uint8)
        if not np.array_equal(src, a):
            raise _lib.MrchipError('mean_estimate_sigma: only uint8-valued images and bool arrays are supported')
        kind = 0
    h, w = src.shape
    _lib.check(_lib.load().mrchip_estimate_sigma(ctx.handle, _lib.ptr(src), w, w, h, kind, C.byref(sigma)),
               'mrchip_estimate_sigma')
    return sigma.value


def estimate_noise(imgf, ctx=None):
    """mrc.estimate_noise (mrc.py:273-296) on float32(gray)."""
    a = np.asarray(imgf)
    src = np.ascontiguousarray(a, dtype=np.uint8)
    h, w = src.shape
    ctx = ctx or _lib.default_context()
    sigma = C.c_double()
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


def hocr_boxes(hocr_word_data, image_width, image_height, downsample=None):
    """Text / confidence / geometry filter of mrc.create_hocr_mask (mrc.py:194-221): host logic."""
    boxes = []
    for paragraph in hocr_word_data:
        for line in paragraph['lines']:
            coords = line['bbox']

            line_text = ' '.join([word['text'] for word in line['words']])
            line_confs = [word['confidence'] for word in line['words']]
            line_conf = sum(line_confs) / len(line_confs) if len(line_confs) else 0

            if line_text.strip() == '' or line_conf < 20:
                continue

            if downsample is not None:
                coords = [int(x / downsample) for x in coords]
            else:
                coords = [int(x) for x in coords]

            left, top, right, bottom = coords
            if left == right or top == bottom:
                continue

            if (left >= right) or (top >= bottom):
                print('Invalid bounding box: (%d, %d, %d, %d)' % (left, top, right, bottom), file=sys.stderr)
                continue

            if (left < 0) or (right > image_width) or (top < 0) or (bottom > image_height):
                print('Invalid bounding box outside image: (%d, %d, %d, %d)' % (left, top, right, bottom),
                      file=sys.stderr)
                continue
            boxes.append((left, top, right, bottom))
    return np.ascontiguousarray(np.asarray(boxes, dtype=np.int32).reshape(-1, 4))


def create_hocr_mask(img, mask_arr, hocr_word_data, downsample=None, dpi=None, timing_data=None, ctx=None):
    """mrc.create_hocr_mask (mrc.py:188-270): img is a PIL 'L' image or uint8[h,w]; mask_arr modified in place."""
    np_img = _lib.as_u8(np.array(img), 'img')
    image_height, image_width = np_img.shape
    t = time()
    boxes = hocr_boxes(hocr_word_data, image_width, image_height, downsample)
    if len(boxes):
        m = np.asarray(mask_arr)
        tmp = _lib.as_u8(m, 'mask_arr')
        ctx = ctx or _lib.default_context()
        _lib.check(_lib.load().mrchip_hocr_mask(ctx.handle, _lib.ptr(np_img), _lib.ptr(tmp), image_width, image_height,
                                                _lib.ptr(boxes, _lib.i32p), len(boxes), _window_size(dpi), None),
                   'mrchip_hocr_mask')
        if not np.shares_memory(tmp, m):
            m[...] = tmp.view(m.dtype) if m.dtype == np.bool_ else tmp
    if timing_data is not None:
        timing_data.append(('hocr_mask_gen', time() - t))


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
            self.close()
        except Exception:
            pass

    def upload(self, arr):
        _lib.check(self.lib.mrchip_page_upload(self._h, _lib.ptr(arr)), 'mrchip_page_upload')

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

    def close(self):
        if self._h:
            self.lib.mrchip_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, page, arr):
        arr = _lib.as_u8(arr)
        _lib.check(self.lib.mrchip_batch_upload(self._h, page, _lib.ptr(arr)), 'mrchip_batch_upload')

    def set_boxes(self, page, boxes):
        boxes = np.ascontiguousarray(boxes, dtype=np.int32).reshape(-1, 4)
        _lib.check(self.lib.mrchip_batch_set_boxes(self._h, page, _lib.ptr(boxes, _lib.i32p) if len(boxes) else None,
                                                   len(boxes)), 'mrchip_batch_set_boxes')

    def mask_begin(self, window):
        _lib.check(self.lib.mrchip_batch_mask_begin(self._h, window), 'mrchip_batch_mask_begin')

    def sigmas(self):
        s = np.zeros(self.n, dtype=np.float64)
        _lib.check(self.lib.mrchip_batch_sigmas(self._h, _lib.ptr(s, _lib.f64p)), 'mrchip_batch_sigmas')
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

    def download_mask(self, page):
        m = np.empty((self.h, self.w), dtype=np.uint8)
        _lib.check(self.lib.mrchip_batch_download_mask(self._h, page, _lib.ptr(m)), 'mrchip_batch_download_mask')
        return m.view(np.bool_)

    def download_mask_packed(self, page):
        """The finished mask at 1 bit per pixel (MSB first, rows of ceil(w/8) bytes): what
        mrc.encode_mrc_mask (mrc.py:474-520) feeds to jbig2 / PNG, an eighth of the bytes over PCIe.
        `PIL.Image.frombytes('1', (w, h), packed.tobytes())` equals `Image.fromarray(mask)`."""
        out = np.empty((self.h, (self.w + 7) // 8), dtype=np.uint8)
        _lib.check(self.lib.mrchip_batch_download_mask_packed(self._h, page, _lib.ptr(out)),
                   'mrchip_batch_download_mask_packed')
        return out

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


def decompose_pages(images, hocr_list, dpi=None, downsample=None, bg_downsample=None, fg_downsample=None,
                    denoise_mask=DENOISE_FAST, ctx=None, max_batch_bytes=64 << 30):
    """Batch form of create_mrc_hocr_components: returns a list of (mask, fg, bg) tuples equal to what the
    generator yields page by page, in input order.  Pages are grouped by (width, height, mode) -- a device
    batch holds pages of one size -- and each group is cut into batches of at most `max_batch_bytes` of
    device memory."""
    arrs = [_image_to_array(im) for im in images]
    if len(arrs) != len(hocr_list):
        raise ValueError('decompose_pages: one hOCR page per image expected')
    if denoise_mask not in (DENOISE_NONE, DENOISE_FAST):
        raise ValueError('Invalid denoise option:', denoise_mask)
    ctx = ctx or _lib.default_context()
    groups = {}
    for i, a in enumerate(arrs):
        groups.setdefault((a.shape[0], a.shape[1], 1 if a.ndim == 2 else 3), []).append(i)
    out = [None] * len(arrs)
    for (h, w, c), idx in groups.items():
        per_page = w * h * (10 + 6 * c) + (4 << 20)          # planes of a page in a batch
        step = max(1, int(max_batch_bytes // per_page))
        for k in range(0, len(idx), step):
            part = idx[k:k + step]
            b = Batch(ctx, len(part), w, h, c)
            try:
                for j, i in enumerate(part):
                    b.upload(j, arrs[i])
                    b.set_boxes(j, hocr_boxes(hocr_list[i], w, h, downsample))
                b.mask_begin(_window_size(dpi))
                b.mask_finish(b.sigmas(), denoise_mask == DENOISE_FAST)
                fgs, bgs, _ = b.layers(fg_downsample, bg_downsample)
                for j, i in enumerate(part):
                    out[i] = (b.download_mask(j), b.download_layer(j, 0, fgs), b.download_layer(j, 1, bgs))
            finally:
                b.close()
    return out


def _image_to_array(image):
    """(array uint8 [h,w] or [h,w,3], had_grey_conversion_mode) from a PIL image or ndarray."""
    if hasattr(image, 'mode'):
        if image.mode not in ('L', 'RGB'):
            # mrc.py:401-404 converts other modes to RGB for the layers; convert('L') of such an
            # image (mrc.py:361) goes through Pillow's own mode rules and is not on the GPU path
            image = image.convert('RGB')
        arr = np.array(image)
    else:
        arr = np.asarray(image)
    if arr.dtype != np.uint8 or arr.ndim not in (2, 3) or (arr.ndim == 3 and arr.shape[2] != 3):
        raise ValueError('expected an 8-bit L or RGB image, got dtype %s shape %s' % (arr.dtype, arr.shape))
    return np.ascontiguousarray(arr)


def create_mrc_hocr_components(image, hocr_word_data,
                               dpi=None,
                               downsample=None,
                               bg_downsample=None,
                               fg_downsample=None,
                               denoise_mask=None, timing_data=None,
                               errors=None, ctx=None):
    """mrc.create_mrc_hocr_components (mrc.py:334-471): generator yielding mask (bool[h,w]),
    foreground and background (uint8 arrays), lazily, with the reference's timing keys."""
    image_arr = _image_to_array(image)
    height_, width_ = image_arr.shape[:2]
    channels = 1 if image_arr.ndim == 2 else 3
    if denoise_mask not in (DENOISE_NONE, DENOISE_FAST):
        if denoise_mask == DENOISE_BREGMAN:
            raise NotImplementedError("denoise_mask='bregman' (mrc.py:90-108) is outside the GPU hot path")
    ctx = ctx or _lib.default_context()
    page = _Page(ctx, width_, height_, channels)
    try:
        t = time()
        page.upload(image_arr)
        boxes = hocr_boxes(hocr_word_data, width_, height_, downsample)
        page.mask_begin(boxes, _window_size(dpi))
        sigma_est = page.sigma()
        now = time()
        if timing_data is not None:
            # the GPU runs these stages back to back; the split of the elapsed time keeps the
            # reference's keys and order (mrc.py:363, 270, 308)
            if channels == 3:
                timing_data.append(('grey_conversion', 0.0))
            timing_data.append(('hocr_mask_gen', now - t))
            timing_data.append(('est_1', 0.0))
        if denoise_mask not in (DENOISE_NONE, DENOISE_FAST):
            raise ValueError('Invalid denoise option:', denoise_mask)         # mrc.py:396
        t = time()
        page.mask_finish(sigma_est, denoise_mask == DENOISE_FAST)
        mask_arr = page.download_mask()
        now = time()
        if timing_data is not None:
            if sigma_est > 1.0:
                timing_data.append(('blur_1', 0.0))                           # mrc.py:313
            timing_data.append(('threshold', now - t))                        # mrc.py:327
            if denoise_mask == DENOISE_FAST:
                timing_data.append(('fast_denoise', 0.0))                     # mrc.py:390
        yield mask_arr

        for is_bg, ds, key in ((0, fg_downsample, 'fg'), (1, bg_downsample, 'bg')):
            t = time()
            ow, oh, too_small = page.layer(is_bg, ds)
            arr = page.download_layer(is_bg, ow, oh)
            now = time()
            if timing_data is not None:
                timing_data.append(('%s_partial_blur' % key, now - t))        # mrc.py:418, 452
            if ds is not None:
                if too_small and errors is not None:
                    errors.add(RECODE_RUNTIME_WARNING_TOO_SMALL_TO_DOWNSAMPLE)  # mrc.py:429-431
                if timing_data is not None:
                    timing_data.append(('%s_downsample' % key, 0.0))          # mrc.py:434, 468
            yield arr
    finally:
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
