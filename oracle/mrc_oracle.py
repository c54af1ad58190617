"""ctypes front-end of the parity oracle (oracle/mrc_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of mrc_oracle.c.  Imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package.

The functions mirror the reference's Python surface (internetarchivepdf/mrc.py,
cython/sauvola.pyx, cython/optimiser.pyx) on numpy arrays so that parity tests
read like calls into the reference.
"""
import ctypes as C
import math
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)


def build(force=False):
    so = os.path.join(_HERE, 'libmrc_oracle.so')
    src = os.path.join(_HERE, 'mrc_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'libmrc_oracle.so'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_sauvola.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
        L.orc_threshold_image.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, u8p]
        L.orc_denoise.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_denoise.restype = None
        L.orc_optimise.argtypes = [u8p, u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_optimise_spec.argtypes = L.orc_optimise.argtypes
        L.orc_luma601.argtypes = [u8p, u8p, C.c_size_t]
        L.orc_luma601.restype = None
        L.orc_dwt_dd_f32.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p]
        L.orc_dwt_dd_f64.argtypes = [f64p, C.c_int, C.c_int, C.c_int, f64p]
        L.orc_sigma_f32.argtypes = [f32p, C.c_int, C.c_int, C.c_int]
        L.orc_sigma_f32.restype = C.c_double
        L.orc_sigma_bool.argtypes = [u8p, C.c_int, C.c_int, C.c_int]
        L.orc_sigma_bool.restype = C.c_double
        L.orc_estimate_noise.argtypes = [f32p, C.c_int, C.c_int]
        L.orc_estimate_noise.restype = C.c_double
        L.orc_gaussian_weights.argtypes = [C.c_double, f64p]
        L.orc_gaussian_f32.argtypes = [f32p, f32p, C.c_int, C.c_int, f64p, C.c_int]
        L.orc_thumbnail_size.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_reduce.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
        L.orc_reduce.restype = None
        L.orc_resize_bicubic.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, u8p]
        L.orc_thumbnail.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
        L.orc_hocr_mask.argtypes = [u8p, u8p, C.c_int, C.c_int, i32p, C.c_int, C.c_int, i32p]
        L.orc_threshold_mask.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, f64p, f64p]
        L.orc_page_mask.argtypes = [u8p, C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.c_int, C.c_int,
                                    u8p, f64p, f64p]
        L.orc_page_layer.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, u8p,
                                     C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def _u8(a):
    """contiguous uint8 view/copy (bool arrays are reinterpreted)."""
    a = np.asarray(a)
    if a.dtype == np.bool_:
        a = a.view(np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


# --- cython/sauvola.pyx ----------------------------------------------------
def binarise_sauvola(in_arr, out_arr, width, height, window_width, window_height, k, R):
    """sauvola.binarise_sauvola (cython/sauvola.pyx:29); out_arr filled in place."""
    src = _u8(in_arr)
    dst = np.empty(width * height, dtype=np.uint8)
    rc = lib().orc_sauvola(_p(src, u8p), _p(dst, u8p), width, height, window_width, window_height, k, R)
    assert rc == 0
    out_arr.reshape(-1).view(np.uint8)[:] = dst
    return 0


def window_size(dpi):
    """mrc.py:68-75"""
    window = 51
    if dpi is not None:
        window = int(dpi / 4)
        if window % 2 == 0:
            window += 1
    return window


def threshold_image(img, dpi, k=0.34):
    """mrc.threshold_image (mrc.py:58-87)"""
    h, w = img.shape
    src = _u8(img)
    out = np.empty((h, w), dtype=np.uint8)
    rc = lib().orc_threshold_image(_p(src, u8p), w, w, h, window_size(dpi), k, 0, _p(out, u8p))
    assert rc == 0
    return out.view(np.bool_)


# --- cython/optimiser.pyx ---------------------------------------------------
def fast_mask_denoise(mask, width, height, mincnt, n_size):
    """optimiser.fast_mask_denoise (optimiser.pyx:436); in place."""
    m = _u8(mask)
    lib().orc_denoise(_p(m, u8p), width, height, mincnt, n_size)
    if m is not mask and not np.shares_memory(m, mask):
        mask[...] = m.view(mask.dtype)
    return mask


def _optimise(fn, mask, img, width, height, n_size, c, invert=0):
    m = _u8(mask)
    i = _u8(img)
    out = np.empty_like(i)
    rc = fn(_p(m, u8p), _p(i, u8p), _p(out, u8p), width, height, c, n_size, invert)
    assert rc == 0
    return out


def optimise_gray2(mask, img, width, height, n_size):
    return _optimise(lib().orc_optimise, mask, img, width, height, n_size, 1)


def optimise_rgb2(mask, img, width, height, n_size):
    return _optimise(lib().orc_optimise, mask, img, width, height, n_size, 3)


def optimise_gray(mask, img, width, height, n_size):
    return _optimise(lib().orc_optimise_spec, mask, img, width, height, n_size, 1)


def optimise_rgb(mask, img, width, height, n_size):
    return _optimise(lib().orc_optimise_spec, mask, img, width, height, n_size, 3)


# --- third-party pieces -----------------------------------------------------
def luma601(rgb):
    rgb = _u8(rgb)
    out = np.empty(rgb.shape[:2], dtype=np.uint8)
    lib().orc_luma601(_p(rgb, u8p), _p(out, u8p), out.size)
    return out


def dwt_dd(arr):
    arr = np.asarray(arr)
    h, w = arr.shape
    if arr.dtype == np.float32:
        a = np.ascontiguousarray(arr)
        dd = np.empty(((h + 3) // 2, (w + 3) // 2), dtype=np.float32)
        lib().orc_dwt_dd_f32(_p(a, f32p), w, h, w, _p(dd, f32p))
    else:
        a = np.ascontiguousarray(arr, dtype=np.float64)
        dd = np.empty(((h + 3) // 2, (w + 3) // 2), dtype=np.float64)
        lib().orc_dwt_dd_f64(_p(a, f64p), w, h, w, _p(dd, f64p))
    return dd


def estimate_sigma(arr):
    """skimage.restoration.estimate_sigma for float32 or bool 2-D arrays."""
    arr = np.asarray(arr)
    h, w = arr.shape
    if arr.dtype == np.float32:
        a = np.ascontiguousarray(arr)
        return lib().orc_sigma_f32(_p(a, f32p), w, h, w)
    a = _u8(arr)
    return lib().orc_sigma_bool(_p(a, u8p), w, h, w)


def estimate_noise(imgf):
    """mrc.estimate_noise (mrc.py:273-296)"""
    a = np.ascontiguousarray(imgf, dtype=np.float32)
    h, w = a.shape
    return lib().orc_estimate_noise(_p(a, f32p), h, w)


def gaussian_weights_numpy(sigma):
    """scipy.ndimage _gaussian_kernel1d (order 0) exactly as scipy builds it (numpy exp)."""
    radius = int(4.0 * float(sigma) + 0.5)
    sigma2 = sigma * sigma
    x = np.arange(-radius, radius + 1)
    phi_x = np.exp(-0.5 / sigma2 * x ** 2)
    phi_x = phi_x / phi_x.sum()
    return np.ascontiguousarray(phi_x[::-1], dtype=np.float64), radius


def gaussian_weights_libm(sigma):
    radius = int(4.0 * float(sigma) + 0.5)
    w = np.empty(2 * radius + 1, dtype=np.float64)
    r = lib().orc_gaussian_weights(sigma, _p(w, f64p))
    assert r == radius
    return w, radius


def gaussian_filter(imgf, sigma, weights=None):
    a = np.ascontiguousarray(imgf, dtype=np.float32)
    h, w = a.shape
    if weights is None:
        weights, radius = gaussian_weights_numpy(sigma)
    else:
        radius = (len(weights) - 1) // 2
    out = np.empty_like(a)
    rc = lib().orc_gaussian_f32(_p(a, f32p), _p(out, f32p), h, w, _p(weights, f64p), radius)
    assert rc == 0
    return out


def thumbnail_size(w, h, req_w, req_h):
    ow, oh = C.c_int(), C.c_int()
    changed = lib().orc_thumbnail_size(w, h, req_w, req_h, C.byref(ow), C.byref(oh))
    return ow.value, oh.value, bool(changed)


def thumbnail(arr, req_w, req_h):
    """np.array(Image.fromarray(arr).thumbnail((req_w, req_h)))"""
    a = _u8(arr)
    h, w = a.shape[:2]
    c = 1 if a.ndim == 2 else a.shape[2]
    ow, oh, changed = thumbnail_size(w, h, req_w, req_h)
    if not changed:
        return a.copy()
    out = np.empty((oh, ow) if a.ndim == 2 else (oh, ow, c), dtype=np.uint8)
    rc = lib().orc_thumbnail(_p(a, u8p), w, h, c, req_w, req_h, _p(out, u8p))
    assert rc == 0
    return out


FILTERS = {'bicubic': 0, 'lanczos': 1}


def thumbnail_ex(arr, req_w, req_h, resample='bicubic', reducing_gap=2.0):
    """np.array of Image.fromarray(arr).thumbnail((req_w, req_h), resample=..., reducing_gap=...);
    reducing_gap=None skips Image.reduce (the page-ingest downsample, recode.py:368-372)."""
    a = _u8(arr)
    h, w = a.shape[:2]
    c = 1 if a.ndim == 2 else a.shape[2]
    ow, oh, changed = thumbnail_size(w, h, int(req_w), int(req_h))
    if not changed:
        return a.copy()
    out = np.empty((oh, ow) if a.ndim == 2 else (oh, ow, c), dtype=np.uint8)
    L = lib()
    L.orc_thumbnail_ex.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, u8p]
    rc = L.orc_thumbnail_ex(_p(a, u8p), w, h, c, int(req_w), int(req_h), FILTERS[resample],
                            float(reducing_gap) if reducing_gap else 0.0, _p(out, u8p))
    assert rc == 0
    return out


def reduce(arr, fx, fy):
    a = _u8(arr)
    h, w = a.shape[:2]
    c = 1 if a.ndim == 2 else a.shape[2]
    oh, ow = (h + fy - 1) // fy, (w + fx - 1) // fx
    out = np.empty((oh, ow) if a.ndim == 2 else (oh, ow, c), dtype=np.uint8)
    lib().orc_reduce(_p(a, u8p), w, h, c, fx, fy, _p(out, u8p))
    return out


# --- mrc.py orchestration ---------------------------------------------------
def hocr_boxes(hocr_word_data, image_width, image_height, downsample=None, log=sys.stderr):
    """The text/confidence/geometry filter of mrc.create_hocr_mask (mrc.py:194-221).
    Returns an int32 [nb,4] array of (left, top, right, bottom) in list order."""
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
                print('Invalid bounding box: (%d, %d, %d, %d)' % (left, top, right, bottom), file=log)
                continue
            if (left < 0) or (right > image_width) or (top < 0) or (bottom > image_height):
                print('Invalid bounding box outside image: (%d, %d, %d, %d)' % (left, top, right, bottom),
                      file=log)
                continue
            boxes.append((left, top, right, bottom))
    return np.asarray(boxes, dtype=np.int32).reshape(-1, 4)


def create_hocr_mask(gray, mask_arr, boxes, dpi=None, decisions=None):
    g = _u8(gray)
    h, w = g.shape
    m = mask_arr.view(np.uint8)
    assert m.flags.c_contiguous
    boxes = np.ascontiguousarray(boxes, dtype=np.int32)
    dec = np.zeros(max(1, len(boxes)), dtype=np.int32)
    rc = lib().orc_hocr_mask(_p(g, u8p), _p(m, u8p), w, h, _p(boxes, i32p), len(boxes),
                             window_size(dpi), _p(dec, i32p))
    assert rc == 0
    if decisions is not None:
        decisions.extend(dec[:len(boxes)].tolist())


def create_mrc_hocr_components(image, hocr_word_data, dpi=None, downsample=None, bg_downsample=None,
                               fg_downsample=None, denoise_mask=None, timing_data=None, errors=None,
                               info=None, gauss_weights='numpy'):
    """Oracle twin of mrc.create_mrc_hocr_components (mrc.py:334-471).

    `image` is a uint8 ndarray [H,W] ('L') or [H,W,3] ('RGB') (or a PIL image).
    """
    gray_of_original = None
    if hasattr(image, 'mode'):
        if image.mode not in ('L', 'RGB'):
            # mrc.py:359-361 thresholds image.convert('L') of the ORIGINAL image (Pillow's own rule for that mode, not the
            # luma of the RGB conversion); mrc.py:401-404 converts to RGB for the layers only
            gray_of_original = _u8(np.array(image.convert('L')))
            image = image.convert('RGB')
        image = np.array(image)
    img = _u8(image)
    h, w = img.shape[:2]
    c = 1 if img.ndim == 2 else 3
    if denoise_mask not in ('none', 'fast'):
        if denoise_mask == 'bregman':
            raise NotImplementedError('bregman denoise is out of scope (SURVEY.md 2 #11)')
        raise ValueError('Invalid denoise option:', denoise_mask)
    gray = gray_of_original if gray_of_original is not None else (luma601(img) if c == 3 else img)
    mask = np.zeros((h, w), dtype=np.bool_)
    boxes = hocr_boxes(hocr_word_data, w, h, downsample)
    dec = []
    create_hocr_mask(gray, mask, boxes, dpi, dec)
    sigma_est = estimate_noise(gray.astype(np.float32))
    wts = None
    if sigma_est > 1.0 and gauss_weights == 'numpy':
        wts, _ = gaussian_weights_numpy(sigma_est * 0.1)
    sig = C.c_double()
    rc = lib().orc_threshold_mask(_p(mask.view(np.uint8), u8p), _p(gray, u8p), w, h, window_size(dpi),
                                  C.byref(sig), _p(wts, f64p) if wts is not None else None)
    assert rc == 0
    if info is not None:
        info.update(sigma_est=sig.value, decisions=dec, boxes=boxes)
    if denoise_mask == 'fast':
        fast_mask_denoise(mask, w, h, 4, 2)
    yield mask

    for is_bg, ds in ((0, fg_downsample), (1, bg_downsample)):
        out = np.empty_like(img)
        ow, oh = C.c_int(), C.c_int()
        rc = lib().orc_page_layer(_p(img, u8p), _p(mask.view(np.uint8), u8p), w, h, c, is_bg,
                                  float(ds) if ds is not None else 0.0, _p(out, u8p),
                                  C.byref(ow), C.byref(oh))
        assert rc >= 0
        if rc == 1 and errors is not None:
            errors.add('too-small-to-downsample')
        n = ow.value * oh.value * c
        res = out.reshape(-1)[:n].reshape((oh.value, ow.value) if c == 1 else (oh.value, ow.value, 3)).copy()
        yield res


# ---- internetarchivepdf/grayconvert.py -----------------------------------------------------------
def level_arr(arr, minv=0, maxv=255):
    """grayconvert.py:24-31 on a uint8 array, in place: (arr - minv) / interval in float64, stored back into the uint8
    array (C truncation), values below minv -> 0, above maxv -> 255."""
    interval = (maxv / 255.) - (minv / 255.)
    arr_zero = arr < minv
    arr_max = arr > maxv
    with np.errstate(all='ignore'):
        arr[::] = ((arr[::] - minv) / interval)
    arr[arr_zero] = 0
    arr[arr_max] = 255
    return arr


def hsl_lightness_u8(rgb):
    """What grayconvert.py:63-66 makes of a uint8 RGB image: skimage.color.rgb2hsv (0.18.3 colorconv.py:190-265) needs
    only V = max and S = ptp / max of the image converted by img_as_float -- for uint8 input `np.multiply(image, 1 / 255,
    dtype=float64)`: a multiplication by the reciprocal, not a division (util/dtype.py:312-320; the computation type is
    the first of (float64, float32, float64) wider than the input) -- then l = V * (1 - S / 2) and uint8(l * 255) truncated."""
    arr = np.multiply(rgb, 1. / 255, dtype=np.float64)
    out_v = arr.max(-1)
    delta = np.ptp(arr, -1)
    with np.errstate(all='ignore'):
        out_s = delta / out_v
    out_s[delta == 0.] = 0.
    out_s[np.isnan(out_s)] = 0
    l = out_v * (1 - (out_s / 2))
    return np.array(l * 255, dtype=np.uint8)


def special_gray_convert(imd):
    """grayconvert.py:38-66, statement for statement (numpy's own min / max / mean / std on the channel views)."""
    components = ('r', 'g', 'b')
    d = {}
    for i, k in enumerate(components):
        for fun in ['min', 'max', 'mean', 'std']:
            d[k + '_' + fun] = getattr(np, fun)(imd[:, :, i]) / 255.
    bright_adjust = round(d['r_mean'] * d['g_mean'] * d['b_mean'] /
                          (d['b_max'] * (1 - d['r_std']) * (1 - d['g_std']) * (1 - d['b_std'])), 4)
    low_thres = min(int((196 * d['r_min'] + 14.5) / 1), 50)
    high_thres = {'r': min(int((35.66 * bright_adjust + 48.5) / 1), 95),
                  'g': min(int((39.22 * bright_adjust + 44.5) / 1), 95),
                  'b': min(int((45.16 * bright_adjust + 36.5) / 1), 95)}
    new_imd = np.copy(imd)
    for i, c in enumerate(components):
        new_imd[:, :, i] = level_arr(new_imd[:, :, i], minv=(low_thres * 255) / 100, maxv=(high_thres[c] * 255) / 100)
    return hsl_lightness_u8(new_imd)
