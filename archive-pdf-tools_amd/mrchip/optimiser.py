"""Drop-in for the reference's top-level `optimiser` module (cython/optimiser.pyx)."""
import numpy as np

from . import _lib


def _check2d(a, name, ndim):
    a = np.asarray(a)
    if a.ndim != ndim:
        raise ValueError('Buffer has wrong number of dimensions (expected %d, got %d)' % (ndim, a.ndim))
    return _lib.as_u8(a, name)


def _optimise(mask, img, width, height, n_size, channels, invert=0, ctx=None):
    m = _check2d(mask, 'mask', 2)
    i = _check2d(img, 'img', 2 if channels == 1 else 3)
    if m.shape != (height, width) or i.shape[:2] != (height, width) or (channels == 3 and i.shape[2] != 3):
        raise ValueError('mask/img shapes do not match width=%d height=%d' % (width, height))
    out = np.empty_like(i)
    ctx = ctx or _lib.default_context()
    _lib.check(_lib.load().mrchip_optimise(ctx.handle, _lib.ptr(m), _lib.ptr(i), _lib.ptr(out), width, height,
                                           channels, int(n_size), invert), 'mrchip_optimise')
    return out


def optimise_gray2(mask, img, width, height, n_size, ctx=None):
    """optimiser.optimise_gray2 (cython/optimiser.pyx:153): returns a new uint8[h,w] array."""
    return _optimise(mask, img, width, height, n_size, 1, ctx=ctx)


def optimise_rgb2(mask, img, width, height, n_size, ctx=None):
    """optimiser.optimise_rgb2 (cython/optimiser.pyx:280): returns a new uint8[h,w,3] array."""
    return _optimise(mask, img, width, height, n_size, 3, ctx=ctx)


# The slow spec versions (pyx:22-76, 83-146) compute the same result; mrc.py:36 imports the names.
optimise_gray = optimise_gray2
optimise_rgb = optimise_rgb2


def fast_mask_denoise(mask, width, height, mincnt, n_size, ctx=None):
    """optimiser.fast_mask_denoise (cython/optimiser.pyx:436): in place, returns `mask`.

    The mask is a bool / 0-1 uint8 array as everywhere in the reference (mrc.py:388)."""
    a = np.asarray(mask)
    if a.ndim != 2:
        raise ValueError('Buffer has wrong number of dimensions (expected 2, got %d)' % a.ndim)
    m = _lib.as_u8(a, 'mask')
    if m.shape != (height, width):
        raise ValueError('mask shape does not match width=%d height=%d' % (width, height))
    ctx = ctx or _lib.default_context()
    _lib.check(_lib.load().mrchip_mask_denoise(ctx.handle, _lib.ptr(m), width, height, int(mincnt), int(n_size)),
               'mrchip_mask_denoise')
    if not np.shares_memory(m, a):
        a[...] = m.view(a.dtype) if a.dtype == np.bool_ else m
    return mask
