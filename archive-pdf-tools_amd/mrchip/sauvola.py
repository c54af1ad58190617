"""Drop-in for the reference's top-level `sauvola` module (cython/sauvola.pyx)."""
import numpy as np

from . import _lib


def binarise_sauvola(in_arr, out_arr, width, height, window_width, window_height, k, R, ctx=None):
    """sauvola.binarise_sauvola(in_arr, out_arr, width, height, window_width,
    window_height, k, R) -- cython/sauvola.pyx:29.

    in_arr: flat uint8[width*height]; out_arr: preallocated flat 1-byte array
    (uint8 or bool), filled in place with 1 = bright/background.  Returns 0."""
    in_arr = np.asarray(in_arr)
    if in_arr.ndim != 1 or np.asarray(out_arr).ndim != 1:
        raise ValueError('Buffer has wrong number of dimensions (expected 1, got %d)' % in_arr.ndim)
    src = _lib.as_u8(in_arr, 'in_arr')
    dst = np.asarray(out_arr)
    if dst.dtype not in (np.uint8, np.bool_):
        raise ValueError("Buffer dtype mismatch, expected 'UINT8DTYPE_t' but got '%s'" % dst.dtype)
    if src.size < width * height or dst.size < width * height:
        raise ValueError('arrays smaller than width*height')
    direct = dst.flags.c_contiguous and dst.flags.writeable
    tmp = dst.view(np.uint8) if direct else np.empty(width * height, dtype=np.uint8)
    ctx = ctx or _lib.default_context()
    _lib.check(_lib.load().mrchip_sauvola_u8(ctx.handle, _lib.ptr(src), _lib.ptr(tmp), width, height,
                                             window_width, window_height, float(k), float(R), 0),
               'mrchip_sauvola_u8')
    if not direct:
        dst[:width * height] = tmp.view(dst.dtype)
    return 0
