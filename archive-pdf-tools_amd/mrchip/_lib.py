"""ctypes binding of libmrchip.so (include/mrchip.h)."""
import ctypes as C
import os
import sys
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), 'lib', 'libmrchip.so')
if os.environ.get('MRCHIP_LIB'):        # A/B runs of two builds in one process tree (tools/)
    LIB_PATH = os.environ['MRCHIP_LIB']

MAX_TAPS = 128        # MRCHIP_MAX_TAPS
u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
f64p = C.POINTER(C.c_double)
f32p = C.POINTER(C.c_float)
intp = C.POINTER(C.c_int)
vp = C.c_void_p


class MrchipError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()

# name -> (restype, argtypes); every symbol include/mrchip.h declares
SIGNATURES = {
    'mrchip_abi_version': (C.c_int, []),
    'mrchip_device_count': (C.c_int, []),
    'mrchip_create': (vp, [C.c_int]),
    'mrchip_destroy': (None, [vp]),
    'mrchip_last_error': (C.c_char_p, []),
    'mrchip_sync': (C.c_int, [vp]),
    'mrchip_device_info': (C.c_int, [vp, C.c_char_p, C.c_int, intp, C.POINTER(C.c_size_t)]),
    'mrchip_device_memory': (C.c_int, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    'mrchip_sauvola_u8': (C.c_int, [vp, u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]),
    'mrchip_mask_denoise': (C.c_int, [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_int]),
    'mrchip_optimise': (C.c_int, [vp, u8p, u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'mrchip_luma601': (C.c_int, [vp, u8p, u8p, C.c_int, C.c_int]),
    'mrchip_estimate_sigma': (C.c_int, [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_int, f64p]),
    'mrchip_estimate_noise_u8': (C.c_int, [vp, u8p, C.c_int, C.c_int, f64p]),
    'mrchip_gaussian_u8': (C.c_int, [vp, u8p, u8p, C.c_int, C.c_int, C.c_double, f64p, C.c_int]),
    'mrchip_thumbnail_size': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, intp, intp]),
    'mrchip_thumbnail': (C.c_int, [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p]),
    'mrchip_thumbnail_ex': (C.c_int, [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, u8p]),
    'mrchip_window_for_dpi': (C.c_int, [C.c_int, C.c_double]),
    'mrchip_hocr_mask': (C.c_int, [vp, u8p, u8p, C.c_int, C.c_int, i32p, C.c_int, C.c_int, i32p]),
    'mrchip_page_create': (vp, [vp, C.c_int, C.c_int, C.c_int]),
    'mrchip_page_destroy': (None, [vp]),
    'mrchip_page_upload': (C.c_int, [vp, u8p]),
    'mrchip_page_mask_begin': (C.c_int, [vp, i32p, C.c_int, C.c_int]),
    'mrchip_page_sigma': (C.c_int, [vp, f64p]),
    'mrchip_page_mask_finish': (C.c_int, [vp, f64p, C.c_int, C.c_int]),
    'mrchip_page_download_mask': (C.c_int, [vp, u8p]),
    'mrchip_page_download_mask_packed': (C.c_int, [vp, u8p]),
    'mrchip_page_layer': (C.c_int, [vp, C.c_int, C.c_double, intp, intp, intp]),
    'mrchip_page_layers': (C.c_int, [vp, C.c_double, C.c_double, intp, intp, intp, intp, intp]),
    'mrchip_device_numa_node': (C.c_int, [vp]),
    'mrchip_page_download_layer': (C.c_int, [vp, C.c_int, u8p]),
    'mrchip_page_sync': (C.c_int, [vp]),
    'mrchip_page_box_decisions': (C.c_int, [vp, i32p, C.c_int]),
    'mrchip_page_device_ptrs': (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_size_t),
                                          C.POINTER(vp), C.POINTER(vp)]),
    'mrchip_batch_create': (vp, [vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    'mrchip_batch_destroy': (None, [vp]),
    'mrchip_batch_upload': (C.c_int, [vp, C.c_int, u8p]),
    'mrchip_batch_upload_gray': (C.c_int, [vp, C.c_int, u8p]),
    'mrchip_batch_upload_mask': (C.c_int, [vp, C.c_int, u8p]),
    'mrchip_page_upload_gray': (C.c_int, [vp, u8p]),
    'mrchip_page_upload_mask': (C.c_int, [vp, u8p]),
    'mrchip_batch_set_count': (C.c_int, [vp, C.c_int]),
    'mrchip_batch_set_boxes': (C.c_int, [vp, C.c_int, i32p, C.c_int]),
    'mrchip_batch_mask_begin': (C.c_int, [vp, C.c_int]),
    'mrchip_batch_threshold': (C.c_int, [vp, C.c_int, C.c_double]),
    'mrchip_batch_sigmas': (C.c_int, [vp, f64p]),
    'mrchip_batch_mask_finish': (C.c_int, [vp, f64p, intp, C.c_int]),
    'mrchip_batch_download_mask': (C.c_int, [vp, C.c_int, u8p]),
    'mrchip_batch_download_mask_packed': (C.c_int, [vp, C.c_int, u8p]),
    'mrchip_batch_layers': (C.c_int, [vp, C.c_int, C.c_double, C.c_double, intp, intp, intp, intp, intp]),
    'mrchip_batch_download_layer': (C.c_int, [vp, C.c_int, C.c_int, u8p]),
    'mrchip_batch_download_layer_async': (C.c_int, [vp, C.c_int, C.c_int, u8p]),
    'mrchip_batch_download_mask_async': (C.c_int, [vp, C.c_int, u8p]),
    'mrchip_batch_download_mask_packed_async': (C.c_int, [vp, C.c_int, u8p]),
    'mrchip_batch_done': (C.c_int, [vp]),
    'mrchip_host_alloc': (C.c_void_p, [vp, C.c_size_t]),
    'mrchip_host_free': (None, [vp, C.c_void_p]),
    'mrchip_batch_sync': (C.c_int, [vp]),
    'mrchip_batch_box_decisions': (C.c_int, [vp, C.c_int, i32p, C.c_int]),
    'mrchip_batch_device_ptrs': (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_size_t),
                                           C.POINTER(vp), C.POINTER(vp)]),
    'mrchip_comm_unique_id': (C.c_int, [u8p]),
    'mrchip_comm_init': (vp, [vp, C.c_int, C.c_int, u8p]),
    'mrchip_comm_destroy': (None, [vp]),
    'mrchip_comm_bcast': (C.c_int, [vp, C.c_void_p, C.c_size_t, C.c_int]),
    'mrchip_comm_allgather': (C.c_int, [vp, C.c_void_p, C.c_size_t, C.c_void_p]),
    'mrchip_comm_allreduce_f64': (C.c_int, [vp, f64p, C.c_int, C.c_int]),
    'mrchip_estimate_sigma_f32': (C.c_int, [vp, f32p, C.c_int, C.c_int, C.c_int, f64p]),
    'mrchip_estimate_noise_f32': (C.c_int, [vp, f32p, C.c_int, C.c_int, f64p]),
    'mrchip_gaussian_f32': (C.c_int, [vp, f32p, u8p, C.c_int, C.c_int, C.c_double, f64p, C.c_int]),
    'mrchip_special_gray_begin': (C.c_int, [vp, u8p, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]),
    'mrchip_special_gray_finish': (C.c_int, [vp, u8p, u8p, u8p]),
    'mrchip_canary_check': (C.c_int, [vp, C.POINTER(C.c_longlong)]),
    'mrchip_canary_selftest': (C.c_int, [vp, C.POINTER(C.c_longlong)]),
    'mrchip_prof_enable': (C.c_int, [vp, C.c_int]),
    'mrchip_prof_reset': (C.c_int, [vp]),
    'mrchip_prof_count': (C.c_int, [vp]),
    'mrchip_prof_get': (C.c_int, [vp, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_longlong), f64p, f64p]),
    'mrchip_hbm_copy_bandwidth': (C.c_int, [vp, C.c_size_t, C.c_int, f64p]),
    'mrchip_selftest_sauvola_quotients': (C.c_int, [vp, C.POINTER(C.c_longlong)]),
    'mrchip_selftest_sauvola_table': (C.c_int, [vp, C.c_double, C.c_double, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                                C.POINTER(C.c_int)]),
    'mrchip_selftest_optimise_quotients': (C.c_int, [vp, C.POINTER(C.c_longlong)]),
    'mrchip_selftest_gauss_fast': (C.c_int, [vp, f64p, C.c_int, C.POINTER(C.c_longlong), f64p]),
}


def load():
    """Load libmrchip.so; raises MrchipError when it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise MrchipError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"`'
                                  ' (make -C archive-pdf-tools_amd/csrc). There is no CPU fallback.' % LIB_PATH)
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name, None)
                if fn is None and os.environ.get('MRCHIP_LIB'):
                    continue              # an older build named for a same-box A/B run (tools/ab.sh)
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def last_error():
    return load().mrchip_last_error().decode('utf-8', 'replace')


def check(rc, what=''):
    if rc != 0:
        raise MrchipError('%s failed (%d): %s' % (what or 'libmrchip call', rc, last_error()))


class Context:
    """One HIP context (device + streams + scratch) -- mrchip_create/mrchip_destroy."""

    def __init__(self, device=0):
        lib = load()
        self._h = lib.mrchip_create(device)
        if not self._h:
            raise MrchipError('mrchip_create(%d): %s' % (device, last_error()))
        self.device = device

    @property
    def handle(self):
        if not self._h:
            raise MrchipError('context destroyed')
        return self._h

    def close(self):
        if self._h:
            load().mrchip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            if not sys.is_finalizing():           # no HIP calls from interpreter shutdown: the runtime tears itself down
                self.close()
        except Exception:
            pass

    def sync(self):
        check(load().mrchip_sync(self.handle), 'mrchip_sync')

    def canary_check(self):
        """Guard bands of the device allocator (MRCHIP_CANARY=<KiB>, a debugging switch): waits for the device, verifies
        every block, returns the guard bytes found overwritten since the context was created (0 with the switch off)."""
        n = C.c_longlong()
        check(load().mrchip_canary_check(self.handle, C.byref(n)), 'mrchip_canary_check')
        return int(n.value)

    def canary_selftest(self):
        """-> guard bytes detected after a deliberate one-byte write on each side of a scratch block (2 on, 0 off)."""
        n = C.c_longlong()
        check(load().mrchip_canary_selftest(self.handle, C.byref(n)), 'mrchip_canary_selftest')
        return int(n.value)

    def numa_node(self):
        """NUMA node of the socket the GPU is attached to, -1 if unknown."""
        return int(load().mrchip_device_numa_node(self.handle))

    def info(self):
        name = C.create_string_buffer(128)
        cus = C.c_int()
        hbm = C.c_size_t()
        check(load().mrchip_device_info(self.handle, name, 128, C.byref(cus), C.byref(hbm)))
        return {'name': name.value.decode(), 'cus': cus.value, 'hbm_bytes': hbm.value}

    def memory(self):
        """(free, total) bytes of device memory right now."""
        fr, tot = C.c_size_t(), C.c_size_t()
        check(load().mrchip_device_memory(self.handle, C.byref(fr), C.byref(tot)))
        return fr.value, tot.value

    def pinned_empty(self, shape, dtype=np.uint8):
        """numpy array over page-locked host memory (freed with the array): the destination of the
        asynchronous downloads (Batch.download_layer(..., out=..., wait=False))."""
        shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        lib = load()
        p = lib.mrchip_host_alloc(self.handle, max(nbytes, 1))
        if not p:
            raise MrchipError('mrchip_host_alloc(%d) failed: %s' % (nbytes, last_error()))
        buf = (C.c_ubyte * max(nbytes, 1)).from_address(p)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        _PINNED[p] = max(nbytes, 1)

        def _release(addr=p):
            _PINNED.pop(addr, None)
            if not sys.is_finalizing():           # at interpreter exit the HIP runtime may already be shutting down:
                lib.mrchip_host_free(None, addr)  # leave the pages to the OS (hipHostFree needs no particular device)
        weakref.finalize(buf, _release)
        return arr

    def hbm_copy_bandwidth(self, nbytes=1 << 30, reps=10):
        """Measured device-to-device copy rate in GB/s (bytes read + written)."""
        g = C.c_double()
        check(load().mrchip_hbm_copy_bandwidth(self.handle, nbytes, reps, C.byref(g)))
        return g.value

    def prof_enable(self, on=True):
        check(load().mrchip_prof_enable(self.handle, 1 if on else 0))
        self.prof_on = bool(on)

    def prof_reset(self):
        check(load().mrchip_prof_reset(self.handle))

    def prof_report(self):
        lib = load()
        n = lib.mrchip_prof_count(self.handle)
        if n < 0:
            check(n, 'mrchip_prof_count')
        out = {}
        for i in range(n):
            name = C.create_string_buffer(64)
            launches = C.c_longlong()
            ms = C.c_double()
            ab = C.c_double()
            check(lib.mrchip_prof_get(self.handle, i, name, 64, C.byref(launches), C.byref(ms), C.byref(ab)))
            out[name.value.decode()] = {'launches': launches.value, 'ms': ms.value, 'alg_bytes': ab.value}
        return out


_PINNED = {}        # address -> bytes of the live pinned_empty allocations


def is_pinned(arr):
    """True when `arr`'s bytes lie inside a Context.pinned_empty allocation: copies from / to it are true
    asynchronous DMAs, anything else makes the HIP runtime stage the copy and block the calling thread."""
    a = arr.ctypes.data
    n = arr.nbytes
    for base, size in _PINNED.items():
        if base <= a and a + n <= base + size:
            return True
    return False


_tls = threading.local()


def default_context():
    """Per-thread default context on MRCHIP_DEVICE (default 0)."""
    ctx = getattr(_tls, 'ctx', None)
    if ctx is None:
        ctx = Context(int(os.environ.get('MRCHIP_DEVICE', '0') or 0))
        _tls.ctx = ctx
    return ctx


def mrchip_visible_devices():
    return load().mrchip_device_count()


def as_u8(a, name='array'):
    """C-contiguous uint8 view of a uint8/bool array (copy only if needed)."""
    a = np.asarray(a)
    if a.dtype == np.bool_:
        a = a.view(np.uint8)
    if a.dtype != np.uint8:
        raise ValueError("Buffer dtype mismatch, expected 'UINT8DTYPE_t' but got '%s' (%s)" % (a.dtype, name))
    return np.ascontiguousarray(a)


def ptr(a, t=u8p):
    return a.ctypes.data_as(t)
