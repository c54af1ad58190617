"""CPU suite: libmrchip.so loads and exports every symbol include/mrchip.h declares
(no compute calls without a GPU), and the product refuses to run without one."""
import os
import re

import pytest

from mrchip import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, 'include', 'mrchip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mrchip_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 30
    for name in syms:
        assert hasattr(lib, name), 'libmrchip.so does not export %s' % name
    # and the binding knows every one of them
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.mrchip_abi_version() == 1


def test_host_logic_without_gpu():
    lib = _lib.load()
    assert lib.mrchip_window_for_dpi(0, 0.0) == 51            # mrc.py:68
    assert lib.mrchip_window_for_dpi(1, 124.0) == 31          # int(31.0)
    assert lib.mrchip_window_for_dpi(1, 400.0) == 101         # 100 -> odd
    assert lib.mrchip_window_for_dpi(1, 364.0) == 91


@pytest.mark.skipif(os.path.exists('/dev/kfd'), reason='GPU present')
def test_fails_loudly_without_gpu():
    with pytest.raises(_lib.MrchipError):
        _lib.Context(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'archive-pdf-tools_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f), errors='replace').read()
                assert 'mrc_oracle' not in text and 'oracle/' not in text.replace('# oracle/', ''), \
                    '%s references the oracle' % os.path.join(dirpath, f)


def test_thumbnail_size_rule_matches_oracle_and_pillow():
    """mrchip_thumbnail_size is host logic (Pillow's Image.thumbnail size rule): check it on the CPU
    against the oracle and against Pillow itself."""
    import ctypes as C
    import mrc_oracle as O
    from PIL import Image
    lib = _lib.load()
    for (w, h) in [(800, 600), (4000, 3000), (3300, 4600), (8000, 6000), (1333, 999), (17, 900), (900, 17), (5, 5), (1, 1)]:
        for f in (1, 2, 3, 4, 5, 7, 10, 33):
            rw, rh = int(w / f), int(h / f)
            if rw <= 0 or rh <= 0:
                continue
            ow, oh = C.c_int(), C.c_int()
            changed = lib.mrchip_thumbnail_size(w, h, rw, rh, C.byref(ow), C.byref(oh))
            assert (ow.value, oh.value, bool(changed)) == O.thumbnail_size(w, h, rw, rh), (w, h, f)
            im = Image.new('L', (w, h))
            im.thumbnail((rw, rh))
            assert im.size == (ow.value, oh.value), (w, h, f, im.size)
    assert lib.mrchip_thumbnail_size(800, 600, 266, 200, C.byref(ow), C.byref(oh)) == 1 and (ow.value, oh.value) == (266, 200)
