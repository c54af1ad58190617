"""Import the REAL reference hot path (internetarchivepdf/mrc.py + the Cython
modules built by oracle/build_ref.sh) inside THIS container only.

TEST INFRASTRUCTURE.  Used by tests/golden/make_golden.py to generate the
committed golden vectors and by tests that cross-check the C oracle when
/root/reference is present.  Never imported by the product path, never shipped:
/root/reference does not exist on the GPU box.

Needs an interpreter that has scikit-image + PyWavelets (here:
/opt/conda/bin/python3.9, skimage 0.18.3 / pywt 1.1.1 / scipy 1.7.1 /
numpy 1.26.4 / Pillow 8.4.0).

mrc.py:39-41 imports PyMuPDF only to call fitz.TOOLS.set_icc(True); PyMuPDF is
not installed here and is not on the hot path, so an in-memory module object
with that one no-op attribute satisfies the import (no file is written, nothing
of it is built or shipped).  internetarchivepdf/__init__.py pulls in the hOCR
parser (not installed, not on the path), so the package object is registered
empty and const.py / jpeg2000.py / mrc.py are executed from where they lie.
"""
import importlib.util
import os
import sys
import types
import warnings

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isdir(os.path.join(REF, 'internetarchivepdf'))


def load_cython():
    """Returns (sauvola, optimiser) reference extension modules for this interpreter."""
    d = os.path.join(HERE, '_ref', 'py%d%d' % sys.version_info[:2])
    if d not in sys.path:
        sys.path.insert(0, d)
    import sauvola, optimiser  # noqa: E401
    if not sauvola.__file__.startswith(d):
        raise ImportError('reference sauvola shadowed by %s' % sauvola.__file__)
    return sauvola, optimiser


def load_mrc():
    """Returns the reference internetarchivepdf.mrc module."""
    warnings.simplefilter('ignore')
    load_cython()
    if 'fitz' not in sys.modules:
        fitz = types.ModuleType('fitz')
        fitz.TOOLS = types.SimpleNamespace(set_icc=lambda v: None)
        sys.modules['fitz'] = fitz
    pkg = types.ModuleType('internetarchivepdf')
    pkg.__path__ = [os.path.join(REF, 'internetarchivepdf')]
    sys.modules['internetarchivepdf'] = pkg
    mods = {}
    for name in ('const', 'jpeg2000', 'mrc'):
        full = 'internetarchivepdf.' + name
        spec = importlib.util.spec_from_file_location(
            full, os.path.join(REF, 'internetarchivepdf', name + '.py'))
        m = importlib.util.module_from_spec(spec)
        sys.modules[full] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods['mrc']
