import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'archive-pdf-tools_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
# the oracle is test infrastructure: only the tests put it on the path
ORACLE_DIR = os.path.join(ROOT, 'oracle')
if ORACLE_DIR not in sys.path:
    sys.path.insert(0, ORACLE_DIR)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _has_gpu():
    return os.path.exists('/dev/kfd') and os.access('/dev/kfd', os.R_OK | os.W_OK)


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
