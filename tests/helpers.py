"""Shared helpers for the parity tests."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name))
    meta = json.loads(str(z['meta']))
    return z, meta


def load_digests():
    with open(os.path.join(GOLDEN, 'digests.json')) as f:
        return json.load(f)


def unpack(bits, w):
    return np.unpackbits(bits, axis=1)[:, :w].astype(np.bool_)


def kernel_cases(kind):
    z, meta = load_npz('kernels.npz')
    return z, [m for m in meta if m[0] == kind]


def thirdparty_cases(kind):
    z, meta = load_npz('thirdparty.npz')
    return z, [m for m in meta if m[0] == kind]


def load_configs():
    with open(os.path.join(GOLDEN, 'configs.json')) as f:
        return json.load(f)


def sha_many(arrays, threads=16):
    """sha256 of many arrays on a thread pool (hashlib releases the GIL on large buffers)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max(1, min(threads, len(arrays)))) as ex:
        return list(ex.map(sha, arrays))
