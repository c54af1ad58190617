#!/usr/bin/env python3
"""Kernel micro-benchmarks on one GPU through the C ABI (HIP-event timing from
mrchip_prof_*).  Usage: python tools/kbench.py [sauvola|all] [--w W --h H --reps N]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'archive-pdf-tools_amd'))
import numpy as np  # noqa: E402
from mrchip import _lib, sauvola  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('what', nargs='?', default='all')
    ap.add_argument('--w', type=int, default=3300)
    ap.add_argument('--h', type=int, default=4600)
    ap.add_argument('--window', type=int, default=51)
    ap.add_argument('--reps', type=int, default=10)
    a = ap.parse_args()
    ctx = _lib.default_context()
    print(json.dumps(ctx.info()))
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (a.h, a.w)).astype(np.uint8)
    out = np.empty(a.h * a.w, np.uint8)
    sauvola.binarise_sauvola(img.reshape(-1), out, a.w, a.h, a.window, a.window, 0.34, 128.0)
    ctx.prof_enable(True)
    ctx.prof_reset()
    t0 = time.time()
    for _ in range(a.reps):
        sauvola.binarise_sauvola(img.reshape(-1), out, a.w, a.h, a.window, a.window, 0.34, 128.0)
    wall = (time.time() - t0) / a.reps
    rep = ctx.prof_report()
    for name, r in rep.items():
        ms = r['ms'] / r['launches']
        gbs = r['alg_bytes'] / r['launches'] / (ms * 1e-3) / 1e9
        print('%-16s %8.3f ms/launch  %8.1f GB/s algorithmic  (%.1f%% of 8 TB/s)' % (name, ms, gbs, gbs / 80.0))
    print('wall per call incl. PCIe staging: %.2f ms' % (wall * 1e3))


if __name__ == '__main__':
    main()
