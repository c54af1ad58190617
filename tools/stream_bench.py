#!/usr/bin/env python3
"""PCIe-inclusive throughput of mrc.decompose_stream on one GPU for a few (batch_pages, slots) settings.
    python tools/stream_bench.py [--pages 512] [--combos 32x3,32x4,16x4,64x3]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'archive-pdf-tools_amd'))
from mrchip import _lib, mrc, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pages', type=int, default=512)
    ap.add_argument('--combos', default='32x3,32x4,16x4,64x3,16x6')
    ap.add_argument('--mask', default='packed')
    ap.add_argument('--node', type=int, default=-1, help='bind this process (and so its page-locked memory) to the CPUs of one NUMA node')
    a = ap.parse_args()
    if a.node >= 0:
        cpus = set()
        for part in open('/sys/devices/system/node/node%d/cpulist' % a.node).read().strip().split(','):
            lo, _, hi = part.partition('-')
            cpus.update(range(int(lo), int(hi or lo) + 1))
        os.sched_setaffinity(0, cpus)
    ctx = _lib.default_context()
    made = synth.synth_pages([dict(w=4000, h=3000, channels=3, seed=202 + i, noise_sigma=6.0, line_div=60) for i in range(8)])
    for combo in a.combos.split(','):
        bp, sl = (int(x) for x in combo.split('x'))
        pool = mrc.StreamPool(ctx)
        ctx.prof_enable(True)
        for rep in range(2):
            ctx.prof_reset()
            st = {}
            gen = mrc.decompose_stream(((made[i % 8][0], made[i % 8][1]) for i in range(a.pages)), bg_downsample=3,
                                       batch_pages=bp, slots=sl, mask_format=a.mask, pool=pool, stats=st)
            t0 = time.perf_counter()
            n = sum(1 for _ in gen)
            ctx.sync()
            dt = time.perf_counter() - t0
        print(json.dumps({'batch_pages': bp, 'slots': sl, 'pages_per_s': round(n / dt, 1), 'ms_per_page': round(dt / n * 1e3, 3),
                          'optimise_ms': {k: round(v['ms'] / v['launches'], 3) for k, v in ctx.prof_report().items() if 'optimise' in k},
                          'phases': {k: round(v, 3) for k, v in st.items()}}))
        pool.close()


if __name__ == '__main__':
    main()
