#!/usr/bin/env python3
"""Headline benchmark: pages/s of the full MRC decomposition (mask + fg + bg) of
4000x3000 RGB pages with hOCR boxes, bg downsample 3 -- BASELINE.json configs[1] --
on N MI355X GPUs of one node, one process per GPU, pages sharded across ranks with no
data-path collective.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over this rank's batch of --pages pages whose pixels
are already resident in HBM (outputs stay in HBM; the PCIe-inclusive drop-in rate is
printed as `pcie_inclusive_pages_per_s`, it is never `value`).  Rank 0 prints ONE JSON line.

roofline: the kernel with the largest share of GPU time, measured with HIP events on the
stream each launch goes to (mrchip_prof_*), algorithmic bytes per SURVEY.md 8d.
cpu_baseline: the C restatement of the reference (oracle/, kind "port") timed on this
host's cores on a bounded sample of the same workload.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'archive-pdf-tools_amd'))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
W, H, C = 4000, 3000, 3
BG_DOWNSAMPLE = 3
DISTINCT = 4                # distinct synthetic pages, cycled through the batch


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_pages(n_distinct, rank):
    from mrchip import synth, mrc
    pages = []
    for i in range(n_distinct):
        seed = 202 + i + 1000 * rank          # seed 202 is the page the reference digested (digests.json)
        img, hocr = synth.synth_page(W, H, C, seed=seed, noise_sigma=6.0, line_div=60)
        boxes = mrc.hocr_boxes(hocr, W, H)
        pages.append((img, hocr, boxes))
    return pages


def cpu_baseline(sample_pages):
    """oracle (C port of the reference) on `sample_pages`: one worker thread per host core, each
    decomposing whole pages (the C calls release the GIL), plus a single-thread run for the
    per-core figure.  Bounded: ~12 s for the all-core run, ~6 s single-thread."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import threading
    import mrc_oracle as O
    O.lib()

    def decompose(i):
        img, hocr, _ = sample_pages[i % len(sample_pages)]
        for _ in O.create_mrc_hocr_components(img, hocr, dpi=None, bg_downsample=BG_DOWNSAMPLE, denoise_mask='fast'):
            pass

    def run(nthreads, seconds):
        done = [0] * nthreads
        t0 = time.time()

        def worker(k):
            i = k
            while time.time() - t0 < seconds:
                decompose(i)
                i += nthreads
                done[k] += 1
        th = [threading.Thread(target=worker, args=(k,)) for k in range(nthreads)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return sum(done), time.time() - t0

    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    nthr = max(1, min(ncpu, 32))                    # ~0.4 GB of numpy temporaries per in-flight page
    n1, dt1 = run(1, 6.0)
    nall, dtall = (n1, dt1) if nthr == 1 else run(nthr, 12.0)
    model = ''
    try:
        with open('/proc/cpuinfo') as f:
            for ln in f:
                if ln.startswith('model name'):
                    model = ln.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return {'value': round(nall / dtall, 4), 'unit': 'pages/s', 'cores': nthr, 'kind': 'port',
            'sample': '%d decompositions of the same 4000x3000 RGB pages through oracle/mrc_oracle.c (-O3), %d threads, %.1f s'
                      % (nall, nthr, dtall),
            'single_thread_value': round(n1 / dt1, 4), 'host_cpus': ncpu, 'cpu_model': model}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--pages', type=int, default=384, help='pages per GPU per step')
    ap.add_argument('--inflight', type=int, default=3, help='device batches in flight per GPU (one HIP stream each)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the untimed parity / PCIe-inclusive section (profiling runs)')
    a = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_
        # RCCL ("nccl") on the node; MRCHIP_DIST_BACKEND=gloo lets the launch path be exercised on a
        # single-GPU box (two ranks cannot share one GPU under RCCL)
        backend = os.environ.get('MRCHIP_DIST_BACKEND', 'nccl')
        ndev = max(1, torch.cuda.device_count())
        if backend == 'nccl':
            torch.cuda.set_device(local_rank % ndev)
            dist_.init_process_group('nccl', device_id=torch.device('cuda', local_rank % ndev))
        else:
            dist_.init_process_group(backend)
        dist = dist_

    from mrchip import _lib, mrc
    ctx = _lib.Context(local_rank % max(1, _lib.mrchip_visible_devices()))
    lib = _lib.load()
    info = ctx.info()

    # work queue: rank 0 owns the descriptor table and broadcasts it over RCCL (bytes, not pixels);
    # every rank then takes the pages i with i mod world == rank
    if dist is not None:
        from mrchip import dist as mdist
        desc = mdist.broadcast_descriptor(dist, {'w': W, 'h': H, 'pages_per_rank': a.pages, 'distinct': DISTINCT}
                                          if rank == 0 else None)
        assert (desc['w'], desc['h'], desc['pages_per_rank']) == (W, H, a.pages)

    host_pages = make_pages(DISTINCT, rank)
    window = 51                                    # dpi=None (bin/compress-pdf-images:66-70)
    # the rank's pages are split into `inflight` device batches, each on its own HIP stream, so that
    # the latency-bound row-sequential kernels of one batch overlap the streaming kernels of another
    nb = max(1, min(a.inflight, a.pages))
    sizes = [a.pages // nb + (1 if i < a.pages % nb else 0) for i in range(nb)]
    batches = []
    k = 0
    for sz in sizes:
        bt = mrc.Batch(ctx, sz, W, H, C)
        for i in range(sz):
            img, hocr, boxes = host_pages[k % DISTINCT]
            bt.upload(i, img)
            bt.set_boxes(i, boxes)
            k += 1
        batches.append(bt)
    batch = batches[0]
    ctx.sync()

    def step():
        for bt in batches:
            bt.mask_begin(window)                    # luma, hOCR-box thresholds, noise estimate
        for bt in batches:
            bt.mask_finish(bt.sigmas(), True)        # host: Gaussian tables; decisions, blur, Sauvola, denoise
            bt.layers(None, BG_DOWNSAMPLE)           # fg + bg optimise (one launch), bg thumbnail

    def barrier():
        ctx.sync()
        if dist is not None:
            if dist.get_backend() == 'nccl':
                import torch
                torch.cuda.synchronize()
            dist.barrier()

    for _ in range(a.warmup):
        step()
    barrier()
    ctx.prof_enable(True)
    ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    prof = ctx.prof_report()
    ctx.prof_enable(False)
    if dist is not None:
        from mrchip import dist as mdist
        dt = mdist.max_over_ranks(dist, dt)

    # untimed: parity evidence + PCIe-inclusive drop-in rate on rank 0
    extra = {}
    prof_iso = None
    if rank == 0 and not a.no_extras and nb > 1:
        # the timed region runs `nb` batches concurrently, so a launch's HIP-event duration there includes the
        # time it shares the chip with other streams; the same batches one at a time give the kernels' own rates
        ctx.prof_enable(True)
        ctx.prof_reset()
        for _ in range(2):
            for bt in batches:
                bt.mask_begin(window)
                bt.mask_finish(bt.sigmas(), True)
                bt.layers(None, BG_DOWNSAMPLE)
                ctx.sync()
        prof_iso = ctx.prof_report()
        ctx.prof_enable(False)
    if rank == 0 and not a.no_extras:
        mask = batch.download_mask(0)
        try:
            with open(os.path.join(ROOT, 'tests', 'golden', 'digests.json')) as f:
                dg = json.load(f)['c2_dpiNone']
            extra['parity'] = 'mask sha256 %s the reference digest' % ('==' if sha(mask) == dg['mask'] else '!=')
        except Exception as e:     # pragma: no cover
            extra['parity'] = 'unchecked: %s' % e
        img, hocr, _ = host_pages[0]
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            for _ in mrc.create_mrc_hocr_components(img, hocr, dpi=None, bg_downsample=BG_DOWNSAMPLE,
                                                    denoise_mask='fast', ctx=ctx):
                pass
        extra['pcie_inclusive_pages_per_s'] = round(reps / (time.perf_counter() - t1), 2)
        # layer hand-off: D2H rate of one full-resolution layer, pageable vs pinned destination
        try:
            pg = np.empty((H, W, C), np.uint8)
            pin = ctx.pinned_empty((H, W, C))
            for name, dstarr in (('pageable', pg), ('pinned', pin)):
                batch.download_layer(0, 0, (W, H), out=dstarr)
                t2 = time.perf_counter()
                for _ in range(3):
                    batch.download_layer(0, 0, (W, H), out=dstarr)
                extra['d2h_layer_GBps_' + name] = round(3 * pg.nbytes / (time.perf_counter() - t2) / 1e9, 1)
        except Exception as e:     # pragma: no cover
            extra['d2h_layer_GBps_error'] = str(e)
        # measured ceiling to read the roofline fractions against: D2D copy, read + write bytes
        extra['hbm_copy_GBps_measured'] = round(ctx.hbm_copy_bandwidth(1 << 30, 10), 1)

    total_pages = a.pages * a.steps * world
    value = total_pages / dt
    def roofline_of(name, prof=prof):
        r = prof[name]
        ms = r['ms'] / r['launches']
        alg = r['alg_bytes'] / r['launches']
        achieved = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {'bound': 'hbm', 'kernel': name, 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': pmc_traffic(name, alg),
                'avg_launch_ms': round(ms, 4), 'launches': r['launches'], 'alg_bytes_per_launch': alg}

    def pmc_traffic(name, alg_per_launch):
        """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_pmc_summary.json:
        2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction), scaled to this run's launch size."""
        sym = {'optimise_rgb': 'optimise_packed_kernel<3', 'optimise_gray': 'optimise_packed_kernel<1',
               'sauvola': 'sauvola_kernel', 'sauvola_boxes': 'sauvola_kernel'}.get(name, name)
        try:
            import glob
            f = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_summary.json')))[-1]
            d = json.load(open(f))
            for k, v in d['kernels'].items():
                if sym in k and 'alg_bytes_per_launch' in d.get('scale', {}).get(name, {}):
                    return round(v['hbm_bytes_per_launch'] * alg_per_launch / d['scale'][name]['alg_bytes_per_launch'])
        except Exception:
            pass
        return None

    # dominant kernel by GPU time
    dom = max(prof.items(), key=lambda kv: kv[1]['ms']) if prof else None
    roof = roofline_of(dom[0]) if dom else None
    sauvola_roof = roofline_of('sauvola') if 'sauvola' in prof else None
    if prof_iso and dom:
        iso = roofline_of(dom[0], prof_iso)
        roof['isolated'] = {'achieved': iso['achieved'], 'frac': iso['frac'], 'avg_launch_ms': iso['avg_launch_ms'],
                            'note': 'same batches run one at a time after the timed region (no other stream on the chip)'}
        roof['note'] = ('%d batches share the chip in the timed region: avg_launch_ms / achieved / frac are per-launch '
                        'figures under that concurrency' % nb)
        if sauvola_roof and 'sauvola' in prof_iso:
            si = roofline_of('sauvola', prof_iso)
            sauvola_roof['isolated'] = {'achieved': si['achieved'], 'frac': si['frac'], 'avg_launch_ms': si['avg_launch_ms']}
    kernels = {k: {'ms_per_launch': round(v['ms'] / v['launches'], 4), 'launches': v['launches'],
                   'alg_GBps': round(v['alg_bytes'] / max(v['ms'], 1e-9) / 1e6, 1)} for k, v in sorted(prof.items())}
    if rank == 0:
        cpu = None if a.no_cpu_baseline else cpu_baseline(host_pages)
        line = {
            'metric': 'pages/sec MRC decompose (4000x3000 RGB)', 'value': round(value, 2), 'unit': 'pages/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
            'config': {'workload': 'configs[1]: 4000x3000 RGB page + hOCR line boxes, dpi=None (window 51), '
                                   'bg_downsample=3, denoise fast, full create_mrc_hocr_components',
                       'pages_per_gpu_per_step': a.pages, 'batches_in_flight': nb, 'distinct_pages': DISTINCT,
                       'hocr_boxes_per_page': int(len(host_pages[0][2])), 'sharding': 'pages round-robin over ranks'},
            'roofline': roof, 'cpu_baseline': cpu,
            'sauvola_roofline': sauvola_roof,      # BASELINE.json also names "Sauvola HBM GB/s"
            'kernels': kernels, 'device': info['name'].strip(),
        }
        line.update(extra)
        print(json.dumps(line))
    for bt in batches:
        bt.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
