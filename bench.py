#!/usr/bin/env python3
"""Headline benchmark: pages/s of the full MRC decomposition (mask + fg + bg) of 4000x3000 RGB pages with hOCR
boxes, bg downsample 3 -- BASELINE.json configs[1] -- on N MI355X GPUs of one node, one process per GPU, pages
sharded across ranks (page i -> rank i mod N) with no data-path collective.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W          (any launcher that sets RANK / LOCAL_RANK / WORLD_SIZE)

A "step" = one pass of the hot path over this rank's --pages pages whose pixels are already resident in HBM;
outputs stay in HBM.  Rank 0 prints ONE JSON line.  Beside `value` the line carries, all measured in this run:

  roofline        the kernel with the largest share of GPU time (ranked with the chip to itself, i.e. in the isolated
                  pass): algorithmic bytes / launch time from HIP events on the stream each launch goes to
                  (mrchip_prof_*), in the timed region; `isolated`: the same batches one at a time
  cpu_baseline    the C restatement of the reference (oracle/, kind "port") on this host's cores (N = 1 only)
  e2e             the PCIe-inclusive rate of the streaming page pipeline (mrc.decompose_stream: host arrays in,
                  packed mask + fg + bg thumbnails out, upload / compute / download of three rotating batches
                  overlapped), with the measured link rates -- never `value`
  parity          SHA-256 of the outputs of the pages this run decomposed against the digests the REFERENCE produced
                  for the same synthetic pages (tests/golden/configs.json); at N > 1 the 512-page stack of
                  configs[3] is sharded, every rank's records are gathered on rank 0 and all 512 are checked

--config c3 | c3gray | c5 run the other BASELINE.json configurations with the same JSON shape (c3gray: the
Sauvola-only batch whose HBM GB/s is BASELINE's second metric).  The control plane between ranks is RCCL through
libmrchip (mrchip_comm_*): no PyTorch anywhere in this process.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'archive-pdf-tools_amd'))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
E2E_BATCH, E2E_SLOTS = 8, 4        # decompose_stream's defaults: the setting that keeps both directions of the link busiest
PCIE_SPEC_GBS = 63.0        # PCIe Gen5 x16 per direction (MI355X_MICROARCH.md)
REFERENCE_PAGES_PER_S_PER_CORE = 0.33   # the reference itself on configs[1], BASELINE.md 2 (survey container, 1 core)

CONFIGS = {
    'c2': dict(label='configs[1]: 4000x3000 RGB page + hOCR line boxes, dpi=None (window 51), bg_downsample=3, '
                     'denoise fast, full create_mrc_hocr_components',
               w=4000, h=3000, c=3, dpi=None, bg=3, fg=None, pages=384, inflight=3, distinct=16, line_div=60,
               seeds=list(range(202, 266)), digests='c2_pages', metric='pages/sec MRC decompose (4000x3000 RGB)'),
    'c3': dict(label='configs[2] (RGB): batch of 64 pages 3300x4600 RGB + hOCR, window 51, bg_downsample=3, full decomposition',
               w=3300, h=4600, c=3, dpi=None, bg=3, fg=None, pages=64, inflight=1, distinct=4, line_div=60,
               seeds=[303, 304, 305, 306], digests='c3_rgb_pages', metric='pages/sec MRC decompose (3300x4600 RGB)'),
    'c3gray': dict(label='configs[2] (gray): batch of 64 pages 3300x4600 gray, Sauvola window 51 k=0.34 only (threshold_image)',
                   w=3300, h=4600, c=1, dpi=None, bg=None, fg=None, pages=64, inflight=1, distinct=8, line_div=60,
                   seeds=list(range(303, 311)), digests='c3_gray_pages', metric='Sauvola HBM GB/s (64 x 3300x4600 gray)'),
    'c5': dict(label='configs[4]: 8000x6000 RGB + hOCR, dpi=364 (window 91), fg and bg downsample 4',
               w=8000, h=6000, c=3, dpi=364, bg=4, fg=4, pages=128, inflight=2, distinct=2, line_div=60,
               seeds=[505, 506], digests=None, metric='pages/sec MRC decompose (8000x6000 RGB)'),
}
STACK_PAGES = 512            # BASELINE.json configs[3]: the 512-page stack, sharded page i -> rank i mod N


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sha_many(arrays, threads=32):
    from concurrent.futures import ThreadPoolExecutor
    if not arrays:
        return []
    with ThreadPoolExecutor(max(1, min(threads, len(arrays)))) as ex:
        return list(ex.map(sha, arrays))


def ncpus():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline: runs BEFORE anything touches the GPU (worker processes are forked)
def _cpu_worker(args):
    k, seconds, cfg, as_limit = args
    if as_limit:
        # a worker that outgrows its allowance gets MemoryError and the baseline is reported as failed; the box survives
        import resource
        try:
            with open('/proc/self/statm') as f:
                now = int(f.read().split()[0]) * os.sysconf('SC_PAGE_SIZE')
            resource.setrlimit(resource.RLIMIT_AS, (now + int(as_limit), now + int(as_limit)))
        except Exception:
            pass
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import mrc_oracle as O
    from mrchip import synth
    O.lib()
    img, hocr = synth.synth_page(cfg['w'], cfg['h'], cfg['c'], seed=cfg['seeds'][k % len(cfg['seeds'])], noise_sigma=6.0,
                                 line_div=cfg['line_div'])
    done = 0
    t0 = time.time()
    while time.time() - t0 < seconds:
        if cfg['bg'] is None and cfg['fg'] is None and cfg['c'] == 1:
            O.threshold_image(img, cfg['dpi'])
        else:
            for _ in O.create_mrc_hocr_components(img, hocr, dpi=cfg['dpi'], bg_downsample=cfg['bg'],
                                                  fg_downsample=cfg['fg'], denoise_mask='fast'):
                pass
        done += 1
    peak = 0
    try:
        import resource
        peak = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss * 1024
    except Exception:
        pass
    return done, time.time() - t0, peak


HOST_GB_DEFAULT = 64.0      # MRCHIP_BENCH_HOST_GB: what ALL host allocations of one bench run on a node may add up to


def cgroup_memory():
    """(limit, used) bytes of this container's memory cgroup (v2 memory.max / v1 limit_in_bytes), (None, None) when there
    is no limit to read ("max", no file): then the ceiling alone bounds the run."""
    for lim, use in (('/sys/fs/cgroup/memory.max', '/sys/fs/cgroup/memory.current'),
                     ('/sys/fs/cgroup/memory/memory.limit_in_bytes', '/sys/fs/cgroup/memory/memory.usage_in_bytes')):
        try:
            with open(lim) as f:
                v = f.read().strip()
            if v and v != 'max' and int(v) < (1 << 60):
                with open(use) as f:
                    return int(v), int(f.read().strip())
        except Exception:
            continue
    return None, None


def host_memory_budget():
    """Bytes THIS RANK's bench legs (CPU-baseline workers, fresh page arrays, page-locked ring, stream pool) may take
    in all: the smallest of an absolute ceiling (MRCHIP_BENCH_HOST_GB, default 64 GB per node), half of what the
    container's memory cgroup still leaves, and half of MemAvailable -- divided by the ranks of this node.
    /proc/meminfo shows the HOST's memory inside a container (the pool's boxes: 3 TB, under a cgroup limit of 300 GiB),
    and a cgroup file that says "max" means unknown, not unlimited: sizing 150 workers x 3.2 GB from MemAvailable is what
    ran two boxes out of memory in round 4 (`bench.py --config c5` with the CPU baseline; profiles/r04_README.md)."""
    ceiling = float(os.environ.get('MRCHIP_BENCH_HOST_GB', HOST_GB_DEFAULT)) * 1e9
    budget = ceiling
    try:
        with open('/proc/meminfo') as f:
            avail = [int(ln.split()[1]) * 1024 for ln in f if ln.startswith('MemAvailable')][0]
        budget = min(budget, 0.5 * avail)
    except Exception:
        pass
    lim, used = cgroup_memory()
    if lim is not None:
        budget = min(budget, 0.5 * max(0, lim - used))
    ranks_here = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1'))))
    return budget / ranks_here


def host_memory_report():
    lim, used = cgroup_memory()
    avail = None
    try:
        with open('/proc/meminfo') as f:
            avail = [int(ln.split()[1]) * 1024 for ln in f if ln.startswith('MemAvailable')][0]
    except Exception:
        pass
    return {'budget_GB': round(host_memory_budget() / 1e9, 1), 'ceiling_GB': float(os.environ.get('MRCHIP_BENCH_HOST_GB', HOST_GB_DEFAULT)),
            'cgroup_limit_GB': None if lim is None else round(lim / 1e9, 1), 'cgroup_used_GB': None if used is None else round(used / 1e9, 1),
            'mem_available_GB': None if avail is None else round(avail / 1e9, 1)}


def cpu_workers(cfg):
    """(workers, allowance per worker in bytes) of the CPU baseline for this workload under the host-memory budget"""
    # measured peak RSS of a worker: 0.83 GB (4000x3000 RGB), 1.05 (3300x4600 RGB), 3.2 (8000x6000 RGB): 1.0 GB per 36 MB
    # of page is the allowance
    per_worker = 1.0e9 * max(0.25, (cfg['w'] * cfg['h'] * cfg['c']) / 36e6)
    budget = 0.8 * host_memory_budget()           # the baseline runs first and alone; its memory is gone before the GPU legs
    return int(max(1, min(ncpus(), budget // per_worker))), per_worker


def cpu_baseline(cfg, seconds=12.0):
    """The oracle (C restatement of the reference, kind "port") decomposing pages of this workload on the host's cores:
    one worker PROCESS per core (each synthesises its own page, then decomposes it in a loop for `seconds`), as many as
    the host-memory budget allows (`cpu_workers`), each under an address-space limit of twice its allowance; plus one
    worker alone for the per-core figure.  A worker that fails ends the baseline with an error record, not the run."""
    import multiprocessing as mp
    ncpu = ncpus()
    workers, per_worker = cpu_workers(cfg)
    as_limit = 2.0 * per_worker + 1.5e9       # numpy / libgomp arenas reserve address space well beyond what they touch
    ctx = mp.get_context('fork')
    try:
        with ctx.Pool(1) as pool:
            n1, dt1, _ = pool.map(_cpu_worker, [(0, seconds / 2, cfg, as_limit)])[0]
        with ctx.Pool(workers) as pool:
            res = pool.map(_cpu_worker, [(k, seconds, cfg, as_limit) for k in range(workers)])
    except Exception as e:          # noqa: BLE001
        return {'value': None, 'unit': 'pages/s', 'cores': workers, 'kind': 'port', 'error': '%s: %s' % (type(e).__name__, e),
                'host_memory': host_memory_report()}
    rate = sum(n / dt for n, dt, _ in res)
    model = ''
    try:
        with open('/proc/cpuinfo') as f:
            for ln in f:
                if ln.startswith('model name'):
                    model = ln.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    capped = workers < ncpu
    return {'value': round(rate, 3), 'unit': 'pages/s', 'cores': workers, 'kind': 'port',
            'sample': '%d decompositions of this workload through oracle/mrc_oracle.c (gcc -O3, no fast-math: it is also '
                      'the bit-exact checker), %d worker processes x %.0f s, one page each in flight'
                      % (sum(n for n, _, _ in res), workers, seconds),
            'single_thread_value': round(n1 / dt1, 4), 'host_cpus': ncpu, 'cpu_model': model,
            'memory_capped': capped, 'worker_peak_rss_GB': round(max(p for _, _, p in res) / 1e9, 2),
            # what a reader who divides `value` of the line by this baseline should hold against ALL of the host, not the
            # memory-capped share of it: two extrapolations, labelled as such (never the reported `value`).  The linear one
            # is optimistic (workers share memory bandwidth and the boost clock: all-core runs of rounds 3-4 measured
            # 27-40 pages/s on these hosts)
            'all_cores_extrapolated': {'linear_from_measured': round(rate * ncpu / max(1, workers), 2),
                                       'single_thread_x_host_cpus': round(n1 / dt1 * ncpu, 2), 'unit': 'pages/s',
                                       'measured_on_cores': workers, 'host_cpus': ncpu},
            'host_memory': host_memory_report(),
            'reference_itself_pages_per_s_per_core': REFERENCE_PAGES_PER_S_PER_CORE,
            'note': ('cores = worker processes actually used: %s; the reference figure is the Python/Cython reference on '
                     'configs[1] measured in the survey container (BASELINE.md)'
                     % ('MEMORY-CAPPED -- %d of the host\'s %d cores, %.1f GB allowed per worker under a %.0f GB budget '
                        '(MRCHIP_BENCH_HOST_GB); all cores would need %.0f GB' % (workers, ncpu, per_worker / 1e9,
                                                                                  host_memory_budget() / 1e9, ncpu * per_worker / 1e9)
                        if capped else 'all host cores'))}


# ---------------------------------------------------------------------------------------------------------------
def sources_hash():
    """hash of the library's sources as the Makefile takes it at link time (csrc/*.hip sorted, mrchip_internal.h, mrchip.h)"""
    import glob
    src = os.path.join(ROOT, 'archive-pdf-tools_amd', 'csrc')
    files = sorted(glob.glob(os.path.join(src, '*.hip')), key=os.path.basename)
    files += [os.path.join(src, 'mrchip_internal.h'), os.path.join(ROOT, 'include', 'mrchip.h')]
    hh = hashlib.sha256()
    for f in files:
        with open(f, 'rb') as fh:
            hh.update(fh.read())
    return hh.hexdigest()[:16]


def library_is_current():
    """the in-tree libmrchip.so was linked from the sources that lie beside it (lib/BUILD_SRCHASH, csrc/Makefile)"""
    try:
        with open(os.path.join(ROOT, 'archive-pdf-tools_amd', 'lib', 'BUILD_SRCHASH')) as f:
            return f.read().strip() == sources_hash()
    except OSError:
        return False


def git_head():
    """Commit the numbers of this run belong to: the stamp `make` left beside the library (lib/BUILD_HEAD: the commit, '+'
    when the tracked tree was dirty at build time) -- the GPU box gets a snapshot without .git.  None, loudly, when the
    library was not linked from the sources beside it: a line must not carry the name of code that did not run."""
    if not library_is_current():
        if not getattr(git_head, 'warned', False):
            sys.stderr.write('bench.py: libmrchip.so was not built from the sources beside it (lib/BUILD_SRCHASH): run make; '
                             '`head` is withheld\n')
            git_head.warned = True
        return None
    try:
        with open(os.path.join(ROOT, 'archive-pdf-tools_amd', 'lib', 'BUILD_HEAD')) as f:
            return f.read().strip() or None
    except OSError:
        return None


# kernel symbol (as rocprofv3 prints it) behind a profile name; the page and the hOCR-box launches of Sauvola are two
# instances of one template (the 3rd argument: both polarities)
KERNEL_SYMBOL = {'optimise_rgb': r'optimise_(band|packed)_kernel<3', 'optimise_gray': r'optimise_(band|packed)_kernel<1',
                 'sauvola': r'sauvola(_tab)?_kernel<\d+, \w+, false', 'sauvola_boxes': r'sauvola(_tab)?_kernel<\d+, \w+, true'}


def _profile_kernel(kernels, name):
    import re
    pat = re.compile(KERNEL_SYMBOL.get(name, re.escape(name)))
    hits = [(k, v) for k, v in kernels.items() if pat.search(k)]
    return max(hits, key=lambda kv: kv[1].get('launches', 0))[1] if hits else None


PROFILE_ROUND = 'r06'


def _pick_profile(kind, config, inflight):
    """The newest committed profiles/<round>_*<kind>_summary.json whose bench command ran THIS configuration (--config)
    with this many batches in flight; None when there is none -- numbers of another workload are not attached."""
    import glob
    best = None
    # (profiles of THIS round only: the kernels of earlier rounds are different code under similar names)
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', PROFILE_ROUND + '_*%s_summary.json' % kind)) +
                    glob.glob(os.path.join(ROOT, 'profiles', PROFILE_ROUND + '_%s_summary.json' % kind))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        args = (d.get('bench_args') or '').split()

        def opt(name, default):
            return args[args.index(name) + 1] if name in args and args.index(name) + 1 < len(args) else default
        cfgname = opt('--config', 'c2')
        infl = int(opt('--inflight', CONFIGS.get(cfgname, {}).get('inflight', 1)))
        if cfgname == config and infl == inflight:
            best = (f, d)
        elif cfgname == config and kind == 'valu' and best is None:
            best = (f, d)              # instruction counts per pixel do not depend on how many batches are in flight
    return best


def same_kernel_sources(srchash):
    """a committed profile describes THIS build when the hash of the library sources it recorded is the current one
    (commits that touch only tests, docs or bench.py keep a profile valid; any change under csrc/ or include/ retires it)"""
    return bool(srchash) and srchash == sources_hash() and library_is_current()


def pmc_traffic(name, alg_per_launch, config='c2', inflight=3):
    """HBM bytes per launch of kernel `name` from the committed rocprofv3 PMC passes (profiles/*_pmc_summary.json:
    2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction), scaled to this run's launch size -- with the provenance, so a
    stale profile cannot pass for a measurement of this build."""
    try:
        hit = _pick_profile('pmc', config, inflight)
        if not hit:
            return None, None
        f, d = hit
        src = {'file': os.path.relpath(f, ROOT), 'profiled_head': d.get('head'), 'this_head': git_head(),
               'kernel_sources_equal': same_kernel_sources(d.get('srchash'))}
        if not src['kernel_sources_equal']:
            return None, src          # counters of other code are not attached to this line
        v = _profile_kernel(d['kernels'], name)
        if v and 'alg_bytes_per_launch' in d.get('scale', {}).get(name, {}):
            return round(v['hbm_bytes_per_launch'] * alg_per_launch / d['scale'][name]['alg_bytes_per_launch']), src
        return None, src
    except Exception:
        return None, None


def valu_roofline(name, alg_per_launch, bytes_per_px, config='c2', inflight=3):
    """The instruction side of the roofline for kernel `name` from the committed SQ / GRBM counter passes
    (profiles/*_valu_summary.json, written by tools/profile_round.sh beside the FETCH / WRITE passes): VALU
    wave-instructions per 64 pixels (= lane-instructions per pixel), cycles per instruction, VALU-busy fraction."""
    try:
        hit = _pick_profile('valu', config, inflight)
        if not hit:
            return None
        f, d = hit
        if not same_kernel_sources(d.get('srchash')):
            return None
        sc = d.get('scale', {}).get(name, {})
        v = _profile_kernel(d['kernels'], name)
        if v and 'alg_bytes_per_launch' in sc:
            px = sc['alg_bytes_per_launch'] / bytes_per_px
            return {'insts_per_px': round(v['valu_wave_insts_per_launch'] / (px / 64.0), 2), 'busy_frac': v['busy_frac'],
                    'cycles_per_inst': v['cycles_per_inst'],
                    'what': 'VALU wave-instructions per 64 pixels; fraction of the launch a SIMD VALU was busy; issue '
                            'cycles per instruction -- at ~4 cycles each and 1024 SIMDs x 2.4 GHz the chip issues 39 T '
                            'lane-instructions/s, i.e. 0.6 of the byte roofline needs <= 16 per Sauvola pixel, <= 57 per '
                            'optimise RGB pixel',
                    'source': os.path.relpath(f, ROOT), 'profiled_head': d.get('head'), 'this_head': git_head()}
        return None
    except Exception:
        return None


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: this process never touches the GPU, starts N fresh child processes
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; ordinary children, never os.exec*), relays rank 0's JSON line on its
    own stdout and everything else on stderr, and exits non-zero if any rank does."""
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), MRCHIP_BENCH_SPAWNED='1')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE, text=True))
    import threading
    outs = [''] * n

    def drain(r):                       # all pipes are read at once: no rank ever blocks on a full one
        outs[r] = procs[r].communicate()[0]

    ths = [threading.Thread(target=drain, args=(r,), daemon=True) for r in range(n)]
    for th in ths:
        th.start()
    # A rank that dies early (GPU fault, import error) leaves the others inside a collective: watch the children, and
    # once one has failed give the rest a grace period, then end them (fresh children of this process, by PID); the whole
    # run is bounded as well.
    deadline = time.time() + float(os.environ.get('MRCHIP_BENCH_TIMEOUT', '3600'))
    failed_at = None
    while any(p.poll() is None for p in procs):
        now = time.time()
        if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
            failed_at = now
        if (failed_at is not None and now - failed_at > 20.0) or now > deadline:
            why = 'a rank failed' if failed_at is not None else 'timeout'
            for r, p in enumerate(procs):
                if p.poll() is None:
                    sys.stderr.write('bench.py: ending rank %d (pid %d): %s\n' % (r, p.pid, why))
                    p.terminate()
            time.sleep(3.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    for th in ths:
        th.join(30.0)
    lines = [ln for ln in outs[0].splitlines() if ln.startswith('{')]
    for r, out in enumerate(outs):
        for ln in out.splitlines():
            if not (r == 0 and lines and ln == lines[-1]):
                sys.stderr.write('[rank %d] %s\n' % (r, ln))
    bad = [(r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0]
    if bad:
        sys.stderr.write('bench.py: ranks failed (rank, exit code): %r\n' % (bad,))
    if lines:
        print(lines[-1])
        sys.stdout.flush()
    elif not bad:
        sys.stderr.write('bench.py: rank 0 printed no result line\n')
        return 1
    return 1 if bad else 0


_RESULT_FD = None


def emit_line(d):
    """the one JSON line, to the process's real stdout (main() points fd 1 at stderr for everything else)"""
    data = (json.dumps(d) + '\n').encode()
    if _RESULT_FD is None:
        sys.stdout.write(data.decode()); sys.stdout.flush()
    else:
        os.write(_RESULT_FD, data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', default='c2', choices=sorted(CONFIGS))
    ap.add_argument('--pages', type=int, default=None, help='pages per GPU per step (default: per config)')
    ap.add_argument('--inflight', type=int, default=None, help='device batches in flight per GPU (one HIP stream each)')
    ap.add_argument('--distinct', type=int, default=None, help='distinct synthetic pages per GPU, cycled through the batch')
    ap.add_argument('--e2e-pages', type=int, default=512, help='pages per GPU pushed through the streaming pipeline (0: skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline', action='store_true', help='time the CPU port on this configuration too (default: only on c2, the workload the metric is quoted on)')
    ap.add_argument('--no-extras', action='store_true', help='timed region only (profiling runs)')
    a = ap.parse_args()
    cfg = dict(CONFIGS[a.config])
    for k in ('pages', 'inflight', 'distinct'):
        if getattr(a, k) is not None:
            cfg[k] = getattr(a, k)
    W, H, Cc = cfg['w'], cfg['h'], cfg['c']
    sauvola_only = a.config == 'c3gray'

    if a.gpus < 1:
        ap.error('--gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a.gpus, sys.argv[1:]))        # no launcher: be the launcher
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != a.gpus:
        sys.stderr.write('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks\n' % (a.gpus, world))
        sys.exit(2)
    # stdout carries exactly ONE line, the result: from here on file descriptor 1 is stderr for everything this process
    # loads (round 6: on one box of the pool librccl printed its version banner to stdout AFTER the line -- an
    # ncclCommInitRank that had timed out finished late), and the line is written to the real stdout by emit_line
    sys.stdout.flush()
    global _RESULT_FD
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get('MRCHIP_BENCH_DRYRUN'):
        # CPU test hook of the rank plumbing (no GPU, no library): the ranks meet over gloo, rank 0 prints a line
        import torch.distributed as tdist
        from mrchip import dist as mdist
        tdist.init_process_group('gloo')
        comm = mdist.TorchComm(tdist)
        ids = comm.allgather_obj({'rank': rank, 'local_rank': local_rank, 'pages': mdist.shard_pages(8, rank, world)})
        if os.environ.get('MRCHIP_BENCH_DRYRUN') == 'fail' and rank == world - 1:
            sys.exit(7)
        if rank == 0:
            emit_line({'dryrun': True, 'n_gpus': world, 'ranks': ids, 'spawned': bool(os.environ.get('MRCHIP_BENCH_SPAWNED'))})
            sys.stdout.flush()
        comm.barrier()
        tdist.destroy_process_group()       # (a rank that exits with gloo's threads alive aborts now and then: "terminate called ...")
        return

    # the CPU baseline forks worker processes: before the first GPU call of this process
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not a.no_extras and (a.config == 'c2' or a.cpu_baseline):
        cpu = cpu_baseline(cfg)

    from mrchip import _lib, mrc, synth
    from mrchip import dist as mdist
    lib = _lib.load()
    ctx = _lib.Context(local_rank % max(1, _lib.mrchip_visible_devices()))
    info = ctx.info()
    transport = 'none (one rank)'
    if world == 1:
        comm = mdist.SoloComm()
        if not a.no_extras and os.environ.get('MRCHIP_BENCH_RCCL1', '1') != '0':
            # the same control-plane code the N > 1 runs use, on a communicator of one: says whether RCCL works on this box
            comm, transport = rccl_of_one(ctx, mdist)
    elif os.environ.get('MRCHIP_BENCH_COMM') == 'gloo':
        # test hook: the N > 1 logic of this file on a box with fewer GPUs than ranks (RCCL refuses two ranks on one
        # GPU); the ranks then share GPU local_rank mod visible-devices
        import torch.distributed as tdist
        tdist.init_process_group('gloo')
        comm = mdist.TorchComm(tdist)
        transport = 'gloo (test hook)'
    else:
        comm, transport = mdist.make_comm(ctx, rank, world)      # RCCL; files of /tmp only if RCCL cannot come up on every rank

    # ---- work queue: rank 0 owns the descriptor table and broadcasts it over RCCL (bytes, not pixels); page j of the
    # global list (world * pages-per-GPU pages) belongs to rank j mod world and is the synthetic page of seed
    # seeds[j mod len(seeds)] -- the pages the reference digested (tests/golden/configs.json)
    desc = comm.bcast_obj({'config': a.config, 'w': W, 'h': H, 'c': Cc, 'pages_per_rank': cfg['pages'], 'seeds': cfg['seeds'],
                           'distinct': cfg['distinct']} if rank == 0 else None)
    seeds = desc['seeds']
    my_global = mdist.shard_pages(world * desc['pages_per_rank'], rank, world)
    # at most `distinct` different images per rank: the t-th page of this rank shows image t mod distinct
    slot_seed = [seeds[my_global[t] % len(seeds)] for t in range(min(desc['distinct'], len(my_global)))]
    made = synth.synth_pages([dict(w=W, h=H, channels=Cc, seed=s, noise_sigma=6.0, line_div=cfg['line_div']) for s in slot_seed])
    host_pages = [(img, hocr, mrc.hocr_boxes(hocr, W, H)) for img, hocr in made]
    nd = len(host_pages)
    window = mrc._window_size(cfg['dpi'])

    # the rank's pages are split into `inflight` device batches, each on its own HIP stream, so that the
    # latency-bound row-sequential kernels of one batch overlap the streaming kernels of another
    nb = max(1, min(cfg['inflight'], cfg['pages']))
    sizes = [cfg['pages'] // nb + (1 if i < cfg['pages'] % nb else 0) for i in range(nb)]
    batches, first_of = [], {}
    k = 0
    for bi, sz in enumerate(sizes):
        bt = mrc.Batch(ctx, sz, W, H, Cc)
        for i in range(sz):
            img, hocr, boxes = host_pages[k % nd]
            bt.upload(i, img)
            bt.set_boxes(i, boxes)
            first_of.setdefault(k % nd, (bi, i))        # where the first copy of each distinct image sits
            k += 1
        batches.append(bt)
    ctx.sync()

    def step():
        if sauvola_only:
            for bt in batches:
                bt.threshold(cfg['dpi'], 0.34)           # one Sauvola launch per batch
            return
        for bt in batches:
            bt.mask_begin(window)                        # luma, hOCR-box thresholds, noise estimate
        for bt in batches:
            bt.mask_finish(bt.sigmas(), True)            # host: Gaussian tables; decisions, blur, Sauvola, denoise
            bt.layers(cfg['fg'], cfg['bg'])              # fg + bg optimise (one launch), thumbnails

    def barrier():
        ctx.sync()
        comm.barrier()

    for _ in range(a.warmup):
        step()
    barrier()
    ctx.prof_enable(True)
    ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = comm.max_f64(time.perf_counter() - t0)
    prof = ctx.prof_report()
    ctx.prof_enable(False)

    # ------------------------------------------------------------------------------------------- untimed extras
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
                bt.layers(cfg['fg'], cfg['bg'])
                ctx.sync()
        prof_iso = ctx.prof_report()
        ctx.prof_enable(False)

    if not a.no_extras:
        # ---- parity: every distinct page this rank decomposed, digests against the reference's
        want = {}
        try:
            with open(os.path.join(ROOT, 'tests', 'golden', 'configs.json')) as f:
                want = json.load(f).get(cfg['digests'] or '', {})
        except Exception:
            pass
        recs = []
        for slot, seed in enumerate(slot_seed):
            bi, i = first_of[slot]
            bt = batches[bi]
            m = bt.download_mask(i)
            if sauvola_only:
                hm = sha(m)
                ok = (str(seed) in want and hm == want[str(seed)]['out'])
                recs.append({'page': my_global[slot], 'rank': rank, 'seed': seed, 'mask_popcount': int(m.sum()), 'mask': hm,
                             'checked': str(seed) in want, 'ok': bool(ok)})
                continue
            fg = bt.download_layer(i, 0, _layer_size(W, H, cfg['fg']))
            bg = bt.download_layer(i, 1, _layer_size(W, H, cfg['bg']))
            hm, hf, hb = sha_many([m, fg, bg], 3)
            w_ = want.get(str(seed))
            ok = bool(w_ and (hm, hf, hb) == (w_['mask'], w_['fg'], w_['bg']))
            recs.append({'page': my_global[slot], 'rank': rank, 'seed': seed, 'mask_popcount': int(m.sum()), 'mask': hm, 'fg': hf,
                         'bg': hb, 'checked': bool(w_), 'ok': ok})
        allrecs = [r for part in comm.allgather_obj(recs) for r in part]
        if rank == 0:
            checked = [r for r in allrecs if r['checked']]
            extra['parity'] = {'pages_checked': len(checked), 'mismatches': sum(1 for r in checked if not r['ok']),
                               'ranks_reporting': len(set(r['rank'] for r in allrecs)),
                               'against': 'SHA-256 of mask / fg / bg made by the reference for the same synthetic pages '
                                          '(tests/golden/configs.json)'}
            if a.config == 'c5':
                try:
                    dg = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'digests.json')))
                    d5 = {505: dg['c5'], 506: dg['c5_506']}          # both distinct pages of the batch have reference digests
                    r0 = [r for r in allrecs if r['seed'] in d5]
                    extra['parity'] = {'pages_checked': len(r0),
                                       'mismatches': sum(1 for r in r0 if (r['mask'], r['fg'], r['bg']) !=
                                                         (d5[r['seed']]['mask'], d5[r['seed']]['fg'], d5[r['seed']]['bg'])),
                                       'against': 'tests/golden/digests.json c5 / c5_506 (reference)'}
                except Exception as e:     # pragma: no cover
                    extra['parity'] = {'error': str(e)}

    for bt in batches:
        bt.close()
    batches = []
    # how many ranks the control plane really joined (an allgather of one byte per rank over it), and whether it is RCCL
    rccl_ok = transport.startswith('rccl')
    joined = len(comm.allgather_bytes(b'r'))
    if rank == 0:
        extra['rccl_ok'] = bool(rccl_ok)
        extra['rccl_ranks'] = joined if rccl_ok else 0
        extra['control_plane_ranks'] = joined
        if world > 1 and not rccl_ok:
            sys.stderr.write('bench.py: WARNING: %d ranks, control plane is NOT RCCL: %s\n' % (world, transport))

    if not a.no_extras and not sauvola_only and a.e2e_pages > 0:
        e2e = e2e_stream(ctx, comm, mrc, cfg, host_pages, a.e2e_pages, rank, world)
        if rank == 0:
            extra['e2e'] = e2e
    if not a.no_extras and a.config == 'c2':
        st = stack_check(ctx, comm, mrc, synth, cfg, rank, world)
        if rank == 0:
            extra['config4_stack'] = st
    if rank == 0 and not a.no_extras and not sauvola_only:
        # single-page latency of the drop-in generator: one page, host arrays in and out
        img, hocr, _ = host_pages[0]
        lat = []
        ctx.prof_enable(True)
        for rep in range(4):
            if rep == 1:
                ctx.prof_reset()
            t1 = time.perf_counter()
            for _ in mrc.create_mrc_hocr_components(img, hocr, dpi=cfg['dpi'], bg_downsample=cfg['bg'], fg_downsample=cfg['fg'],
                                                    denoise_mask='fast', ctx=ctx):
                pass
            lat.append(time.perf_counter() - t1)
        pr = ctx.prof_report()
        ctx.prof_enable(False)
        extra['single_page'] = {'latency_ms': round(min(lat[1:]) * 1e3, 2), 'pages_per_s': round(1.0 / min(lat[1:]), 1),
                                'kernel_ms': {k: round(v['ms'] / 3, 3) for k, v in sorted(pr.items(), key=lambda kv: -kv[1]['ms'])[:8]},
                                'what': 'mrc.create_mrc_hocr_components on one page, three yields, pageable host arrays in and out'}
    if rank == 0 and not a.no_extras:
        # measured ceiling to read the roofline fractions against: D2D copy, read + write bytes
        extra['hbm_copy_GBps_measured'] = round(ctx.hbm_copy_bandwidth(1 << 30, 10), 1)

    # ------------------------------------------------------------------------------------------------ the line
    total_pages = cfg['pages'] * a.steps * world
    step_s = dt / a.steps

    def roofline_of(name, prof=prof):
        r = prof[name]
        ms = r['ms'] / r['launches']
        alg = r['alg_bytes'] / r['launches']
        achieved = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        traffic, src = pmc_traffic(name, alg, a.config, nb)
        bpp = {'optimise_rgb': 7.0, 'optimise_gray': 3.0, 'sauvola': 2.0, 'sauvola_boxes': 4.0}.get(name)
        return {'bound': 'hbm', 'kernel': name, 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic, 'traffic_source': src,
                'avg_launch_ms': round(ms, 4), 'launches': r['launches'], 'alg_bytes_per_launch': alg,
                'valu': valu_roofline(name, alg, bpp, a.config, nb) if bpp else None}

    # The dominant kernel: the one with the largest share of GPU time when kernels have the chip to themselves (the
    # isolated pass, when it ran).  With several batches in flight a launch's HIP-event duration includes the time it
    # shares CUs with other streams, so the ranking in the timed region moves with the overlap pattern (a memory-bound
    # kernel squeezed by two compute-bound ones can come out on top without being where the GPU's time goes).
    rank_by = prof_iso if prof_iso else prof
    dom = max(((k, v) for k, v in rank_by.items() if k in prof), key=lambda kv: kv[1]['ms']) if prof else None
    roof = roofline_of(dom[0]) if dom else None
    if roof is not None:
        roof['chosen_by'] = 'largest share of GPU time in the isolated pass' if prof_iso else 'largest share of GPU time in the timed region'
    if roof is not None and prof_iso and a.steps > 0:
        top_timed = max(prof.items(), key=lambda kv: kv[1]['ms'])[0]
        if top_timed != dom[0]:
            roof['largest_hip_event_share_in_timed_region'] = top_timed
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
    copy_gbps = extra.get('hbm_copy_GBps_measured')
    if copy_gbps:
        # against what a plain device copy of the same bytes reaches on THIS box (the spec peak is not reachable by any kernel)
        for rf in (roof, sauvola_roof):
            if rf:
                rf['frac_of_measured_copy'] = round(rf['achieved'] / copy_gbps, 4)
                if 'isolated' in rf:
                    rf['isolated']['frac_of_measured_copy'] = round(rf['isolated']['achieved'] / copy_gbps, 4)
    kernels = {k: {'ms_per_launch': round(v['ms'] / v['launches'], 4), 'launches': v['launches'],
                   'alg_GBps': round(v['alg_bytes'] / max(v['ms'], 1e-9) / 1e6, 1)} for k, v in sorted(prof.items())}
    alg_per_step = sum(v['alg_bytes'] for v in prof.values()) / max(a.steps, 1)
    if rank == 0:
        if sauvola_only:
            value = (2.0 * W * H * total_pages) / dt / 1e9            # BASELINE: 2 * W * H * N_pages / t
            unit = 'GB/s'
        else:
            value = total_pages / dt
            unit = 'pages/s'
        line = {
            'metric': cfg['metric'], 'value': round(value, 2), 'unit': unit,
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(step_s * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
            'config': {'workload': cfg['label'], 'pages_per_gpu_per_step': cfg['pages'], 'batches_in_flight': nb,
                       'distinct_pages_per_gpu': nd, 'hocr_boxes_per_page': int(len(host_pages[0][2])),
                       'sharding': 'page i of the global list -> rank i mod N (the control plane carries descriptors, records '
                                   'and the time only)', 'control_plane': transport},
            'roofline': roof, 'cpu_baseline': cpu,
            'pipeline_alg_GBps': round(alg_per_step / step_s / 1e9, 1),      # all kernels' algorithmic bytes / step time
            'pipeline_frac': round(alg_per_step / step_s / 1e9 / (HBM_PEAK_GBS * world), 4),      # ... of the HBM peak of the GPUs used
            'pages_per_s': round(total_pages / dt, 2),
            'sauvola_roofline': sauvola_roof,      # BASELINE.json also names "Sauvola HBM GB/s"
            'kernels': kernels, 'device': info['name'].strip(), 'head': git_head(),
            'srchash': sources_hash() if library_is_current() else None,      # of csrc/ + include/: what a profile must match to be quoted
        }
        line.update(extra)
        emit_line(line)
    comm.barrier()
    comm.close()
    # A run of several ranks whose control plane is NOT RCCL has measured the sharded pipeline (the line above stands, it
    # says `rccl_ok: false` and names the transport) but not the thing a multi-GPU run is for: every rank exits 3, so that a
    # launcher on real hardware cannot take a silent fallback to files for a pass.  Rehearsals that
    # cannot have RCCL by construction -- several ranks on one GPU, the gloo hook of the CPU tests -- say so with
    # MRCHIP_BENCH_ALLOW_NON_RCCL=1.
    if non_rccl_is_fatal(world, transport):
        sys.stderr.write('bench.py: %d ranks but the control plane is %r, not RCCL: exit status 3 '
                         '(MRCHIP_BENCH_ALLOW_NON_RCCL=1 for a rehearsal)\n' % (world, transport))
        sys.stdout.flush()
        sys.exit(3)


def non_rccl_is_fatal(world, transport, env=None):
    """several ranks + a control plane other than RCCL + no rehearsal switch"""
    env = os.environ if env is None else env
    return world > 1 and not str(transport).startswith('rccl') and env.get('MRCHIP_BENCH_ALLOW_NON_RCCL') != '1'


def rccl_of_one(ctx, mdist):
    """(comm, description) for a single rank: an RCCL communicator of one when librccl can make it (the calls of a
    multi-GPU run, exercised on every default run), else SoloComm -- said so in the line, never fatal."""
    import threading
    box = {}

    def init():
        try:
            box['comm'] = mdist.RcclComm(ctx, 0, 1, rendezvous=mdist.rendezvous_path() + '_solo%d' % os.getpid(),
                                         redirect_stdout=False)
        except Exception as e:          # noqa: BLE001
            box['err'] = '%s: %s' % (type(e).__name__, e)

    th = threading.Thread(target=init, daemon=True)
    with mdist._stdout_to_stderr(True):
        th.start()
        th.join(60.0)
    if 'comm' in box:
        return box['comm'], 'rccl (communicator of one rank)'
    why = box.get('err', 'ncclCommInitRank did not return within 60 s')
    sys.stderr.write('bench.py: RCCL communicator of one rank unavailable: %s\n' % why)
    return mdist.SoloComm(), 'none (one rank; RCCL unavailable -- %s)' % why


def _layer_size(w, h, ds):
    """(width, height) of a layer after mrc.py:422-428's thumbnail (host-side size rule of the library)"""
    if not ds:
        return (w, h)
    import ctypes as C
    from mrchip import _lib
    ow, oh = C.c_int(), C.c_int()
    _lib.load().mrchip_thumbnail_size(w, h, int(w / ds), int(h / ds), C.byref(ow), C.byref(oh))
    return (ow.value, oh.value)


def _cpus_of_node(node):
    cpus = set()
    with open('/sys/devices/system/node/node%d/cpulist' % node) as f:
        for part in f.read().strip().split(','):
            lo, _, hi = part.partition('-')
            cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def link_rates(ctx):
    """Measured PCIe rates of this GPU with page-locked host memory, one direction at a time (GB/s)."""
    import ctypes as C
    from mrchip import _lib, mrc
    w, h = 8000, 6000                       # 144 MB per copy
    bt = mrc.Batch(ctx, 1, w, h, 3)
    pin = ctx.pinned_empty((h, w, 3))
    pin[...] = 1
    out = {}
    bt.upload(0, pin)
    ctx.sync()
    t = time.perf_counter()
    for _ in range(4):
        bt.upload(0, pin)
    ctx.sync()
    out['h2d_GBps'] = round(4 * pin.nbytes / (time.perf_counter() - t) / 1e9, 1)
    # D2H of the same plane through the mask-free path: the image plane is not downloadable, so time a layer instead
    bt.set_boxes(0, np.zeros((0, 4), np.int32))
    bt.mask_begin(51)
    bt.mask_finish(bt.sigmas(), True)
    size, _, _ = bt.layers(None, None, which=1)
    bt.download_layer(0, 0, size, out=pin)
    t = time.perf_counter()
    for _ in range(4):
        bt.download_layer(0, 0, size, out=pin, wait=False)
    bt.sync()
    out['d2h_GBps'] = round(4 * pin.nbytes / (time.perf_counter() - t) / 1e9, 1)
    # both directions at once (a second batch = a second stream uploads while the first downloads): what a pipeline that
    # moves pages in and results out at the same time can get from the link, each way
    bt2 = mrc.Batch(ctx, 1, w, h, 3)
    pin2 = ctx.pinned_empty((h, w, 3))
    pin2[...] = 2
    bt2.upload(0, pin2)
    ctx.sync()
    t = time.perf_counter()
    for _ in range(4):
        bt2.upload(0, pin2)
        bt.download_layer(0, 0, size, out=pin, wait=False)
    ctx.sync()
    out['duplex_GBps_each_way'] = round(4 * pin.nbytes / (time.perf_counter() - t) / 1e9, 1)
    bt.close()
    bt2.close()
    return out


def link_mix_rate(ctx, mrc, cfg, host_page, rounds=6):
    """Copy-only run of the stream's own transfers: the pipeline's copy sizes (a page in; packed mask + fg + bg out), its
    batch size and its four streams -- two batches uploading while two download, rotating -- with no kernel in between
    and nothing else on the host.  Pages per second of that is what the link gives THIS mix; the streaming pipeline
    cannot beat it by more than noise.  Best of three samples of `rounds` x 4 batches (the symmetric
    `duplex_GBps_each_way` of link_rates moves equal bytes both ways, which this mix does not: 36 MB in, 41.5 MB out)."""
    W, H, Cc = cfg['w'], cfg['h'], cfg['c']
    img, hocr = host_page
    bts = [mrc.Batch(ctx, E2E_BATCH, W, H, Cc) for _ in range(E2E_SLOTS)]
    fgs, bgs = _layer_size(W, H, cfg['fg']), _layer_size(W, H, cfg['bg'])
    shp = lambda sz: (sz[1], sz[0]) if Cc == 1 else (sz[1], sz[0], 3)          # noqa: E731
    pin_in = []
    for _ in range(E2E_BATCH):
        p = ctx.pinned_empty(img.shape); p[...] = img; pin_in.append(p)
    outs = [[(ctx.pinned_empty((H, (W + 7) // 8)), ctx.pinned_empty(shp(fgs)), ctx.pinned_empty(shp(bgs))) for _ in range(E2E_BATCH)]
            for _ in range(2)]
    boxes = mrc.hocr_boxes(hocr, W, H)
    for bt in bts:                       # every batch holds finished layers to download
        for i in range(E2E_BATCH):
            bt.upload(i, pin_in[i]); bt.set_boxes(i, boxes)
        bt.mask_begin(mrc._window_size(cfg['dpi']))
        bt.mask_finish(bt.sigmas(), True)
        bt.layers(cfg['fg'], cfg['bg'])
        bt.sync()
    out_bytes = sum(a.nbytes for a in outs[0][0])
    best = 0.0
    # (an upload invalidates a batch's derived results, so the roles are fixed: the first half of the batches only takes
    # uploads, the second half only gives its finished layers -- two streams each way, like the pipeline in steady state)
    ups, dns = bts[:E2E_SLOTS // 2], bts[E2E_SLOTS // 2:]
    for _rep in range(3):
        ctx.sync()
        t = time.perf_counter()
        for r in range(rounds * E2E_SLOTS):
            up, dn = ups[r % len(ups)], dns[r % len(dns)]
            dst = outs[r & 1]
            for i in range(E2E_BATCH):
                up.upload(i, pin_in[i])
                dn.download_mask_packed(i, out=dst[i][0], wait=False)
                dn.download_layer(i, 0, fgs, out=dst[i][1], wait=False)
                dn.download_layer(i, 1, bgs, out=dst[i][2], wait=False)
        for bt in bts:
            bt.sync()
        dt = time.perf_counter() - t
        best = max(best, rounds * E2E_SLOTS * E2E_BATCH / dt)
    for bt in bts:
        bt.close()
    return {'pages_per_s': round(best, 1), 'in_GBps': round(best * W * H * Cc / 1e9, 1), 'out_GBps': round(best * out_bytes / 1e9, 1),
            'pages_per_sample': rounds * E2E_SLOTS * E2E_BATCH,
            'what': 'copy-only: %d batches of %d pages on %d streams, uploads and downloads of the stream\'s sizes at once, best of 3'
                    % (E2E_SLOTS, E2E_BATCH, E2E_SLOTS)}


def e2e_stream(ctx, comm, mrc, cfg, host_pages, n_pages, rank, world):
    """PCIe-inclusive throughput of the streaming pipeline: `n_pages` host pages per rank in, packed mask + fg + bg
    out into page-locked arrays, every rank at once (so that host-side contention between ranks shows)."""
    from mrchip import _lib
    W, H, Cc = cfg['w'], cfg['h'], cfg['c']
    nd = len(host_pages)
    res = {}
    # Host placement: the stream moves ~78 MB per page through page-locked host memory.  MRCHIP_BENCH_NUMA=1 binds this
    # process to the CPUs of the NUMA node the GPU hangs off before anything page-locked is allocated; measured on three
    # two-socket boxes of the pool (GPU on node 0 and on node 1) it made no difference (1 118 vs 1 118 pages/s), so the
    # default leaves the affinity alone and only records where the GPU sits.  (The boxes on which the stream runs at
    # ~810 pages/s are the ones whose link measures 53.6 instead of 56.9 GB/s outward.)
    placement = {'gpu_numa_node': ctx.numa_node(), 'bound': False}
    try:
        if placement['gpu_numa_node'] >= 0 and os.environ.get('MRCHIP_BENCH_NUMA', '0') == '1':
            cpus = _cpus_of_node(placement['gpu_numa_node'])
            if cpus:
                os.sched_setaffinity(0, cpus & os.sched_getaffinity(0) or cpus)
                placement['bound'] = True
    except Exception as e:          # noqa: BLE001 - placement is best effort
        placement['error'] = str(e)
    link = link_rates(ctx) if rank == 0 else None
    if link is not None:
        try:
            link['stream_mix_copy_only'] = link_mix_rate(ctx, mrc, cfg, (host_pages[0][0], host_pages[0][1]))
        except Exception as e:          # noqa: BLE001 - a measurement aid must not cost the line
            link['stream_mix_copy_only'] = {'error': str(e)[:200]}
    pool = mrc.StreamPool(ctx)        # device batches + page-locked result arrays made once, as a long-running caller would

    def run_stream(page_iter_factory, timed_hook=None):
        gen = mrc.decompose_stream(page_iter_factory(), dpi=cfg['dpi'], bg_downsample=cfg['bg'], fg_downsample=cfg['fg'],
                                   batch_pages=E2E_BATCH, slots=E2E_SLOTS, mask_format='packed', pool=pool)
        for _ in gen:                   # first pass also allocates the slots: run the stream twice, time the second
            pass
        comm.barrier()
        phases = {}
        gen = mrc.decompose_stream(page_iter_factory(), dpi=cfg['dpi'], bg_downsample=cfg['bg'], fg_downsample=cfg['fg'],
                                   batch_pages=E2E_BATCH, slots=E2E_SLOTS, mask_format='packed', pool=pool, stats=phases)
        t0 = time.perf_counter()
        n = 0
        out_bytes = 0
        for m, fg, bg in gen:
            n += 1
            out_bytes = m.nbytes + fg.nbytes + bg.nbytes
            if timed_hook:
                timed_hook(n)
        ctx.sync()
        comm.barrier()
        dt = comm.max_f64(time.perf_counter() - t0)
        in_bytes = W * H * Cc
        return {'pages_per_s': round(n * world / dt, 1), 'h2d_GBps_per_gpu': round(n * in_bytes / dt / 1e9, 1),
                'd2h_GBps_per_gpu': round(n * out_bytes / dt / 1e9, 1), 'seconds': round(dt, 3),
                'host_phase_seconds': {k: round(v, 3) for k, v in phases.items()}}, out_bytes

    out_bytes = 0
    for mode in ('pageable', 'pinned'):
        if mode == 'pinned':       # the page source decodes into page-locked arrays: uploads are asynchronous DMAs
            src = []
            for img, hocr, _ in host_pages[:min(nd, 8)]:
                p = ctx.pinned_empty(img.shape)
                p[...] = img
                src.append((p, hocr))
        else:
            src = [(img, hocr) for img, hocr, _ in host_pages]
        res[mode], out_bytes = run_stream(lambda: ((src[i % len(src)][0], src[i % len(src)][1]) for i in range(n_pages)))
        res[mode]['source'] = '%d host arrays recycled over %d pages (the HIP runtime has seen every buffer before)' % (len(src), n_pages)

    # ---- what a page loop that DECODES pages gets (recode.py:343-348): every page in a host buffer the runtime has never
    # seen.  (a) n_pages distinct pageable arrays handed to the stream as they are; (b) the same arrays copied by a
    # producer thread into a ring of page-locked buffers ahead of the stream.  Both passes of a run use
    # their own fresh arrays; bounded by the host memory that is free.
    page_bytes = W * H * Cc
    # one pass of distinct arrays is alive at a time (the previous pass's are released first): 40 % of this rank's
    # host-memory budget; the page-locked ring below takes at most 15 % of it and never more than 16 GB
    budget = host_memory_budget()
    n_fresh = int(max(E2E_BATCH * E2E_SLOTS, min(n_pages, 0.4 * budget // page_bytes)))
    if os.environ.get('MRCHIP_BENCH_FRESH', '1') != '0':
        def fresh_arrays():
            return [np.array(host_pages[i % nd][0], copy=True) for i in range(n_fresh)]       # new allocations, never uploaded
        keep = {}

        def fresh_factory():                # each pass of run_stream (warm, timed) gets arrays of its own; the previous
            keep.clear()                    # pass's are released BEFORE the next pass is timed (unmapping 18 GB takes a while)
            keep['arrs'] = arrs = fresh_arrays()
            return ((arrs[i], host_pages[i % nd][1]) for i in range(n_fresh))
        res['fresh_pageable'], out_bytes = run_stream(fresh_factory)
        res['fresh_pageable']['source'] = '%d distinct pageable arrays, each uploaded once' % n_fresh
        keep.clear()

        import threading
        RING = int(max(E2E_BATCH * (E2E_SLOTS + 2), min(E2E_BATCH * (E2E_SLOTS + 12), min(16e9, 0.15 * budget) // page_bytes)))
        RING_THREADS = 4
        import ctypes
        ring = [ctx.pinned_empty(host_pages[0][0].shape) for _ in range(RING)]
        state = {}

        def ring_factory():
            keep.clear()                    # (the previous pass's arrays go before this pass is timed)
            keep['arrs'] = arrs = fresh_arrays()
            st = {'taken': 0, 'lock': threading.Condition()}
            state['st'] = st

            ready = [threading.Event() for _ in range(n_fresh)]

            def producer(k):                    # RING_THREADS threads, page i by thread i mod RING_THREADS: one thread
                for i in range(k, n_fresh, RING_THREADS):       # copies ~15 GB/s, a page is 36 MB
                    with st['lock']:            # a slot is rewritten only when the page that used it has been handed out
                        while i >= st['taken'] + RING - E2E_BATCH:
                            st['lock'].wait(0.05)
                    ctypes.memmove(ring[i % RING].ctypes.data, arrs[i].ctypes.data, arrs[i].nbytes)   # a foreign call: no GIL
                    ready[i].set()
            go = threading.Event()          # set by the timed loop right after its clock starts: nothing is copied off the clock
            state['go'] = go

            def gated(k):
                go.wait()
                producer(k)
            for k in range(RING_THREADS):
                threading.Thread(target=gated, args=(k,), daemon=True).start()

            def pages():
                for i in range(n_fresh):
                    ready[i].wait()
                    yield ring[i % RING], host_pages[i % nd][1]
            return pages()

        def took(n):
            st = state['st']
            with st['lock']:
                st['taken'] = n
                st['lock'].notify_all()
        res['pinned_ring'], out_bytes = run_stream_ring(mrc, ctx, comm, cfg, pool, ring_factory, took, world, W * H * Cc,
                                                        lambda: state['go'].set())
        res['pinned_ring']['source'] = ('%d distinct pageable arrays copied by a producer thread into a ring of %d page-locked '
                                        'buffers ahead of the stream (%d copy threads)' % (n_fresh, RING, RING_THREADS))
        keep.clear()
        del ring
    pool.close()
    # the headline of this leg: pages in buffers the runtime has not seen (the recycled-array figures are upper bounds)
    fresh = [res[k] for k in ('fresh_pageable', 'pinned_ring') if k in res]
    best = max(fresh or list(res.values()), key=lambda r: r['pages_per_s'])
    out = {'pages_per_s': best['pages_per_s'], 'pages_per_gpu': n_pages, 'batch_pages': E2E_BATCH, 'slots': E2E_SLOTS,
           'host_placement': placement,
           'host_arrays': res, 'bytes_in_per_page': W * H * Cc, 'bytes_out_per_page': out_bytes,
           'what': 'mrc.decompose_stream: host page arrays in (pageable numpy / page-locked), packed 1-bpp mask + fg + bg '
                   'thumbnail out into page-locked arrays; upload, compute and download of rotating batches overlap'}
    if link:
        out['link_measured'] = link
        out['link_spec_GBps_per_direction'] = PCIE_SPEC_GBS
        lim = min(link['h2d_GBps'] / (W * H * Cc / 1e9), link['d2h_GBps'] / (out_bytes / 1e9))
        out['link_ceiling_pages_per_s_per_gpu'] = round(lim, 1)           # each direction at the rate it reaches alone
        out['frac_of_link_ceiling'] = round(best['pages_per_s'] / world / lim, 3)
        if link.get('duplex_GBps_each_way'):
            # ... at the rate both directions reach when they run at once.  Two measurements of that: equal bytes each way
            # (link_rates) and the stream's own mix on its own streams, copies only (link_mix_rate); the ceiling is the
            # larger -- round 5 quoted the first alone and the pipeline "exceeded" it by 4-10 %, because its outward
            # direction carries more bytes than its inward one and gets more than half of a symmetric duplex run
            dlim = link['duplex_GBps_each_way'] / (max(W * H * Cc, out_bytes) / 1e9)
            mix = (link.get('stream_mix_copy_only') or {}).get('pages_per_s') or 0.0
            out['duplex_symmetric_pages_per_s_per_gpu'] = round(dlim, 1)
            out['duplex_ceiling_pages_per_s_per_gpu'] = round(max(dlim, mix), 1)
            out['frac_of_duplex_ceiling'] = round(best['pages_per_s'] / world / max(dlim, mix), 3)
    return out


def run_stream_ring(mrc, ctx, comm, cfg, pool, factory, took, world, in_bytes, start_producers):
    """the pinned-ring variant of e2e_stream's timed loop: both passes report the pages handed out to the producer"""
    out_bytes = 0
    res = None
    for timed in (False, True):
        phases = {}
        pages = factory()                # allocates this pass's fresh arrays and parks the producer threads
        gen = mrc.decompose_stream(pages, dpi=cfg['dpi'], bg_downsample=cfg['bg'], fg_downsample=cfg['fg'],
                                   batch_pages=E2E_BATCH, slots=E2E_SLOTS, mask_format='packed', pool=pool, stats=phases)
        if timed:
            comm.barrier()               # every rank has its arrays
        t0 = time.perf_counter()
        start_producers()
        n = 0
        for m, fg, bg in gen:
            n += 1
            out_bytes = m.nbytes + fg.nbytes + bg.nbytes
            took(n)
        ctx.sync()
        if timed:
            comm.barrier()
            dt = comm.max_f64(time.perf_counter() - t0)
            res = {'pages_per_s': round(n * world / dt, 1), 'h2d_GBps_per_gpu': round(n * in_bytes / dt / 1e9, 1),
                   'd2h_GBps_per_gpu': round(n * out_bytes / dt / 1e9, 1), 'seconds': round(dt, 3),
                   'host_phase_seconds': {k: round(v, 3) for k, v in phases.items()}, 'pages_copied_before_the_clock': 0}
    return res, out_bytes


def stack_check(ctx, comm, mrc, synth, cfg, rank, world):
    """BASELINE.json configs[3]: ONE 512-page stack (the 64 reference-digested pages, mixed), page i -> rank i mod N;
    every rank streams its shard, digests every output, the records are gathered on rank 0 and all 512 are checked."""
    with open(os.path.join(ROOT, 'tests', 'golden', 'configs.json')) as f:
        want = json.load(f)['c2_pages']
    seeds = sorted(int(s) for s in want)
    mine = list(range(rank, STACK_PAGES, world))
    need = sorted(set(seeds[(i * 5) % len(seeds)] for i in mine))
    made = dict(zip(need, synth.synth_pages([dict(w=cfg['w'], h=cfg['h'], channels=3, seed=s, noise_sigma=6.0, line_div=60)
                                             for s in need])))
    t0 = time.perf_counter()
    recs, pend = [], []
    gen = mrc.decompose_stream(((made[seeds[(i * 5) % len(seeds)]][0], made[seeds[(i * 5) % len(seeds)]][1]) for i in mine),
                               bg_downsample=3, batch_pages=32, mask_format='bool', ctx=ctx)
    for i, (m, fg, bg) in zip(mine, gen):
        pend.append((i, m, fg, bg))
        if len(pend) == 16 or i == mine[-1]:          # views stay valid for batch_pages (32) further pages
            hs = sha_many([x for p in pend for x in p[1:]])
            for k, p in enumerate(pend):
                recs.append({'page': p[0], 'rank': rank, 'mask_popcount': int(np.count_nonzero(p[1])), 'mask': hs[3 * k],
                             'fg': hs[3 * k + 1], 'bg': hs[3 * k + 2]})
            pend = []
    dt = comm.max_f64(time.perf_counter() - t0)
    allrecs = [r for part in comm.allgather_obj(recs) for r in part]
    if rank != 0:
        return None
    bad = 0
    for r in allrecs:
        w_ = want[str(seeds[(r['page'] * 5) % len(seeds)])]
        if (r['mask'], r['fg'], r['bg'], r['mask_popcount']) != (w_['mask'], w_['fg'], w_['bg'], w_['mask_sum']) or \
                r['rank'] != r['page'] % world:
            bad += 1
    return {'pages': len(allrecs), 'distinct': len(seeds), 'all_pages_present': sorted(r['page'] for r in allrecs) == list(range(STACK_PAGES)),
            'mismatches': bad, 'seconds_incl_hashing': round(dt, 2),
            'what': '512-page stack sharded page i -> rank i mod N, per-page {page, rank, mask_popcount, sha256 x3} records '
                    'gathered over the control plane (config.control_plane) and checked against the reference digests on rank 0'}


if __name__ == '__main__':
    main()
