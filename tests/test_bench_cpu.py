"""CPU suite: the parts of bench.py that need no GPU -- the build stamp, the provenance-checked traffic figures it
quotes from the committed PMC passes, the streaming defaults it reports."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_profile_figures_come_from_profiles_of_this_build_only():
    """bench.py quotes counter figures from profiles/<round>_*_summary.json of the CURRENT round, and only when the summary
    recorded the hash of the library sources (csrc/ + include/) that this tree still has: the profiles of earlier rounds
    -- other kernels under similar names, no hash -- are never attached."""
    import bench
    assert bench.PROFILE_ROUND == 'r06'
    assert bench.library_is_current(), 'libmrchip.so is older than csrc/: run make (the CPU suite runs after build())'
    assert bench.same_kernel_sources(bench.sources_hash()) and not bench.same_kernel_sources(None)
    assert not bench.same_kernel_sources('0' * 16)
    old = bench.PROFILE_ROUND
    try:
        bench.PROFILE_ROUND = 'r04'          # a round whose summaries carry no source hash
        assert bench._pick_profile('valu', 'c2', 1)
        assert bench.valu_roofline('sauvola', 3.072e9, 2.0, 'c2', 3) is None
        bench.PROFILE_ROUND = 'r03'
        t, src = bench.pmc_traffic('optimise_rgb', 21.504e9, 'c2', 3)
        assert t is None and src['kernel_sources_equal'] is False and src['file'] == 'profiles/r03_pmc_summary.json'
    finally:
        bench.PROFILE_ROUND = old
    for name, alg in (('optimise_rgb', 21.504e9), ('sauvola', 3.072e9)):
        t, src = bench.pmc_traffic(name, alg)
        assert t is None or (src['kernel_sources_equal'] and t > 0.9 * alg)


def test_host_memory_budget_is_bounded(monkeypatch):
    """Round 4 lost two GPU boxes to a CPU-baseline pool sized from the HOST's MemAvailable inside a 300 GiB memory cgroup.
    Every host allocation of a bench run now comes out of one budget: an absolute ceiling (MRCHIP_BENCH_HOST_GB), at most
    half of what the cgroup / the host leaves, split between the ranks of the node."""
    import bench
    monkeypatch.delenv('LOCAL_WORLD_SIZE', raising=False)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setenv('MRCHIP_BENCH_HOST_GB', '8')
    b1 = bench.host_memory_budget()
    assert 0 < b1 <= 8e9
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')
    assert abs(bench.host_memory_budget() - b1 / 8) < 1e6
    monkeypatch.delenv('LOCAL_WORLD_SIZE')
    monkeypatch.setenv('MRCHIP_BENCH_HOST_GB', '64')
    monkeypatch.setattr(bench, 'cgroup_memory', lambda: (300 << 30, 290 << 30))       # 10 GiB left in the cgroup
    assert bench.host_memory_budget() <= 0.5 * (10 << 30)
    monkeypatch.setattr(bench, 'cgroup_memory', lambda: (None, None))                  # "max": unknown, the ceiling holds
    assert bench.host_memory_budget() <= 64e9
    monkeypatch.setattr(bench, 'ncpus', lambda: 256)
    for name, most in (('c2', 52), ('c3', 40), ('c5', 13)):        # 0.8 x 64 GB over 1.0 / 1.27 / 4.0 GB per worker
        workers, per = bench.cpu_workers(bench.CONFIGS[name])
        assert 1 <= workers <= most and workers * per <= 0.8 * 64e9


def test_cpu_worker_over_its_allowance_fails_without_taking_the_host_along():
    """A baseline worker runs under RLIMIT_AS: one that outgrows it raises MemoryError inside the worker, and cpu_baseline
    reports an error record instead of a figure (the bench line still prints)."""
    import multiprocessing as mp
    import bench
    cfg = dict(bench.CONFIGS['c2'], w=2000, h=1500)
    with mp.get_context('fork').Pool(1) as pool:
        import pytest
        with pytest.raises(Exception):       # MemoryError from numpy, or the loader failing to map a library: never a figure
            pool.map(bench._cpu_worker, [(0, 0.1, cfg, 16 << 20)])       # 16 MiB over the current size: cannot hold a page
    with mp.get_context('fork').Pool(1) as pool:
        n, dt, peak = pool.map(bench._cpu_worker, [(0, 0.1, cfg, 4 << 30)])[0]
        assert n >= 1 and 0 < peak < 2e9


def test_committed_bench_lines_have_the_contract_fields():
    for cfg in ('c2', 'c3', 'c3gray', 'c5'):
        with open(os.path.join(ROOT, 'profiles', 'r03_bench_%s.json' % cfg)) as f:
            d = json.load(f)
        for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                    'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
            assert key in d, (cfg, key)
        assert d['vs_baseline'] is None and d['dtype'] == 'u8' and 'workload' in d['config']
        r = d['roofline']
        assert r['bound'] == 'hbm' and r['peak'] == 8000.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-4
        assert d['parity']['mismatches'] == 0
    with open(os.path.join(ROOT, 'profiles', 'r03_bench_c2.json')) as f:
        c2 = json.load(f)
    assert c2['cpu_baseline']['kind'] == 'port' and c2['cpu_baseline']['cores'] >= 1
    assert c2['roofline']['kernel'] == 'optimise_rgb'            # ranked in the isolated pass, not by overlapped HIP events
    assert c2['config4_stack']['mismatches'] == 0 and c2['config4_stack']['all_pages_present']
    # round 3: the instruction side of the roofline, the control plane's status, page sources the runtime has never seen
    v = c2['roofline']['valu']
    assert v['source'].startswith('profiles/') and 30 < v['insts_per_px'] < 60 and 0.3 < v['busy_frac'] <= 1.0
    assert 3.5 < v['cycles_per_inst'] < 5.0 and 35 < c2['sauvola_roofline']['valu']['insts_per_px'] < 60
    assert c2['rccl_ok'] is True and c2['rccl_ranks'] == 1 and c2['n_gpus'] == 1
    assert {'fresh_pageable', 'pinned_ring'} <= set(c2['e2e']['host_arrays'])
    assert c2['e2e']['pages_per_s'] == max(c2['e2e']['host_arrays'][k]['pages_per_s'] for k in ('fresh_pageable', 'pinned_ring'))
    with open(os.path.join(ROOT, 'profiles', 'r03_bench_c5.json')) as f:
        assert json.load(f)['parity']['pages_checked'] == 2                     # both 8000x6000 pages have reference digests


def test_build_stamp_and_stream_defaults():
    import bench
    head = bench.git_head()
    assert head and len(head.rstrip('+')) >= 7
    sys.path.insert(0, os.path.join(ROOT, 'archive-pdf-tools_amd'))
    import inspect
    from mrchip import mrc
    sig = inspect.signature(mrc.decompose_stream)
    assert (sig.parameters['batch_pages'].default, sig.parameters['slots'].default) == (bench.E2E_BATCH, bench.E2E_SLOTS)


def _run_bench(args, env_extra, timeout=300):
    import subprocess
    env = dict(os.environ, **env_extra)
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus_n_without_a_launcher_spawns_n_ranks():
    """`python bench.py --gpus 2` used to run ONE rank and print n_gpus 1."""
    r = _run_bench(['--gpus', '2'], {'MRCHIP_BENCH_DRYRUN': '1'})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout                      # exactly one result line on the parent's stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['spawned'] is True
    assert [x['rank'] for x in d['ranks']] == [0, 1] and [x['local_rank'] for x in d['ranks']] == [0, 1]
    assert sorted(p for x in d['ranks'] for p in x['pages']) == list(range(8))       # page i -> rank i mod N


def test_a_failing_rank_fails_the_run():
    r = _run_bench(['--gpus', '2'], {'MRCHIP_BENCH_DRYRUN': 'fail'})
    assert r.returncode != 0
    assert 'ranks failed' in r.stderr


def test_world_size_must_match_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0', MRCHIP_BENCH_DRYRUN='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 2 and 'WORLD_SIZE=2' in r.stderr


def test_eight_ranks_dry_run():
    """Dress rehearsal of the N = 8 launcher path without a GPU: eight children, one line, page i -> rank i mod 8."""
    r = _run_bench(['--gpus', '8'], {'MRCHIP_BENCH_DRYRUN': '1'}, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and [x['rank'] for x in d['ranks']] == list(range(8))
    assert [x['pages'] for x in d['ranks']] == [[i] for i in range(8)]


def test_profile_figures_are_only_attached_to_the_configuration_they_were_taken_on():
    """the newest *_summary.json used to be quoted for every --config / --inflight."""
    import bench
    old = bench.PROFILE_ROUND
    try:
        bench.PROFILE_ROUND = 'r04'
        hit = bench._pick_profile('valu', 'c2', 1)
        assert hit and 'inflight1' in os.path.basename(hit[0])
        assert bench._pick_profile('valu', 'c3gray', bench.CONFIGS['c3gray']['inflight']) is None       # no c3gray pass in round 4
        assert bench._pick_profile('pmc', 'c5', bench.CONFIGS['c5']['inflight']) is None
        assert bench.pmc_traffic('optimise_rgb', 1e9, 'c5', bench.CONFIGS['c5']['inflight']) == (None, None)
        bench.PROFILE_ROUND = 'r03'          # the selection logic on the fuller round-3 set
        hit = bench._pick_profile('pmc', 'c3gray', bench.CONFIGS['c3gray']['inflight'])
        assert hit and 'c3gray' in os.path.basename(hit[0])
        hit = bench._pick_profile('pmc', 'c2', 3)
        assert hit and os.path.basename(hit[0]) == 'r03_pmc_summary.json'
        assert bench._pick_profile('pmc', 'c5', bench.CONFIGS['c5']['inflight']) is None
    finally:
        bench.PROFILE_ROUND = old


def test_round4_bench_lines():
    """The lines of round 4 that survived (profiles/r04_README.md): the full default line and the timed region of the final sources."""
    with open(os.path.join(ROOT, 'profiles', 'r04_bench_c2.json')) as f:
        d = json.load(f)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'pipeline_frac'):
        assert key in d, key
    assert d['value'] > 9500 and d['unit'] == 'pages/s' and d['vs_baseline'] is None and d['dtype'] == 'u8'
    r = d['roofline']
    assert r['kernel'] == 'optimise_rgb' and r['bound'] == 'hbm' and r['traffic'] is None           # no counter pass this round
    assert abs(r['frac'] - r['achieved'] / 8000.0) < 1e-4 and 0 < r['frac_of_measured_copy'] < 1 and 'isolated' in r
    assert abs(d['pipeline_frac'] - d['pipeline_alg_GBps'] / 8000.0) < 1e-3
    s = d['sauvola_roofline']
    assert s['isolated']['frac'] > 0.2 and 25 < s['valu']['insts_per_px'] < 30
    assert d['parity']['mismatches'] == 0 and d['parity']['pages_checked'] == 16
    assert d['config4_stack']['mismatches'] == 0 and d['config4_stack']['all_pages_present']
    assert d['cpu_baseline']['kind'] == 'port' and d['cpu_baseline']['cores'] >= 1
    assert 'optimise_bands' in d['kernels'] and 'provenance' in d
    with open(os.path.join(ROOT, 'profiles', 'r04_bench_c2_timed_region_final_sources.json')) as f:
        t = json.load(f)
    assert t['value'] > 10000 and t['kernels']['optimise_rgb']['launches'] == 30          # one optimise launch per batch and step
    with open(os.path.join(ROOT, 'profiles', 'r04_bench_gpus8_on_one_gpu.json')) as f:
        g = json.loads(f.read().strip().splitlines()[-1])
    assert g['n_gpus'] == 8 and g['control_plane_ranks'] == 8 and g['rccl_ok'] is False
    assert g['config4_stack']['all_pages_present'] and g['config4_stack']['mismatches'] == 0 and g['parity']['mismatches'] == 0


def test_round5_bench_lines():
    """The lines of round 5 (profiles/r05_README.md): the driver's command on the committed head, the three other
    configurations (one gpurun call each), the eight-rank rehearsal on one GPU."""
    with open(os.path.join(ROOT, 'profiles', 'r05_bench_c2.json')) as f:
        d = json.loads(f.read().strip().splitlines()[-1])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'pipeline_frac', 'head', 'srchash'):
        assert key in d, key
    assert d['value'] > 10000 and d['unit'] == 'pages/s' and d['vs_baseline'] is None and d['dtype'] == 'u8' and d['n_gpus'] == 1
    assert d['head'] and len(d['head']) >= 12 and not d['head'].endswith('+') and d['srchash']      # built from a clean, committed tree
    r = d['roofline']
    assert r['kernel'] == 'optimise_rgb' and r['bound'] == 'hbm' and abs(r['frac'] - r['achieved'] / 8000.0) < 1e-4 and 'isolated' in r
    assert r['isolated']['frac'] > 0.45 and d['sauvola_roofline']['isolated']['frac'] > 0.23
    # counter traffic of THIS build's kernels (profiles/r05_pmc_summary.json records the same source hash)
    assert r['traffic'] and 1.2 < r['traffic'] / r['alg_bytes_per_launch'] < 1.4 and r['traffic_source']['kernel_sources_equal'] is True
    assert 1.8 < d['sauvola_roofline']['traffic'] / d['sauvola_roofline']['alg_bytes_per_launch'] < 2.3      # (2.55 before the page pass stopped storing mask bytes)
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and not c.get('error') and c['value'] > 10
    assert c['memory_capped'] is True and c['cores'] <= 64 and c['worker_peak_rss_GB'] < 1.0 and c['host_memory']['cgroup_limit_GB'] > 300
    assert d['parity']['mismatches'] == 0 and d['parity']['pages_checked'] == 16
    assert d['config4_stack']['mismatches'] == 0 and d['config4_stack']['all_pages_present']
    assert d['rccl_ok'] is True and 'gauss_fix' not in d['kernels'] and 'gauss_redo' in d['kernels']      # radius 2: the fix rides inside the blur
    for cfg, unit, least, pages in (('c3', 'pages/s', 5000, 4), ('c3gray', 'GB/s', 1500, 8), ('c5', 'pages/s', 1900, 2)):
        with open(os.path.join(ROOT, 'profiles', 'r05_bench_%s.json' % cfg)) as f:
            x = json.loads(f.read().strip().splitlines()[-1])
        assert x['unit'] == unit and x['value'] > least and x['parity']['mismatches'] == 0 and x['parity']['pages_checked'] == pages
        assert x['cpu_baseline'] is None and x['head'] and x['roofline']['bound'] == 'hbm'
    with open(os.path.join(ROOT, 'profiles', 'r05_bench_gpus8_on_one_gpu.json')) as f:
        g = json.loads(f.read().strip().splitlines()[-1])
    assert g['n_gpus'] == 8 and g['control_plane_ranks'] == 8 and g['rccl_ok'] is False
    assert g['config4_stack']['all_pages_present'] and g['config4_stack']['mismatches'] == 0 and g['parity']['mismatches'] == 0
    assert g['parity']['pages_checked'] == 128


def test_several_ranks_without_rccl_is_a_failing_run_unless_it_is_a_rehearsal():
    """make_comm falls back to files when RCCL cannot come up; the line says so, and the run now
    exits 3 instead of 0 -- except for rehearsals that cannot have RCCL (ranks sharing a GPU, the gloo hook)."""
    import bench
    assert bench.non_rccl_is_fatal(8, 'files (RCCL bootstrap failed on rank 3)', env={})
    assert bench.non_rccl_is_fatal(2, 'gloo (test hook)', env={})
    assert not bench.non_rccl_is_fatal(8, 'files (...)', env={'MRCHIP_BENCH_ALLOW_NON_RCCL': '1'})
    assert not bench.non_rccl_is_fatal(8, 'rccl (8 ranks, librccl.so.1)', env={})
    assert not bench.non_rccl_is_fatal(1, 'none (one rank)', env={})
    src = open(bench.__file__).read()
    assert 'sys.exit(3)' in src and 'non_rccl_is_fatal(world, transport)' in src


def test_round6_bench_lines():
    """The lines of round 6 (profiles/r06_README.md) on the final library sources: contract fields, counter traffic of THIS
    build attached, the duplex ceiling is a ceiling, the all-cores extrapolation of the CPU baseline, parity of every line."""
    with open(os.path.join(ROOT, 'profiles', 'r06_bench_c2.json')) as f:
        txt = f.read().strip()
    assert len(txt.splitlines()) == 1                      # stdout carries the one line only
    d = json.loads(txt)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'pipeline_frac', 'head', 'srchash'):
        assert key in d, key
    assert d['value'] > 10000 and d['unit'] == 'pages/s' and d['vs_baseline'] is None and d['dtype'] == 'u8' and d['n_gpus'] == 1
    assert d['head'] and not d['head'].endswith('+') and d['srchash']
    r = d['roofline']
    assert r['kernel'] == 'optimise_rgb' and r['bound'] == 'hbm' and abs(r['frac'] - r['achieved'] / 8000.0) < 1e-4
    assert r['traffic'] and 1.2 < r['traffic'] / r['alg_bytes_per_launch'] < 1.4 and r['traffic_source']['kernel_sources_equal'] is True
    assert r['traffic_source']['file'].startswith('profiles/r06_')
    e = d['e2e']
    assert e['frac_of_duplex_ceiling'] <= 1.0 and e['link_measured']['stream_mix_copy_only']['pages_per_s'] > 500
    assert e['duplex_ceiling_pages_per_s_per_gpu'] >= e['duplex_symmetric_pages_per_s_per_gpu']
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and not c.get('error') and c['memory_capped'] is True
    x = c['all_cores_extrapolated']
    assert x['measured_on_cores'] == c['cores'] and x['host_cpus'] == c['host_cpus'] and x['linear_from_measured'] > c['value']
    assert d['parity']['mismatches'] == 0 and d['parity']['pages_checked'] == 16
    assert d['config4_stack']['mismatches'] == 0 and d['config4_stack']['all_pages_present'] and d['rccl_ok'] is True
    for cfg, unit, least, pages in (('c3', 'pages/s', 5000, 4), ('c3gray', 'GB/s', 1500, 8), ('c5', 'pages/s', 1900, 2)):
        with open(os.path.join(ROOT, 'profiles', 'r06_bench_%s.json' % cfg)) as f:
            y = json.loads(f.read().strip().splitlines()[-1])
        assert y['unit'] == unit and y['value'] > least and y['parity']['mismatches'] == 0 and y['parity']['pages_checked'] == pages
        assert y['srchash'] == d['srchash']
    with open(os.path.join(ROOT, 'profiles', 'r06_bench_gpus8_on_one_gpu.json')) as f:
        g = json.loads(f.read().strip().splitlines()[-1])
    assert g['n_gpus'] == 8 and g['control_plane_ranks'] == 8 and g['rccl_ok'] is False
    assert g['config4_stack']['all_pages_present'] and g['config4_stack']['mismatches'] == 0 and g['parity']['mismatches'] == 0
    assert g['parity']['pages_checked'] == 128
    # the committed counter profiles are of THESE library sources (csrc/ + include/mrchip.h): a later edit of a kernel or of the
    # header must come with new profiles, or the driver's line loses `roofline.traffic`
    import bench
    for tag in ('r06', 'r06_inflight1', 'r06_c3gray'):
        with open(os.path.join(ROOT, 'profiles', tag + '_pmc_summary.json')) as f:
            assert bench.same_kernel_sources(json.load(f)['srchash']), tag
    t, src = bench.pmc_traffic('optimise_rgb', 21.504e9)
    assert t and src['kernel_sources_equal'] is True
