"""CPU suite: the N>1 logic of bench.py (page sharding, descriptor broadcast, max-over-ranks
timing) with world_size-2 gloo processes -- no GPU, no pixels across ranks."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    import torch, torch.distributed as dist
    sys.path.insert(0, os.path.join(%r, 'archive-pdf-tools_amd'))
    from mrchip import dist as mdist
    dist.init_process_group('gloo')
    comm = mdist.TorchComm(dist)
    rank, world = comm.rank, comm.world
    # rank 0 owns the work queue; every rank derives its own shard from the broadcast descriptor
    desc = comm.bcast_obj({'w': 4000, 'h': 3000, 'pages': 10, 'seed0': 202, 'boxes': [[1, 2, 30, 40]] * 3} if rank == 0 else None)
    mine = mdist.shard_pages(desc['pages'], rank, world)
    elapsed = comm.max_f64(1.0 + rank)                          # slowest rank defines the step time
    parts = comm.allgather_obj([{'page': p, 'rank': rank, 'pad': 'x' * (rank * 7)} for p in mine])
    gathered = [r for part in parts for r in part]
    if rank == 0:
        print(json.dumps({'desc': desc, 'elapsed': elapsed, 'records': gathered}))
    comm.barrier()
    dist.destroy_process_group()
''')


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sharding_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % ROOT)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS='1')
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['desc'] == {'w': 4000, 'h': 3000, 'pages': 10, 'seed0': 202, 'boxes': [[1, 2, 30, 40]] * 3}
    assert out['elapsed'] == 2.0
    pages = sorted(rec['page'] for rec in out['records'])
    assert pages == list(range(10))                              # every page exactly once
    for rec in out['records']:
        assert rec['rank'] == rec['page'] % 2                    # round-robin page i -> rank i mod G


def test_shard_pages_properties():
    sys.path.insert(0, os.path.join(ROOT, 'archive-pdf-tools_amd'))
    from mrchip import dist as mdist
    for world in (1, 2, 4, 8):
        for n in (0, 1, 7, 8, 512):
            shards = [mdist.shard_pages(n, r, world) for r in range(world)]
            assert sorted(p for s in shards for p in s) == list(range(n))
            assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1


def test_solo_comm_and_rendezvous_name():
    sys.path.insert(0, os.path.join(ROOT, 'archive-pdf-tools_amd'))
    from mrchip import dist as mdist
    c = mdist.SoloComm()
    assert c.bcast_obj({'a': [1, 2]}) == {'a': [1, 2]}
    assert c.allgather_obj({'r': 0}) == [{'r': 0}]
    assert c.max_f64(3.5) == 3.5
    c.barrier()
    p = mdist.rendezvous_path()
    assert os.path.basename(p).startswith('mrchip_rccl_id_')


FILE_WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, os.path.join(%r, 'archive-pdf-tools_amd'))
    from mrchip import dist as mdist
    rank, world, base, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    if mode == 'file':
        comm, how = mdist.FileComm(rank, world, base=base, timeout=60), 'files'
    else:
        # no GPU here: RCCL cannot make a communicator, every rank has to come out on the file transport
        os.environ['MRCHIP_RENDEZVOUS'] = base
        class Ctx: handle = None
        comm, how = mdist.make_comm(Ctx(), rank, world, timeout=20)
    desc = comm.bcast_obj({'pages': 11, 'note': 'x' * 300} if rank == 0 else None)
    mine = mdist.shard_pages(desc['pages'], rank, world)
    for rep in range(5):                                   # several rounds: files of old exchanges are removed on the way
        elapsed = comm.max_f64(1.0 + rank + rep)
    parts = comm.allgather_obj([{'page': p, 'rank': rank, 'pad': 'y' * (rank * 5)} for p in mine])
    other = comm.bcast_obj({'from': world - 1} if rank == world - 1 else None, root=world - 1)
    comm.barrier()
    comm.close()
    if rank == 0:
        print(json.dumps({'how': how, 'desc_pages': desc['pages'], 'elapsed': elapsed, 'other': other,
                          'records': [r for part in parts for r in part]}))
''')


def _run_file_ranks(tmp_path, world, mode):
    import json
    script = tmp_path / 'fworker.py'
    script.write_text(FILE_WORKER % ROOT)
    base = str(tmp_path / 'fc')
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), base, mode], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    out = json.loads([l for l in outs[0][0].splitlines() if l.startswith('{')][-1])
    assert out['desc_pages'] == 11 and out['other'] == {'from': world - 1}
    assert out['elapsed'] == 1.0 + (world - 1) + 4
    assert sorted(r['page'] for r in out['records']) == list(range(11))
    assert all(r['rank'] == r['page'] % world for r in out['records'])
    left = [f for f in os.listdir(tmp_path) if f.startswith('fc')]
    assert len(left) <= 2 * world, left                      # only the last barrier's files may remain
    return out


def test_file_comm_three_ranks(tmp_path):
    assert _run_file_ranks(tmp_path, 3, 'file')['how'] == 'files'


def test_make_comm_falls_back_to_files_when_rccl_cannot_start(tmp_path):
    how = _run_file_ranks(tmp_path, 2, 'auto')['how']
    assert how.startswith('files (RCCL unavailable'), how


BLOCK_WORKER = textwrap.dedent('''
    import os, sys, json, threading
    sys.path.insert(0, os.path.join(%r, 'archive-pdf-tools_amd'))
    from mrchip import dist as mdist
    rank, world, base, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ['MRCHIP_RENDEZVOUS'] = base
    if mode == 'require':
        os.environ['MRCHIP_REQUIRE_RCCL'] = '1'
    class Ctx: handle = None
    def stuck():                     # what ncclCommInitRank does when a peer never joins: prints, then never returns
        os.write(1, b'RCCL banner on fd 1\\n')
        threading.Event().wait()
    try:
        comm, how = mdist.make_comm(Ctx(), rank, world, timeout=2, comm_factory=stuck)
    except mdist.RcclUnavailable as e:
        print(json.dumps({'raised': str(e)}))
        sys.exit(3)
    comm.barrier()
    print(json.dumps({'how': how, 'rank': rank}))        # the result line must still reach the real stdout
    sys.stdout.flush()
    os._exit(0)                      # the stuck helper thread is a daemon; leave without joining it
''')


def _run_block_ranks(tmp_path, mode):
    script = tmp_path / 'bworker.py'
    script.write_text(BLOCK_WORKER % ROOT)
    base = str(tmp_path / 'bc')
    procs = [subprocess.Popen([sys.executable, str(script), str(r), '2', base, mode], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    return procs, [p.communicate(timeout=120) for p in procs]


def test_make_comm_keeps_stdout_when_rccl_init_never_returns(tmp_path):
    """the helper thread stuck in ncclCommInitRank must not leave fd 1 pointing at stderr."""
    import json
    procs, outs = _run_block_ranks(tmp_path, 'auto')
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, se[-2000:]
        lines = [l for l in so.splitlines() if l.startswith('{')]
        assert lines, 'rank %d: result line missing from stdout; stderr: %s' % (r, se[-500:])
        out = json.loads(lines[-1])
        assert out['how'].startswith('files (RCCL unavailable') and 'did not return' in out['how']
        assert 'RCCL banner' not in so and 'RCCL banner' in se          # the banner went to stderr, the result did not
        assert 'RCCL UNAVAILABLE' in se                                  # and the fallback is loud


def test_make_comm_require_rccl_raises_on_every_rank(tmp_path):
    procs, outs = _run_block_ranks(tmp_path, 'require')
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 3, (so, se[-1000:])
        assert 'MRCHIP_REQUIRE_RCCL' in so


def test_native_stdout_banner_goes_to_stderr_even_when_buffered():
    """What native code printf's inside the redirection (RCCL's version banner) must not reach the real stdout later:
    with stdout a pipe C stdio holds it in its buffer until exit, i.e. until after bench.py's JSON line, unless the
    redirection flushes it while fd 1 still points at stderr."""
    import subprocess
    import sys
    code = (
        "import sys, ctypes\n"
        "sys.path.insert(0, %r)\n"
        "from mrchip import dist\n"
        "libc = ctypes.CDLL(None)\n"
        "with dist._stdout_to_stderr(True):\n"
        "    libc.printf(b'BANNER from native code\\n')\n"
        "print('{\"the\": \"line\"}')\n"
    ) % os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'archive-pdf-tools_amd')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]
    assert r.stdout.strip() == '{"the": "line"}', r.stdout
    assert 'BANNER from native code' in r.stderr
