"""Page sharding across GPUs: pages are independent, one process per GPU, static round-robin
`page i -> rank i mod G`.  The only communication is control data -- the work-queue descriptor from rank 0,
per-page result records back, the maximum of the elapsed times.  Pixels never cross ranks.

Two transports with one interface (bcast_obj / allgather_obj / max_f64 / barrier):
  RcclComm   the native one: libmrchip's mrchip_comm_* calls, i.e. RCCL over xGMI through ctypes, no PyTorch.
             The 128-byte RCCL unique id goes from rank 0 to the others through a file (one node).
  TorchComm  torch.distributed with a CPU backend (gloo): the world_size-2 CPU tests of the sharding logic.
  FileComm   files of the node's temporary directory: the fallback make_comm() takes when RCCL cannot make a
             communicator on every rank, so that a multi-GPU run still reports instead of hanging.
"""
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np


def shard_pages(n_pages, rank, world):
    """Indices of the pages rank `rank` of `world` processes."""
    return list(range(rank, n_pages, world))


class _Comm:
    rank = 0
    world = 1

    def bcast_bytes(self, raw, root=0):
        raise NotImplementedError

    def allgather_bytes(self, raw):
        raise NotImplementedError

    def max_f64(self, value):
        raise NotImplementedError

    # ---- objects on top of bytes -------------------------------------------------------------------
    def bcast_obj(self, obj, root=0):
        """rank `root` passes a JSON-able object; every rank returns it (bytes, not pixels)."""
        raw = json.dumps(obj).encode() if self.rank == root else b''
        n = int(self.max_f64(float(len(raw))))
        return json.loads(self.bcast_bytes(raw.ljust(n, b' '), root).decode())

    def allgather_obj(self, obj):
        """every rank passes a JSON-able object; returns the list of all of them in rank order"""
        raw = json.dumps(obj).encode()
        n = int(self.max_f64(float(len(raw))))
        parts = self.allgather_bytes(raw.ljust(n, b' '))
        return [json.loads(p.decode()) for p in parts]

    def barrier(self):
        self.max_f64(0.0)

    def close(self):
        pass


class SoloComm(_Comm):
    """world of one: nothing to exchange"""

    def bcast_bytes(self, raw, root=0):
        return raw

    def allgather_bytes(self, raw):
        return [raw]

    def max_f64(self, value):
        return float(value)


class _stdout_to_stderr:
    """fd 1 -> fd 2 for the duration of the block (native code that prints on stdout), restored on exit"""

    def __init__(self, on=True):
        self.on, self.saved = on, None

    def __enter__(self):
        if self.on:
            sys.stdout.flush()
            self.saved = os.dup(1)
            os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        if self.saved is not None:
            # what the native code printf'ed sits in C stdio's buffer when stdout is a pipe or a file: push it out
            # while fd 1 still points at stderr, or it lands on the real stdout at exit (after bench.py's JSON line)
            try:
                C.CDLL(None).fflush(None)
            except OSError:
                pass
            os.dup2(self.saved, 1)
            os.close(self.saved)
            self.saved = None
        return False


class RcclComm(_Comm):
    """RCCL through libmrchip (mrchip_comm_*).  `ctx`: the rank's mrchip Context (its device is the rank's GPU)."""

    def __init__(self, ctx, rank, world, rendezvous=None, timeout=120.0, redirect_stdout=True):
        """redirect_stdout: point fd 1 at stderr around the two RCCL calls that print a banner.  Only for a caller that
        runs this constructor synchronously: make_comm() runs it on a helper thread that may never return, so it does
        the redirection itself (around the join, restored unconditionally) and passes False."""
        from . import _lib
        self._lib = _lib
        self.lib = _lib.load()
        self.rank, self.world = rank, world
        path = rendezvous or rendezvous_path()
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            with _stdout_to_stderr(redirect_stdout):
                _lib.check(self.lib.mrchip_comm_unique_id(ident), 'mrchip_comm_unique_id')
            tmp = path + '.tmp'
            with open(tmp, 'wb') as f:
                f.write(bytes(ident))
            os.replace(tmp, path)                  # atomic: readers see all 128 bytes or no file
        else:
            t0 = time.time()
            while not os.path.exists(path):
                if time.time() - t0 > timeout:
                    raise _lib.MrchipError('RCCL rendezvous: %s did not appear within %.0f s' % (path, timeout))
                time.sleep(0.01)
            with open(path, 'rb') as f:
                raw = f.read()
            C.memmove(ident, raw, 128)
        # RCCL prints a version banner on stdout when a communicator is made; a benchmark's stdout is its result line
        with _stdout_to_stderr(redirect_stdout):
            self._h = self.lib.mrchip_comm_init(ctx.handle, rank, world, ident)
        if not self._h:
            raise _lib.MrchipError('mrchip_comm_init: %s' % _lib.last_error())
        self._path = path

    def bcast_bytes(self, raw, root=0):
        buf = np.frombuffer(bytearray(raw), dtype=np.uint8)
        if buf.size:
            self._lib.check(self.lib.mrchip_comm_bcast(self._h, buf.ctypes.data, buf.size, root), 'mrchip_comm_bcast')
        return buf.tobytes()

    def allgather_bytes(self, raw):
        send = np.frombuffer(bytearray(raw), dtype=np.uint8)
        recv = np.zeros(send.size * self.world, dtype=np.uint8)
        if send.size:
            self._lib.check(self.lib.mrchip_comm_allgather(self._h, send.ctypes.data, send.size, recv.ctypes.data),
                            'mrchip_comm_allgather')
        return [recv[i * send.size:(i + 1) * send.size].tobytes() for i in range(self.world)]

    def max_f64(self, value):
        v = np.array([float(value)], dtype=np.float64)
        self._lib.check(self.lib.mrchip_comm_allreduce_f64(self._h, v.ctypes.data_as(C.POINTER(C.c_double)), 1, 1),
                        'mrchip_comm_allreduce_f64')
        return float(v[0])

    def close(self):
        if self._h:
            self.barrier()
            self.lib.mrchip_comm_destroy(self._h)
            self._h = None
            if self.rank == 0:
                try:
                    os.remove(self._path)
                except OSError:
                    pass


class FileComm(_Comm):
    """Control data through files of one node's temporary directory: the fallback when RCCL cannot make a
    communicator (no usable network interface for its bootstrap, a rank without librccl ...), and the transport over
    which the ranks agree whether RCCL came up everywhere.  Exchange number k of rank r is the file `<base>.<k>.<r>`
    (written under a temporary name and renamed: readers see all of it or nothing).  A rank removes a file of its own
    once it has COMPLETED a later exchange in which every rank writes: a rank writes there only after it is through
    with everything before, so by then every rank has read the file."""

    def __init__(self, rank, world, base=None, timeout=600.0):
        self.rank, self.world = rank, world
        self.base = base or (rendezvous_path() + '_fc')
        self.timeout = timeout
        self.k = 0
        self.mine = []              # exchange numbers of this rank's files still on disk
        self.full_done = -1         # last completed exchange in which every rank wrote

    def _name(self, k, r):
        return '%s.%d.%d' % (self.base, k, r)

    def _exchange(self, raw, writers):
        k = self.k
        self.k += 1
        for old in [x for x in self.mine if x < self.full_done]:
            try:
                os.remove(self._name(old, self.rank))
            except OSError:
                pass
            self.mine.remove(old)
        if self.rank in writers:
            path = self._name(k, self.rank)
            with open(path + '.tmp', 'wb') as f:
                f.write(raw)
            os.replace(path + '.tmp', path)
            self.mine.append(k)
        out = {}
        t0 = time.time()
        for r in writers:
            path = self._name(k, r)
            while not os.path.exists(path):
                if time.time() - t0 > self.timeout:
                    raise RuntimeError('FileComm: %s did not appear within %.0f s' % (path, self.timeout))
                time.sleep(0.002)
            with open(path, 'rb') as f:
                out[r] = f.read()
        if len(writers) == self.world:
            self.full_done = k
        return out

    def bcast_bytes(self, raw, root=0):
        return self._exchange(raw, [root])[root]

    def allgather_bytes(self, raw):
        got = self._exchange(raw, list(range(self.world)))
        return [got[r] for r in range(self.world)]

    def max_f64(self, value):
        parts = self.allgather_bytes(np.array([float(value)], dtype=np.float64).tobytes())
        return float(max(np.frombuffer(p, dtype=np.float64)[0] for p in parts))

    def close(self):
        self.barrier()
        self.barrier()              # completes an all-writers exchange after the first one: everything before it may go
        self._exchange(b'', [])     # (removes them; the 8-byte files of the last barrier stay behind)


class RcclUnavailable(RuntimeError):
    """MRCHIP_REQUIRE_RCCL=1 and RCCL did not come up on every rank"""


def make_comm(ctx, rank, world, timeout=90.0, comm_factory=None):
    """The control-plane transport of a multi-rank run: RCCL if every rank gets its communicator, else files.
    ncclCommInitRank blocks until all ranks have joined and cannot be cancelled, so it runs on a helper thread with a
    deadline; the ranks then tell each other (through files) whether it returned, and only if it did everywhere is the
    RCCL communicator used.  Returns (comm, description); the description starts with 'rccl' only when RCCL carries
    the control plane.  The fallback is never silent: it is written to stderr, and with MRCHIP_REQUIRE_RCCL=1 it raises
    RcclUnavailable instead (after the ranks have agreed, so all of them raise).
    comm_factory: stands in for RcclComm in the CPU tests (a stub whose init blocks, or fails)."""
    if world == 1:
        return SoloComm(), 'none (one rank)'
    fc = FileComm(rank, world)
    import threading
    box = {}
    factory = comm_factory or (lambda: RcclComm(ctx, rank, world, redirect_stdout=False))

    def init():
        try:
            box['comm'] = factory()
        except Exception as e:          # noqa: BLE001 - any failure means "no RCCL on this rank"
            box['err'] = '%s: %s' % (type(e).__name__, e)

    # RCCL prints a banner on stdout while the communicator is made; a benchmark's stdout is its result line.  The
    # redirection is process-wide, so it is made and undone HERE, around the join: a helper thread stuck in
    # ncclCommInitRank never reaches a `finally` of its own, and fd 1 would stay on stderr for the rest of the run.
    th = threading.Thread(target=init, daemon=True)
    with _stdout_to_stderr(True):
        th.start()
        th.join(timeout)
    ok = 'comm' in box
    flags = fc.allgather_obj({'ok': ok, 'err': box.get('err', 'ncclCommInitRank did not return within %.0f s' % timeout if not ok else '')})
    if all(f['ok'] for f in flags):
        fc.close()
        return box['comm'], 'rccl'
    why = '; '.join('rank %d: %s' % (i, f['err']) for i, f in enumerate(flags) if not f['ok'])
    if os.environ.get('MRCHIP_REQUIRE_RCCL', '') not in ('', '0'):
        raise RcclUnavailable('MRCHIP_REQUIRE_RCCL is set and RCCL did not come up on every rank -- %s' % why)
    sys.stderr.write('mrchip.dist: rank %d: RCCL UNAVAILABLE, control plane falls back to files -- %s\n' % (rank, why))
    sys.stderr.flush()
    return fc, 'files (RCCL unavailable -- %s)' % why


class TorchComm(_Comm):
    """torch.distributed process group with a CPU backend (gloo): the CPU tests of the N > 1 logic."""

    def __init__(self, dist):
        import torch
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def bcast_bytes(self, raw, root=0):
        t = self.torch.tensor(list(raw), dtype=self.torch.uint8)
        self.dist.broadcast(t, root)
        return bytes(t.tolist())

    def allgather_bytes(self, raw):
        t = self.torch.tensor(list(raw), dtype=self.torch.uint8)
        outs = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return [bytes(o.tolist()) for o in outs]

    def max_f64(self, value):
        t = self.torch.tensor([float(value)], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def rendezvous_path():
    """File through which rank 0 hands the RCCL unique id to the other ranks of this launch (one node).  Named after
    the launcher's MASTER_PORT, run id and process id (the ranks of one launch are children of one launcher process),
    so that neither a concurrent launch nor the leftover of a crashed one can be mistaken for it;
    MRCHIP_RENDEZVOUS overrides the path for launchers that do not fork the ranks from one parent."""
    if os.environ.get('MRCHIP_RENDEZVOUS'):
        return os.environ['MRCHIP_RENDEZVOUS']
    tag = '%s_%s_%d' % (os.environ.get('MASTER_PORT', '0'), os.environ.get('TORCHELASTIC_RUN_ID', 'none'), os.getppid())
    tag = ''.join(ch if ch.isalnum() or ch in '_-' else '_' for ch in tag)
    return os.path.join(tempfile.gettempdir(), 'mrchip_rccl_id_%s_%d' % (tag, os.getuid()))


def page_records(pages, rank, masks, fgs, bgs, digest):
    """Per-page result records gathered on rank 0: {page, rank, mask_popcount, digests}."""
    return [{'page': int(p), 'rank': int(rank), 'mask_popcount': int(np.count_nonzero(m)),
             'mask': digest(m), 'fg': digest(f), 'bg': digest(b)} for p, m, f, b in zip(pages, masks, fgs, bgs)]
