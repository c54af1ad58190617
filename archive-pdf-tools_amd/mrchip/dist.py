"""Page sharding across GPUs: pages are independent, one process per GPU, static round-robin
`page i -> rank i mod G`.  The only communication is control data -- the work-queue descriptor from rank 0,
per-page result records back, the maximum of the elapsed times.  Pixels never cross ranks.

Two transports with one interface (bcast_obj / allgather_obj / max_f64 / barrier):
  RcclComm   the native one: libmrchip's mrchip_comm_* calls, i.e. RCCL over xGMI through ctypes, no PyTorch.
             The 128-byte RCCL unique id goes from rank 0 to the others through a file (one node).
  TorchComm  torch.distributed with a CPU backend (gloo): the world_size-2 CPU tests of the sharding logic.
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


class RcclComm(_Comm):
    """RCCL through libmrchip (mrchip_comm_*).  `ctx`: the rank's mrchip Context (its device is the rank's GPU)."""

    def __init__(self, ctx, rank, world, rendezvous=None, timeout=120.0):
        from . import _lib
        self._lib = _lib
        self.lib = _lib.load()
        self.rank, self.world = rank, world
        path = rendezvous or rendezvous_path()
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            sys.stdout.flush()
            saved = os.dup(1)
            try:
                os.dup2(2, 1)
                _lib.check(self.lib.mrchip_comm_unique_id(ident), 'mrchip_comm_unique_id')
            finally:
                os.dup2(saved, 1)
                os.close(saved)
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
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            self._h = self.lib.mrchip_comm_init(ctx.handle, rank, world, ident)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
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
