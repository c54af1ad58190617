"""The native control plane (mrchip_comm_*: RCCL through libmrchip, no PyTorch) on one GPU: a communicator of one
rank goes through the real ncclCommInitRank / ncclBroadcast / ncclAllGather / ncclAllReduce calls."""
import os

import pytest

from mrchip import _lib, dist as mdist

pytestmark = pytest.mark.gpu


def test_rccl_comm_world_of_one(tmp_path):
    ctx = _lib.default_context()
    comm = mdist.RcclComm(ctx, 0, 1, rendezvous=str(tmp_path / 'id'))
    assert os.path.getsize(tmp_path / 'id') == 128
    table = {'pages': [{'w': 4000, 'h': 3000, 'c': 3, 'n_boxes': 29}] * 5, 'boxes': list(range(400))}
    assert comm.bcast_obj(table) == table
    assert comm.allgather_obj([{'page': 3, 'rank': 0, 'mask_popcount': 780537, 'mask': 'ab' * 32}]) == \
        [[{'page': 3, 'rank': 0, 'mask_popcount': 780537, 'mask': 'ab' * 32}]]
    assert comm.max_f64(2.5) == 2.5
    assert comm.bcast_bytes(b'') == b''
    comm.barrier()
    comm.close()
    assert not os.path.exists(tmp_path / 'id')
