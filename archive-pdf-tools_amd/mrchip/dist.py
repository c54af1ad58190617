"""Page sharding across GPUs: pages are independent, one process per GPU,
static round-robin `page i -> rank i mod G`.  The only communication is control data --
the work-queue descriptor from rank 0, per-page result records back, and the max of the
elapsed time -- over torch.distributed (RCCL on the GPU node, gloo in the CPU tests).
Pixels never cross ranks."""
import json


def shard_pages(n_pages, rank, world):
    """Indices of the pages rank `rank` of `world` processes."""
    return list(range(rank, n_pages, world))


def _device(dist):
    import torch
    return torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')


def broadcast_descriptor(dist, desc):
    """rank 0 passes a small JSON-able dict; every rank returns it (bytes, not pixels)."""
    import torch
    dev = _device(dist)
    if dist.get_rank() == 0:
        raw = json.dumps(desc).encode()
        n = torch.tensor([len(raw)], dtype=torch.int64, device=dev)
    else:
        n = torch.zeros(1, dtype=torch.int64, device=dev)
    dist.broadcast(n, 0)
    buf = torch.zeros(int(n.item()), dtype=torch.uint8, device=dev)
    if dist.get_rank() == 0:
        buf.copy_(torch.tensor(list(raw), dtype=torch.uint8))
    dist.broadcast(buf, 0)
    return json.loads(bytes(buf.cpu().tolist()).decode())


def max_over_ranks(dist, seconds):
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_records(dist, records):
    """Per-page result records (digests, counts) to every rank; returns the flat list."""
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, records)
    return [r for part in out for r in part]
