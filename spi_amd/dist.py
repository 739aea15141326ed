"""Multi-GPU plumbing: one process per MI355X, images sharded across ranks, ONE small all-reduce of
throughput statistics over RCCL/xGMI.  There is no data-path collective: every image's
inversion is independent (its own G, optimiser state and W+), exactly like the reference's
``--dataset_block i/N`` processes (spi/data/images_dataset.py:149-158).
"""
import os
import torch
import torch.distributed as dist


def init_from_env(backend=None, timeout_s=900):
    """-> (rank, world_size, local_rank).  Initialises torch.distributed when WORLD_SIZE > 1.  ``timeout_s`` bounds every collective:
    a rank that died cannot hang the survivors' final reduce for longer than that (the launcher kills them earlier)."""
    import datetime
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'      # 'nccl' is RCCL on ROCm
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
    return rank, world, local


def shard_indices(n_items, rank, world_size, mode='block'):
    """Indices of the items this rank owns.  'block' reproduces the reference's contiguous blocks
    (block = n // world + 1); 'stride' is round-robin."""
    if mode == 'stride':
        return list(range(rank, n_items, world_size))
    block = n_items // world_size + 1
    return list(range(min(rank * block, n_items), min((rank + 1) * block, n_items)))


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def reduce_stats(values, device=None, op='sum'):
    """All-reduce a short list of floats (fp64).  A failed image contributes through its done-count, so
    one rank's failure never blocks the others (the reduce happens once, at the end)."""
    t = torch.tensor(list(values), dtype=torch.float64, device=device if device is not None else 'cpu')
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 'sum' else dist.ReduceOp.MAX)
    return t.cpu().tolist()


def shutdown():
    """Barrier + destroy the process group (no-op in a single-process run)."""
    import torch.distributed as td
    if td.is_available() and td.is_initialized():
        td.barrier()
        td.destroy_process_group()
