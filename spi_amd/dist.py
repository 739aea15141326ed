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


def device_index(local_rank):
    """Index of this rank's GPU among the VISIBLE devices.  `HIP_VISIBLE_DEVICES` / `ROCR_VISIBLE_DEVICES` are honoured by the runtime itself
    (indices are relative to the visible set); a launcher that shows every rank exactly one device (HIP_VISIBLE_DEVICES=<rank>) makes that
    device index 0 for everybody.  (The reference overwrites CUDA_VISIBLE_DEVICES with '0', run_inversion.py:109 -- one process per job.)"""
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    return local_rank % n if n else 0


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        a, _, b = part.partition('-')
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def _gpu_numa_cpus(index):
    """host CPUs of the NUMA node the GPU with this visible index hangs off (sysfs), or None"""
    try:
        p = torch.cuda.get_device_properties(index)
        bdf = '%04x:%02x:%02x.0' % (getattr(p, 'pci_domain_id', 0), p.pci_bus_id, p.pci_device_id)
        node = int(open(f'/sys/bus/pci/devices/{bdf}/numa_node').read())
        if node < 0:
            return None
        return _parse_cpulist(open(f'/sys/devices/system/node/node{node}/cpulist').read())
    except Exception:                                            # noqa: BLE001  (no sysfs, no GPU: fall back to an even split)
        return None


def plan_affinity(local_world, allowed, numa_cpus=None):
    """Disjoint CPU sets for the `local_world` ranks of this host -> list of sorted lists.  Ranks whose GPUs share a NUMA node split THAT node's
    CPUs (numa_cpus[r] = set or None); ranks without NUMA information split what is left evenly.  Pure function (tested on CPU)."""
    allowed = sorted(allowed)
    numa_cpus = numa_cpus or [None] * local_world
    groups = {}
    for r in range(local_world):
        key = tuple(sorted(numa_cpus[r] & set(allowed))) if numa_cpus[r] else None
        groups.setdefault(key if key else None, []).append(r)
    plan = [None] * local_world
    taken = set()
    for key, ranks in groups.items():
        if key is None:
            continue
        per = max(1, len(key) // len(ranks))
        for j, r in enumerate(ranks):
            plan[r] = list(key[j * per:(j + 1) * per]) or [key[j % len(key)]]
            taken.update(plan[r])
    rest = [c for c in allowed if c not in taken] or allowed
    ranks = groups.get(None, [])
    if ranks:
        per = max(1, len(rest) // len(ranks))
        for j, r in enumerate(ranks):
            plan[r] = rest[j * per:(j + 1) * per] or [rest[j % len(rest)]]
    return plan


def local_world_size(world):
    """Ranks on THIS host: LOCAL_WORLD_SIZE when the launcher exports it (torchrun does), else the GPUs this process sees (a multi-node run
    without torchrun must not split the host's cores by the GLOBAL world size), at most `world`."""
    v = os.environ.get('LOCAL_WORLD_SIZE')
    if v is not None:
        return int(v)
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    return max(1, min(world, n)) if n else world


def pin_rank_affinity(local_rank, local_world):
    """One process per GPU on one host: give every rank its own host cores, next to its GPU.  An eager stage-2 iteration needs ~25 ms of
    single-thread launch work per ~25 ms of GPU time, so ranks that share cores (or sit on the far socket) lose throughput to each other;
    8 ranks with one all-core OpenMP pool each would also oversubscribe the host.  No-op for a single-rank run, on platforms without
    sched_setaffinity, and when SPI_PIN_AFFINITY=0; an even split of the allowed cores when the process cannot see all local GPUs.  -> the CPU list this rank now holds (or None)."""
    if local_world <= 1 or os.environ.get('SPI_PIN_AFFINITY', '1') == '0' or not hasattr(os, 'sched_setaffinity'):
        return None
    allowed = os.sched_getaffinity(0)
    # NUMA placement needs every local rank's PHYSICAL device.  Under a launcher that shows each rank only its own GPU (HIP_VISIBLE_DEVICES=<rank>)
    # device_index(r) is 0 for every r: all ranks would believe they share this rank's node and take 1 / local_world of it.  Then: even split.
    sees_all = torch.cuda.is_available() and torch.cuda.device_count() >= local_world
    numa = [_gpu_numa_cpus(device_index(r)) for r in range(local_world)] if sees_all else None
    mine = plan_affinity(local_world, allowed, numa)[local_rank]
    try:
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(len(mine), int(os.environ.get('OMP_NUM_THREADS', len(mine))))))
    except OSError:
        return None
    return mine


def reserve_allocator_pool(device, gib=None):
    """Start-up step of a long-running inversion process: allocate and release one large block so that it stays in torch's caching allocator
    and later requests are carved out of it.  The workspace sizes of the masked pseudo-view branches depend on each iteration's random cameras;
    without the pool a later iteration can ask for a block size the allocator has not seen, and a first-time hipMalloc of GBs is a host-side
    stall in the middle of the loop (288 GB of HBM per GPU: 24 GiB is < 10 %).  -> GiB reserved."""
    gib = int(os.environ.get('SPI_POOL_GIB', '24')) if gib is None else gib
    try:
        free_b, _ = torch.cuda.mem_get_info(device)
        gib = int(min(gib, free_b / 2 ** 30 * 0.25))
        if gib > 0:
            block = torch.empty(gib << 30, dtype=torch.uint8, device=device)
            del block
        return max(gib, 0)
    except Exception:                                            # noqa: BLE001  (a failed reservation only loses the protection)
        return 0


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
