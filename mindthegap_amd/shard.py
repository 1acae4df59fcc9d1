"""Multi-GPU sharding of the fill hot path: sites (or seeds) are independent (Dispatcher.iterate, src/Filler.cpp:824,844), so
each rank takes a contiguous slice, the index is replicated, and only the results travel: byte payloads are gathered on one
rank with an all_gather of sizes followed by a padded gather (RCCL over xGMI with backend "nccl", gloo in the CPU tests)."""
import numpy as np


def shard_range(n_items, rank, world):
    """contiguous slice [lo, hi) of n_items for this rank; slices differ by at most one item"""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_bytes(payload, dst=0, device=None):
    """payload: 1-D uint8 numpy array of this rank.  Returns the list of all ranks' payloads on rank dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    mine = torch.from_numpy(np.ascontiguousarray(payload, dtype=np.uint8)).to(dev)
    sz = torch.tensor([mine.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(sz) for _ in range(world)]
    dist.all_gather(sizes, sz)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=torch.uint8, device=dev)
    pad[: mine.numel()] = mine
    bufs = [torch.zeros(mx, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return [b[:n].cpu().numpy() for b, n in zip(bufs, sizes)]
