"""Multi-GPU use of the engine: one process per GPU, contiguous shards, results gathered once.

Every (scalar, point) pair is independent, so the path shards with no exchange step (SURVEY.md
section 8e): rank g owns elements [g*n/G, (g+1)*n/G).  The only collective is an optional gather
of results over `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" on CPU in
the tests).  xGMI is point-to-point, so a gather to one rank lands the peers' shards over distinct
links in parallel; nothing here is ring-shaped.
"""
import os


def shard_bounds(n, rank, world):
    """Contiguous slice [lo, hi) of an n-element batch owned by `rank` of `world`."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return (rank * n) // world, ((rank + 1) * n) // world


def env_rank():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (defaults 0,0,1)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend=None):
    """Initialise torch.distributed from the launcher's environment; no-op for a single process."""
    import torch.distributed as dist
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend or "nccl", rank=rank, world_size=world)
    return rank, local_rank, world


def gather_rows(local, n_total, dst=0, group=None):
    """Gather row-shards (torch tensors, equal trailing shape) into one (n_total, ...) tensor on `dst`.

    Shards may differ by one row (n not divisible by the world size): every rank pads to the largest
    shard for the collective and `dst` trims.  Returns the full tensor on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return local                          # (a group of ONE rank still goes through the collective: same code, same result)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if dist.get_backend(group) == "gloo" and local.is_cuda:
        local = local.cpu()                   # rehearsals on one GPU: gloo gathers host tensors only
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    biggest = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)


def sharded_map(fn, arrays, n_total, group=None):
    """Apply `fn(*shard_of_each_array)` to this rank's contiguous shard of host arrays."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_bounds(n_total, rank, world)
    return fn(*[a[lo:hi] for a in arrays])
