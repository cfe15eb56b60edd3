"""Instance sharding across the GPUs of one node: instances never interact, so every rank rolls out a
contiguous block with no data-path collective; the only exchange is the gather of results to rank 0 (RCCL over xGMI on
GPUs, gloo on CPU for tests).  One process per GPU, torch.distributed for the plumbing."""
import os
import sys

import numpy as np


def shard_bounds(n_total, rank, world):
    """contiguous block [lo, hi) of rank `rank`: sizes differ by at most one, earlier ranks take the remainder"""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(array, rank, world):
    lo, hi = shard_bounds(len(array), rank, world)
    return array[lo:hi]


def init_from_env(backend=None):
    """torch.distributed init from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun); returns (rank, world, local_rank)"""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            try:
                dist.init_process_group(backend="nccl", rank=rank, world_size=world)
                probe = torch.zeros(1, device=torch.device("cuda", local))
                dist.all_reduce(probe)           # RCCL creates its communicator lazily: fail here, not inside the timed region
                torch.cuda.synchronize()
            except Exception as e:               # the instances do not need the collective; the final-state gather can run on host buffers
                sys.stderr.write("cclqr.dist: RCCL unavailable (%s); gathering through gloo on host buffers\n" % (e,))
                if dist.is_initialized():
                    dist.destroy_process_group()
                os.environ["MASTER_PORT"] = str(int(os.environ["MASTER_PORT"]) + 1)
                dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def gather_to_root(local, n_total, rank, world, dst=0):
    """gather per-rank blocks (torch tensors [n_local, ...]) into [n_total, ...] on `dst` (None elsewhere).
    Blocks are padded to the largest shard so that ONE fixed-size gather moves everything: each peer sends its block
    straight to the root over its own link instead of circulating a ring."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local
    sizes = [shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)]
    nmax = max(sizes)
    pad = local
    if local.shape[0] < nmax:
        pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
    pad = pad.contiguous()
    dev = pad.device
    if dist.get_backend() == "gloo" and pad.is_cuda:
        pad = pad.cpu()          # gloo rehearsal of the GPU path: the collective itself runs on host buffers
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([bufs[r][:sizes[r]] for r in range(world)], dim=0).to(dev)


def max_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=None if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sharded_rollout_numpy(rollout_fn, z0, rank, world):
    """run `rollout_fn(block) -> zT_block` on this rank's block of z0 and gather the final states on rank 0 (numpy in/out)"""
    import torch
    block = shard(z0, rank, world)
    zT = rollout_fn(block)
    out = gather_to_root(torch.from_numpy(np.ascontiguousarray(zT)), len(z0), rank, world)
    return None if out is None else out.numpy()
