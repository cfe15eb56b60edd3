"""Instance sharding across the GPUs of one node: instances never interact, so every rank rolls out a
contiguous block with no data-path collective; the only exchange is the collection of results on rank 0 -- final states,
and the recorded trajectories (`Storage`, lqr_tracking.jl:32-35) in time chunks that travel while the next chunk is being
computed.  RCCL over xGMI on GPUs, gloo on CPU for tests.  One process per GPU, torch.distributed for the plumbing."""
import os

import numpy as np


def shard_bounds(n_total, rank, world):
    """contiguous block [lo, hi) of rank `rank`: sizes differ by at most one, earlier ranks take the remainder"""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(array, rank, world):
    lo, hi = shard_bounds(len(array), rank, world)
    return array[lo:hi]


def init_from_env(backend=None, force_init=False):
    """torch.distributed init from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun); returns (rank, world, local_rank).

    backend None = "nccl" (RCCL) when a GPU is visible, else "gloo".  There is NO fallback: if RCCL cannot be initialised the
    error propagates and the job ends non-zero on every rank -- a multi-GPU line can never silently be a gloo run.  gloo on a GPU
    box has to be asked for by name (bench.py --allow-gloo / --rehearse-shared-gpu)."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_init) and not dist.is_initialized():      # force_init: a one-rank process group (the RCCL smoke test)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # RCCL's cross-process buffer sharing needs dmabuf IPC on this driver
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world)
            probe = torch.zeros(1, device=torch.device("cuda", local))
            dist.all_reduce(probe)           # RCCL creates its communicator lazily: fail here, not inside the timed region
            torch.cuda.synchronize()
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class RootGather:
    """Fixed-size rooted gather of per-rank blocks [n_local, ...] into [n_total, ...] on rank 0 with every buffer allocated ONCE
    (the padded send block, the N receive blocks and the assembled result): nothing is allocated or concatenated inside a timed
    region.  Blocks are padded to the largest shard so that one gather moves everything: each peer sends its block straight to
    the root over its own link instead of circulating a ring.  `force_collective` makes a single rank call the collective too
    (the RCCL smoke test: one process, world_size 1)."""

    def __init__(self, n_total, tail_shape, dtype, device, rank, world, dst=0, force_collective=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world, self.dst, self.n_total = rank, world, dst, int(n_total)
        self.collective = world > 1 or force_collective
        self.sizes = [shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)]
        self.nmax = max(self.sizes) if self.sizes else 0
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.host_staged = self.collective and dist.is_initialized() and dist.get_backend() == "gloo" and self.device.type == "cuda"
        cdev = torch.device("cpu") if self.host_staged else self.device      # gloo rehearsal of the GPU path: host buffers carry the collective
        shape = (self.nmax,) + tuple(tail_shape)
        self.pad = torch.zeros(shape, dtype=dtype, device=cdev) if self.collective else None
        self.bufs = [torch.empty(shape, dtype=dtype, device=cdev) for _ in range(world)] if (self.collective and rank == dst) else None
        self.out = torch.empty((self.n_total,) + tuple(tail_shape), dtype=dtype, device=self.device) if (self.collective and rank == dst) else None

    def __call__(self, local):
        if not self.collective:
            return local
        assert local.shape[0] == self.sizes[self.rank], "block size does not match this rank's shard"
        self.pad[:local.shape[0]].copy_(local)
        self.dist.gather(self.pad, self.bufs, dst=self.dst)
        if self.rank != self.dst:
            return None
        lo = 0
        for r in range(self.world):
            self.out[lo:lo + self.sizes[r]].copy_(self.bufs[r][:self.sizes[r]], non_blocking=True)
            lo += self.sizes[r]
        return self.out


def gather_to_root(local, n_total, rank, world, dst=0):
    """one-off form of RootGather (numpy helpers, tests): allocates its buffers per call -- timed loops keep a RootGather"""
    if world == 1:
        return local
    return RootGather(n_total, tuple(local.shape[1:]), local.dtype, local.device, rank, world, dst)(local)


class TrajectoryGather:
    """Collection of the recorded trajectories of all ranks on rank 0, in time chunks, overlapped with compute.

    Every rank rolls its `n_local` instances out in `chunks` launches of `T / chunks` steps (the k0 continuation of
    cclqr_rollout_dev) into one of two chunk slabs [n_local][Tc][nb][13].  After the launch of chunk c has been enqueued,
    `submit(c, slab)` enqueues -- on a second stream that waits for that launch only -- ONE rooted gather of the slab (RCCL:
    a grouped send/recv fan-in, so each of the N-1 peers uses its own xGMI link into the root) and, on rank 0, the copies of
    the N received slabs into the final Storage layout out[n_total][T][nb][13].  The compute stream goes on with chunk c+1 and
    only waits (`wait_slab_free`) before it overwrites a slab that is still being sent.

    Cost model: per GPU 104 nb bytes per instance-step leave over one link at <= 153 GB/s; at the measured
    24 M instance-steps/s and nb = 17 that is 42 GB/s per GPU = 27 % of a link, so the transfer hides behind compute except
    for the last chunk: exposed time ~ (bytes of one chunk) / link rate + the root's copies."""

    def __init__(self, rank, world, n_local, T, nb, chunks, device, dtype=None, force_collective=False, n_total=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        assert T % chunks == 0, "sim steps must be a multiple of the number of trajectory chunks"
        self.rank, self.world, self.n, self.T, self.nb, self.H, self.Tc = rank, world, n_local, T, nb, chunks, T // chunks
        # shards: equal blocks of n_local (bench.py: weak scaling), or the contiguous blocks of shard_bounds(n_total) -- sizes then differ by at
        # most one and every slab is padded to the largest, so that ONE fixed-size gather per chunk still moves everything
        self.n_total = int(n_total) if n_total is not None else n_local * world
        self.sizes = [n_local] * world if n_total is None else [shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)]
        assert self.sizes[rank] == n_local, "n_local does not match this rank's shard of n_total"
        self.nmax = max(self.sizes)
        self.device = device
        dtype = dtype or torch.float64
        self.cuda = device is not None and torch.device(device).type == "cuda"
        # force_collective: a single rank (world_size 1, process group initialised) really calls dist.gather -- the RCCL smoke test
        self.collective = world > 1 or force_collective
        self.gloo_on_gpu = self.cuda and self.collective and dist.get_backend() == "gloo"
        self.slabs = [torch.empty((self.nmax, self.Tc, nb, 13), dtype=dtype, device=device) for _ in range(2)]
        if self.nmax > n_local:
            for sl in self.slabs:
                sl[n_local:].zero_()          # the padding rows travel with every gather: defined bits, written once
        self.out = torch.empty((self.n_total, T, nb, 13), dtype=dtype, device=device) if rank == 0 else None
        self.recv = None
        if rank == 0 and self.collective:
            rdev = "cpu" if self.gloo_on_gpu else device
            self.recv = [[torch.empty((self.nmax, self.Tc, nb, 13), dtype=dtype, device=rdev) for _ in range(world)] for _ in range(2)]
        self.comm = torch.cuda.Stream(device=device) if self.cuda else None
        self.free_ev = [None, None]
        self.bytes_gathered = 0

    def slab(self, c):
        """chunk c's slab: rows [0, n_local) are this rank's instances (the rollout launch writes them), the rest is padding"""
        return self.slabs[c % 2]

    def wait_slab_free(self, c):
        """the compute stream waits until the gather that last used chunk c's slab has finished"""
        ev = self.free_ev[c % 2]
        if ev is not None and self.cuda:
            self.torch.cuda.current_stream(self.device).wait_event(ev)

    def submit(self, c):
        """enqueue the collection of chunk c (its rollout launch is already enqueued on the current stream)"""
        torch, dist = self.torch, self.dist
        slab = self.slabs[c % 2]
        t0, t1 = c * self.Tc, (c + 1) * self.Tc
        if self.cuda:
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))
            self.comm.wait_event(done)
            ctx = torch.cuda.stream(self.comm)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            if not self.collective:
                self.out[:, t0:t1].copy_(slab[:self.n])
            else:
                send = slab.cpu() if self.gloo_on_gpu else slab
                bufs = self.recv[c % 2] if self.rank == 0 else None
                dist.gather(send, bufs, dst=0)
                self.bytes_gathered += slab.numel() * slab.element_size() * (self.world - 1) if self.rank == 0 else 0
                if self.rank == 0:
                    lo = 0
                    for r in range(self.world):
                        self.out[lo:lo + self.sizes[r], t0:t1].copy_(bufs[r][:self.sizes[r]], non_blocking=True)
                        lo += self.sizes[r]
            if self.cuda:
                ev = torch.cuda.Event()
                ev.record(self.comm)
                self.free_ev[c % 2] = ev

    def finish(self):
        """the current stream waits for every outstanding collection; returns out [n_total][T][nb][13] on rank 0 (None elsewhere)"""
        if self.cuda:
            self.torch.cuda.current_stream(self.device).wait_stream(self.comm)
        return self.out


def collection_plan(rank, world, n_local, T, nb, chunks, record, collect, n_total=None, itemsize=8):
    """Device bytes this rank allocates for one bench rollout and its collection, by purpose (bench.py prints rank 0's plan and refuses to
    start when it exceeds the free HBM: the first 8-GPU run must fail with a sentence, not with an allocator error inside the timed region).
    collect: "trajectory" = the recorded trajectories of all ranks are assembled on rank 0 (TrajectoryGather), "final" = only the final
    states travel (RootGather) and every rank keeps its own trajectory."""
    n_total = n_local * world if n_total is None else int(n_total)
    sizes = [shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)]
    nmax = max(sizes) if sizes else 0
    row = nb * 13 * itemsize
    plan = {"initial + final states, status, multipliers": (2 * n_local * row + 4 * n_local + (5 * nb * itemsize * n_local if chunks > 1 else 0))}
    gathering = record and collect == "trajectory" and (world > 1 or chunks > 1)
    if record and not gathering:
        plan["own recorded trajectory [n_local][T][nb][13]"] = n_local * T * row
    if gathering:
        Tc = T // chunks
        plan["two chunk slabs [nmax][T/chunks][nb][13]"] = 2 * nmax * Tc * row
        if rank == 0:
            plan["assembled Storage layout out[n_total][T][nb][13] (rank 0)"] = n_total * T * row
            if world > 1:
                plan["2 x world receive slabs (rank 0)"] = 2 * world * nmax * Tc * row
    if world > 1:
        plan["final-state gather: padded send block" + (", world receive blocks, assembled result (rank 0)" if rank == 0 else "")] = \
            nmax * row + ((world * nmax + n_total) * row if rank == 0 else 0)
    plan["total"] = sum(plan.values())
    return plan


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
