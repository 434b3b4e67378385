"""Multi-GPU plumbing: one process per GPU, torch.distributed ("nccl" = RCCL over
xGMI on the box, "gloo" in CPU tests).

The path shards naturally, so there is exactly one collective: the per-step SUM all-reduce of the
flat gradient (P floats, 0.1-2 MB: latency-bound on xGMI, one fused buffer, no bucketing).
Chains and posterior samples shard round-robin with no collective.
"""
import numpy as np
import torch.distributed as dist


# dev switch (bench.py --force-collective): run the collective on a single-rank group too, so that its
# stream choreography and cost can be exercised on one GPU
FORCE_COLLECTIVE = False


def _collective_on():
    return dist_info()[1] > 1 or (FORCE_COLLECTIVE and dist.is_available() and dist.is_initialized())


def dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def allreduce_sum_(t):
    """In-place SUM over ranks of one flat tensor (no-op on a single rank)."""
    if _collective_on():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_sum_start(t):
    """Start the in-place SUM over ranks; returns a handle for allreduce_wait (None on one rank).
    The collective runs on RCCL's own stream once the current stream's work on `t` is done, so
    kernels enqueued after this call overlap with it."""
    if _collective_on():
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
    return None


def allreduce_wait(handle):
    """The current stream waits for the collective started by allreduce_sum_start."""
    if handle is not None:
        handle.wait()


def step_rng(seed, rank, key):
    """Host RNG of one (rank, step): ranks draw different (ls, var, sigma), replays are exact."""
    return np.random.default_rng([int(seed), int(rank)] + [int(k) for k in key])


def philox_stream_id(rank, key, purpose):
    """64-bit Philox stream id: rank | epoch key | step | purpose (1: eps, 2: eps', 3: eps_z)."""
    return ((rank & 0xFF) << 56) | ((int(key[0]) & 0xFFFFFF) << 28) | ((int(key[-1]) & 0xFFFFFF) << 4) | (purpose & 0xF)


def shard(n_items, rank=None, world=None):
    """Round-robin share of range(n_items) for this rank (chains, posterior samples)."""
    if rank is None:
        rank, world = dist_info()
    return range(rank, n_items, world)
