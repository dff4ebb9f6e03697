"""
Multi-GPU sharding of the hot path: one independent channel (capture) per rank, no collective on the data path.

The reference has no distributed runtime; its natural data parallelism is per channel / capture.  Each
rank owns one GPU and one channel; ``torch.distributed`` (backend "nccl" = RCCL on the GPUs, "gloo" in the CPU tests)
carries only the barrier, the MAX of the elapsed time and the SUM of the symbol-error counters.  torch is plumbing here and
is imported lazily so that the package itself does not depend on it.
"""
import os

BASE_SEED = 1000


def rank_info(env=None):
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when launched plainly."""
    env = os.environ if env is None else env
    return int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")), int(env.get("WORLD_SIZE", "1"))


def channel_seed(rank):
    """Seed of the synthetic channel processed by ``rank`` (seeds 1000 + channel)."""
    return BASE_SEED + int(rank)


def reduce_max_time(elapsed, dist=None, device="cpu"):
    """Job time = slowest rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed)
    import torch
    t = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum_counts(counts, dist=None, device="cpu"):
    """Element-wise sum over ranks of a small table of counters (e.g. [[errors, symbols] per mode])."""
    import numpy as np
    counts = np.asarray(counts, dtype=np.float64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return counts
    import torch
    c = torch.tensor(counts, dtype=torch.float64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return c.cpu().numpy()


def aggregate_throughput(nsym_per_channel, world, steps, elapsed_max):
    """Whole-job MSym/s: every rank processed ``steps`` passes over ``nsym_per_channel`` symbol periods (weak scaling)."""
    return nsym_per_channel * world * steps / elapsed_max / 1e6
