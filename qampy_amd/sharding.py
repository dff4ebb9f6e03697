"""
Multi-GPU sharding of the hot path: one independent channel (capture) per rank, no collective on the data path.

The reference has no distributed runtime; its natural data parallelism is per channel / capture.  Each
rank owns one GPU and one channel; the process group (:mod:`qampy_amd.comm`: RCCL through ctypes on the GPUs, a socket star
on CPU) carries only the barrier, the MAX of the elapsed time and the SUM of the symbol-error counters.
"""
import numpy as np

from .comm import rank_info          # noqa: F401  (re-exported: the launcher's RANK / LOCAL_RANK / WORLD_SIZE)

BASE_SEED = 1000


def channel_seed(rank):
    """Seed of the synthetic channel processed by ``rank`` (seeds 1000 + channel)."""
    return BASE_SEED + int(rank)


def reduce_max_time(elapsed, comm=None):
    """Job time = slowest rank."""
    if comm is None or comm.world == 1:
        return float(elapsed)
    return float(comm.allreduce([float(elapsed)], "max")[0])


def reduce_sum_counts(counts, comm=None):
    """Element-wise sum over ranks of a small table of counters (e.g. [[errors, symbols] per mode])."""
    counts = np.asarray(counts, dtype=np.float64)
    if comm is None or comm.world == 1:
        return counts
    return comm.allreduce(counts, "sum")


def aggregate_throughput(nsym_per_channel, world, steps, elapsed_max):
    """Whole-job MSym/s: every rank processed ``steps`` passes over ``nsym_per_channel`` symbol periods (weak scaling)."""
    return nsym_per_channel * world * steps / elapsed_max / 1e6
