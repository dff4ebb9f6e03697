"""
One capture over several processes / GPUs (optional mode, BASELINE.json north star: "capture segments across the GPUs ... with
RCCL broadcast of converged taps").  The natural way to use N GPUs on this path is N independent captures (bench.py --gpus N: no
collective on the data path).  This module is the other way round: the segments of the parallel-in-time trainer
 of ONE capture are spread over the ranks of a ``torch.distributed`` group.

Every rank holds the whole capture (128 MiB at C3) and makes the same calls.  In every pass a rank trains only the segments it
owns; the segments' end taps (S x 2.6 KiB: 5 - 10 MiB) are then summed over the ranks in place - an all-reduce of buffers that
are zero outside the owned segments, i.e. an all-gather (RCCL over xGMI; gloo works too and is what the tests use) - and the
cheap part of a pass (boundary defects, coarse correction, convergence decision) runs redundantly on identical data, so all
ranks take identical decisions without further communication.  Acquisition, filter and carrier recovery are not split.

PyTorch is used for the process group only; the tensors it reduces alias the library's device buffers.
"""
import ctypes as C

import numpy as np

from . import _lib
from .pipeline import ResidentReceiver


class _DeviceBytes:
    """A raw device pointer as something ``torch.as_tensor`` accepts (no copy)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = dict(shape=(int(nbytes) // 4,), typestr="<f4", data=(int(ptr), False), version=3)


def owned_segments(S, rank, world):
    """Contiguous, nearly equal shares of the S segments: (first, count) of ``rank``."""
    first = (S * rank) // world
    return first, (S * (rank + 1)) // world - first


class SplitCaptureReceiver(ResidentReceiver):
    """:class:`qampy_amd.pipeline.ResidentReceiver` (tier b) whose training passes are shared by the ranks of ``group``
    (default: the world group).  All ranks must load the same capture and call the same methods in the same order."""

    def __init__(self, *args, group=None, **kw):
        import torch
        import torch.distributed as dist
        kw["tier"] = "b"
        super().__init__(*args, **kw)
        self._torch, self._dist, self._group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.exchanged_bytes = 0
        self.exchanges = 0

        def exchange(user, ptr, nbytes):
            try:
                t = torch.as_tensor(_DeviceBytes(ptr, nbytes), device="cuda")
                dist.all_reduce(t, group=group)
                torch.cuda.synchronize()
                self.exchanged_bytes += int(nbytes)
                self.exchanges += 1
                return 0
            except Exception as e:                     # never let an exception cross the C boundary
                import sys
                print("qampy_amd.distributed: exchange failed: %r" % (e,), file=sys.stderr)
                return 1

        self._exchange = _lib.PIT_EXCHANGE(exchange)                  # keep the thunk alive
        for o in self.pit:
            first, count = owned_segments(int(o["segments"]), self.rank, self.world)
            o["seg_first"], o["seg_count"] = first, count
            o["exchange"] = C.cast(self._exchange, C.c_void_p).value

    def broadcast_taps(self, src=0):
        """The north star's 'broadcast of converged taps': ranks end a stage with identical taps by construction; this makes it
        explicit (e.g. after rank ``src`` alone continued training)."""
        t = self._torch.as_tensor(_DeviceBytes(self.wxy.ptr, self.wxy.nbytes), device="cuda")
        self._dist.broadcast(t, src=src, group=self._group)
        self._torch.cuda.synchronize()
