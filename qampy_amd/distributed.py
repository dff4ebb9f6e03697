"""
One capture over several processes / GPUs (optional mode, BASELINE.json north star: "capture segments across the GPUs ... with
RCCL broadcast of converged taps").  The natural way to use N GPUs on this path is N independent captures (bench.py --gpus N: no
collective on the data path).  This module is the other way round: the segments of the parallel-in-time trainer
 of ONE capture are spread over the ranks of a :class:`qampy_amd.comm.Comm` group.

Every rank holds the whole capture (128 MiB at C3) and makes the same calls.  In every pass a rank trains only the segments it
owns; the segments' end taps (S x 2.6 KiB: 5 - 10 MiB) are then summed over the ranks in place - an all-reduce of buffers that
are zero outside the owned segments, i.e. an all-gather, done on the 32-bit integer view so that the one non-zero contribution
arrives bit for bit (x + 0 is exact for integers; it is not for -0.0 or NaN payloads) - and the cheap part of a pass (boundary
defects, coarse correction, convergence decision) runs redundantly on identical data, so all ranks take identical decisions
without further communication.  With RCCL the all-reduce is ENQUEUED on the library stream between the kernels of the pass (no
host synchronisation, the next pass is enqueued ahead as in the single-process case); the socket backend (tests: two ranks on
one GPU) stages through the host.  Acquisition, filter and carrier recovery are not split.
"""
import ctypes as C
import sys

import numpy as np

from . import _lib
from .pipeline import ResidentReceiver


def owned_segments(S, rank, world):
    """Contiguous, nearly equal shares of the S segments: (first, count) of ``rank``."""
    first = (S * rank) // world
    return first, (S * (rank + 1)) // world - first


class SplitCaptureReceiver(ResidentReceiver):
    """:class:`qampy_amd.pipeline.ResidentReceiver` (tier b) whose training passes are shared by the ranks of ``comm``.
    All ranks must load the same capture and call the same methods in the same order."""

    def __init__(self, *args, comm=None, **kw):
        if comm is None:
            raise ValueError("SplitCaptureReceiver needs the process group (qampy_amd.comm.Comm)")
        kw["tier"] = "b"
        super().__init__(*args, **kw)
        if any(self.adaptive):                         # (however it was passed: by keyword or by position)
            raise ValueError("split capture: the adaptive step is solved one output mode at a time inside one process (csrc/train_pit.h); "
                             "use a fixed step size, or a ResidentReceiver per rank")
        self.comm = comm
        self.rank, self.world = comm.rank, comm.world
        self.exchanged_bytes = 0
        self.exchanges = 0
        from .core.equalisation import hip_equalisation as _k
        for s_, o in enumerate(self.pit):              # the grid the library will really use (it keeps >= 4 blocks per segment)
            if int(o["segments"]) == 0:                # "automatic": the library's own count for this sweep, as it would pick it
                o["segments"] = _k.pit_auto_segments(self.TrSyms[s_], float(self.mu0[s_]), self.modes.size, cold=bool(o.get("acquire")))
            o["segments"] = _k.pit_effective_segments(o["segments"], self.TrSyms[s_])
            if int(o["segments"]) < self.world:
                raise ValueError("split capture: stage %d has %d segments, too few to be shared by %d ranks" % (s_, int(o["segments"]), self.world))

        def exchange(user, ptr, nbytes):
            try:
                comm.allreduce_dev(ptr, int(nbytes) // 4, np.uint32, "sum")
                self.exchanged_bytes += int(nbytes)
                self.exchanges += 1
                return 0
            except Exception as e:                     # never let an exception cross the C boundary
                print("qampy_amd.distributed: exchange failed: %r" % (e,), file=sys.stderr)
                return 1

        self._exchange = _lib.PIT_EXCHANGE(exchange)                  # keep the thunk alive
        for o in self.pit:
            first, count = owned_segments(int(o["segments"]), self.rank, self.world)
            o["seg_first"], o["seg_count"] = first, count
            o["exchange"] = C.cast(self._exchange, C.c_void_p).value
            o["exchange_on_stream"] = 1 if comm.on_stream else 0

    def broadcast_taps(self, src=0):
        """The north star's 'broadcast of converged taps': ranks end a stage with identical taps by construction; this makes it
        explicit (e.g. after rank ``src`` alone continued training)."""
        self.comm.broadcast_dev(self.wxy.ptr, self.wxy.nbytes, root=src)
        _lib.sync()
