"""The N > 1 path of bench.py on CPU: world size 2, 127.0.0.1 rendezvous, qampy_amd.comm's socket backend (the collectives RCCL
carries on the GPUs).  Covers the per-rank channel assignment, the barrier / MAX-time / SUM-counter reductions, the launcher
(ours and torch.distributed.run, as the driver starts bench.py) and the whole-job throughput formula (no data-path collective exists)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT
from qampy_amd import sharding, synth


def test_channel_assignment_is_independent_per_rank():
    assert sharding.rank_info({}) == (0, 0, 1)
    assert sharding.rank_info({"RANK": "3", "LOCAL_RANK": "1", "WORLD_SIZE": "8"}) == (3, 1, 8)
    assert [sharding.channel_seed(r) for r in range(3)] == [1000, 1001, 1002]
    a = synth.make_capture(16, 256, seed=sharding.channel_seed(0), snr_db=20)
    b = synth.make_capture(16, 256, seed=sharding.channel_seed(1), snr_db=20)
    assert a.shape == b.shape and not np.allclose(np.asarray(a), np.asarray(b))
    assert sharding.aggregate_throughput(2 ** 22, 8, 5, 2.0) == 2 ** 22 * 8 * 5 / 2.0 / 1e6
    assert sharding.reduce_max_time(1.5) == 1.5                      # no process group: identity
    assert np.array_equal(sharding.reduce_sum_counts([[1, 2], [3, 4]]), [[1, 2], [3, 4]])


WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import numpy as np
    from qampy_amd import sharding
    from qampy_amd.comm import Comm
    rank, local, world = sharding.rank_info()
    cm = Comm(device=None)                    # no GPU here: socket collectives
    cm.barrier()
    elapsed = 1.0 + rank                      # rank 1 is the slow one
    tmax = sharding.reduce_max_time(elapsed, cm)
    counts = sharding.reduce_sum_counts([[rank + 1, 100], [0, 100]], cm)
    lo = cm.allreduce([float(rank)], "min")
    cm.barrier()
    if rank == 0:
        print(json.dumps(dict(tmax=tmax, counts=counts.tolist(), seed=sharding.channel_seed(rank), backend=cm.backend, lo=lo.tolist(),
                              value=sharding.aggregate_throughput(1000, world, 2, tmax))))
    cm.close()
""")


def _check_worker_output(out):
    import json
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["tmax"] == 2.0                                  # MAX over ranks
    assert res["counts"] == [[3.0, 200.0], [0.0, 200.0]]       # SUM over ranks
    assert res["lo"] == [0.0] and res["backend"] == "tcp"
    assert res["value"] == 1000 * 2 * 2 / 2.0 / 1e6


def test_world_size_two_own_launcher(tmp_path):
    """Two ranks started by qampy_amd.comm.launch (what `bench.py --gpus 2` does when it is not inside a launcher)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    code = "import sys; sys.path.insert(0, %r); from qampy_amd import comm; sys.exit(comm.launch(%r, [], 2))" % (ROOT, str(script))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240, env=dict(os.environ, OMP_NUM_THREADS="1"))
    _check_worker_output(out)


def test_world_size_two_under_torchrun(tmp_path):
    """The same two ranks under `python -m torch.distributed.run` - how the driver starts bench.py for N > 1: the ranks only read its
    environment (its own store keeps MASTER_PORT; ours publishes an ephemeral port in a file keyed by that port and the parent pid)."""
    pytest.importorskip("torch")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=dict(os.environ, OMP_NUM_THREADS="1"))
    _check_worker_output(out)


def test_product_tree_does_not_import_torch():
    """North star: Python host code over a ctypes C-ABI, no PyTorch - neither the package nor bench.py imports it."""
    import re
    bad = []
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for d, _, fs in os.walk(os.path.join(ROOT, "qampy_amd")):
        files += [os.path.join(d, f) for f in fs if f.endswith(".py")]
    for f in files:
        for i, line in enumerate(open(f), 1):
            if re.match(r"\s*(import torch|from torch)", line):
                bad.append("%s:%d" % (f, i))
    assert not bad, bad


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` (no torchrun around it) starts two ranks itself, checks WORLD_SIZE == --gpus and reports
    ranks_seen from an all-reduce of ones; --dry-run replaces the kernels by a sleep (socket collectives), so this runs without a GPU."""
    import json
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2 and res["steps"] == 3 and res["dry_run"] is True and res["comm_backend"] == "tcp"
    assert res["ser"]["symbols_all"] == 2 * 2 * 2 ** 22            # both ranks' counters were summed
    assert res["value"] == pytest.approx(2 * 2 ** 22 * 3 / (res["ms_per_step"] * 3e-3) / 1e6, rel=1e-3)
    # a launch whose world size contradicts --gpus is refused
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120, env=env2)
    assert bad.returncode == 2 and "WORLD_SIZE=1" in bad.stdout


def test_owned_segments_partition_the_grid():
    """qampy_amd.distributed: contiguous, disjoint, complete and nearly equal shares for any segment count / world size."""
    from qampy_amd.distributed import owned_segments
    for S in (1, 7, 512, 1920, 3968, 4096):
        for world in (1, 2, 3, 8):
            parts = [owned_segments(S, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == S
            assert all(parts[r][0] + parts[r][1] == parts[r + 1][0] for r in range(world - 1))
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_pit_opts_struct_matches_the_header():
    """The ctypes mirror of qh_pit_opts carries the split-capture fields of include/qampy_hip.h in the header's order."""
    import re
    from qampy_amd import _lib
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "qampy_hip.h")).read()
    body = src[src.index("typedef struct qh_pit_opts {") + len("typedef struct qh_pit_opts {"):src.index("} qh_pit_opts;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for stmt in body.split(";"):
        stmt = stmt.strip()
        if not stmt or stmt.startswith("typedef"):
            continue
        m = re.search(r"\(\*(\w+)\)", stmt)
        if m:
            names.append(m.group(1))
        else:
            names += [n.strip(" *") for n in stmt.split(None, 1)[1].split(",")]
    assert names == [f[0] for f in _lib.PitOpts._fields_]
