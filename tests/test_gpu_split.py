"""One capture over two ranks (qampy_amd.distributed.SplitCaptureReceiver) on ONE MI355X: two processes share the GPU, gloo
carries the all-reduce of the segments' end taps.  scripts/split_check.py compares with the single-process tier-b run of the
same capture: identical pass counts and symbol errors, taps and outputs equal (the same kernels train every segment, only on
different ranks), all ranks end with identical taps."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_split_capture_equals_single_process():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "scripts", "split_check.py"), "--same-gpu", "--backend", "gloo"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert "SPLIT_CHECK_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rep = json.loads(line)
    assert rep["world"] == 2 and rep["ranks_identical"] and rep["max_tap_diff"] < 1e-6 and rep["exchanges"] >= 4


def test_bench_split_capture_flag():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1", QAMPY_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--split-capture", "--nsym", str(2 ** 20), "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rep = json.loads(line)
    assert rep["n_gpus"] == 2 and rep["ranks_seen"] == 2 and rep["scaling"] == "strong" and rep["config"]["channels"] == 1
    assert "all-reduce" in rep["config"]["parallelism"] and rep["tier_b"]["certified"] is not None
    assert rep["ser"]["errors_rank0"] == [0, 0]
