"""The line `bench.py` prints is the record the driver keeps: it has to be ONE JSON object, small enough for the driver to hold
(round 5's 30 KB line came back as `parsed: null`), and carry the contract's keys + `roofline` + `cpu_baseline`.  The full result goes to a
detail file.  CPU only: the assembly is run on a stored full result and through `--dry-run` (kernels replaced by a sleep)."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("stored", ["r05_bench_c3.json", "r04_bench_c3.json", "r05_bench_c5.json"])
def test_headline_line_of_a_full_result_is_small_and_complete(stored, tmp_path):
    bench = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", stored)))
    assert len(json.dumps(full)) > 2000
    line = bench.headline_line(full, "gpurun_out/bench_detail.json")
    assert "\n" not in line and len(line) < bench.LINE_MAX_BYTES < 8000
    res = json.loads(line)
    for k in CONTRACT:
        assert k in res, k
    assert res["value"] == pytest.approx(full["value"], rel=1e-5) and res["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert res["config"]["workload"] == full["config"]["workload"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in res["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in res["cpu_baseline"], k
    assert all(len(v) <= bench.LINE_STR_MAX for v in res["config"].values() if isinstance(v, str))
    # emit(): the full result lands in the detail file, stdout gets the line only
    import io, contextlib
    buf = io.StringIO()
    path = str(tmp_path / "d" / "detail.json")
    with contextlib.redirect_stdout(buf):
        bench.emit(full, path)
    printed = buf.getvalue()
    assert printed.count("\n") == 1 and json.loads(printed)["detail"] == path
    assert json.load(open(path))["value"] == full["value"]


def test_headline_line_survives_an_oversized_block():
    """Whatever a block grows to, the printed line stays under the bound (optional blocks are dropped, the contract keys stay)."""
    bench = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_c3.json")))
    full["ser"] = dict(per_mode_rank0=[0.0] * 4000)
    full["tier_b"]["checks"] = {("k%d" % i): True for i in range(2000)}
    line = bench.headline_line(full, None)
    assert len(line) < bench.LINE_MAX_BYTES
    res = json.loads(line)
    assert all(k in res for k in CONTRACT) and "roofline" in res and "cpu_baseline" in res


def test_two_rank_line_parses_and_is_small(tmp_path):
    """`bench.py --gpus 2 --dry-run` end to end over the socket backend: rank 0's line is the compact one, with the per-rank step times."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    detail = str(tmp_path / "detail.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--detail-out", detail],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 8000
    res = json.loads(lines[0])
    for k in CONTRACT:
        assert k in res, k
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2 and len(res["ms_per_step_per_rank"]) == 2 and res["comm_backend"] == "tcp"
    assert res["detail"] == detail and json.load(open(detail))["n_gpus"] == 2


def test_two_rank_line_under_torchrun_as_the_driver_starts_it(tmp_path):
    """The driver's own command for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N --steps K --warmup W` (here with --dry-run: no GPU).  Exactly ONE JSON line on stdout, from rank 0, compact, with both ranks seen."""
    pytest.importorskip("torch")
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    detail = str(tmp_path / "detail.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--detail-out", detail]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 8000, lines
    res = json.loads(lines[0])
    for k in CONTRACT:
        assert k in res, k
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2 and res["steps"] == 3 and res["warmup"] == 1 and len(res["ms_per_step_per_rank"]) == 2
    assert res["value"] == pytest.approx(2 * res["config"]["nsym_per_channel"] * 3 / (res["ms_per_step"] * 3e-3) / 1e6, rel=1e-3)
