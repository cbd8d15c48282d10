"""The N > 1 control flow of bench.py on a 1-GPU box: two ranks share cuda:0 and rendezvous over gloo
(test hooks FPX_BENCH_SHARE_GPU / FPX_BENCH_BACKEND; the driver's runs use one GPU per rank and RCCL).
Covers both sharding modes: acceptor groups per rank (no exchange) and the replica axis with the
all-reduce(sum) of the vote bitmaps."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(extra):
    env = dict(os.environ, FPX_BENCH_SHARE_GPU="1", FPX_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_group_sharded_bench_two_ranks():
    d = _run([])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    # every rank commits its own 2^20 slots per step: whole-job value counts both
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * 3 - 2 * 3 * (1 << 20)) < 1e-3 * 2 * 3 * (1 << 20)
    assert d["roofline"]["launches_timed"] == 3 and d["roofline"]["frac"] > 0
    assert "cpu_baseline" not in d


def test_replica_sharded_bench_two_ranks():
    d = _run(["--shard", "replica", "--ballot", "acceptor"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["sharding"] == "replica"
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * 3 - 3 * (1 << 20)) < 1e-3 * 3 * (1 << 20)


def test_single_gpu_bench_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d
    assert d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "int32"
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0
    assert "workload" in d["config"]
