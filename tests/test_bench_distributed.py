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
           os.path.join(ROOT, "bench.py"), "--test-hooks", "--gpus", "2", "--steps", "3", "--warmup", "1"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks():
    """VERDICT r01 missing #4: `python bench.py --gpus 2` with no torchrun around it must become 2 ranks
    (here: sharing cuda:0 over gloo) and report n_gpus 2 -- not silently run one rank."""
    env = dict(os.environ, FPX_BENCH_SHARE_GPU="1", FPX_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--test-hooks", "--gpus", "2", "--steps", "2",
                          "--warmup", "1", "--replica-row-steps", "2"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2
    # both rows of SURVEY.md 8e in one run: the group-sharded headline and the replica-axis row
    r = d["replica_axis"]
    assert r["scaling"] == "strong" and r["steps"] == 2 and r["value"] > 0
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 * 2 - 2 * (1 << 20)) < 1e-3 * 2 * (1 << 20)
    assert r["xgmi_bytes_per_gpu_per_step"] == 16 << 20


def test_group_sharded_bench_two_ranks():
    d = _run(["--replica-row-steps", "0"])
    assert "replica_axis" not in d
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    # every rank commits its own 2^20 slots per step: whole-job value counts both
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * 3 - 2 * 3 * (1 << 20)) < 1e-3 * 2 * 3 * (1 << 20)
    assert d["roofline"]["launches_timed"] == 3 and d["roofline"]["frac"] > 0
    # the CPU baseline rides on every line, at N > 1 the single-thread flat port only (rank 0 times it)
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert d["rccl_ranks"] == 0 and "rccl" not in d           # gloo hook: no RCCL communicator was created
    assert d["test_hooks"] == ["FPX_BENCH_BACKEND", "FPX_BENCH_SHARE_GPU"]   # a hooked line says so


def test_replica_sharded_bench_two_ranks():
    d = _run(["--shard", "replica", "--ballot", "acceptor"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["sharding"] == "replica"
    assert d["collective"]["xgmi_bytes_per_gpu_per_step"] == 16 << 20 and d["rccl_ranks"] == 0   # gloo hook
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


@pytest.mark.parametrize("config", ["2", "3", "4", "5"])
def test_bench_lines_of_the_other_configs(config):
    """every BASELINE.json config has a bench line of the headline's schema"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "3",
                          "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "dtype", "config", "roofline"):
        assert key in d
    assert d["config"]["baseline_config"] == int(config) and d["value"] > 0 and d["steps"] == 3
    assert d["roofline"]["launches_timed"] == 3 and d["roofline"]["achieved"] > 0


def test_headline_line_survives_an_extra_row_that_never_finishes():
    """the replica-axis row runs RCCL between real GPUs for the first time in the driver's own multi-GPU job: if a
    collective in it never returns, the headline line is still printed (with the row marked) and every rank leaves
    with status 0.  A deadline of 0 s makes the row 'never finish' here."""
    d = _run(["--replica-row-steps", "2", "--replica-row-deadline", "0"])
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert "no answer" in d["replica_axis"]["error"]


def test_config5_bench_line_on_two_ranks():
    """BASELINE.json configs[4] sharded by leader group over 2 ranks (gloo hook, both on cuda:0):
    128 leader groups and half of the band per rank, strong scaling, whole-job value"""
    d = _run(["--config", "5", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["baseline_config"] == 5
    assert d["config"]["leader_groups_per_gpu"] == 128 and d["config"]["slots_per_step_per_gpu"] == 1 << 21
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - (1 << 22)) < 1e-3 * (1 << 22)


def test_bench_spawns_eight_ranks():
    """the driver's SCALE run asks for --gpus 8: the self-spawn path with 8 ranks (all on cuda:0 here, gloo)"""
    env = dict(os.environ, FPX_BENCH_SHARE_GPU="1", FPX_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--test-hooks", "--gpus", "8", "--steps", "2", "--warmup", "1",
                          "--replica-row-steps", "0"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 2 and d["scaling"] == "weak"
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 8 * (1 << 20)) < 1e-3 * 8 * (1 << 20)
    assert d["cpu_baseline"]["cores"] == 1


def _double():
    from tests.test_gpu_sharding_world2 import build_double
    return build_double()


@pytest.mark.parametrize("shard", ["group", "replica"])
def test_eight_rank_lines_through_the_library_communicator(shard):
    """the lines the driver's 8-GPU run prints, end to end with EIGHT ranks before the driver does --
    all on cuda:0, rendezvous over gloo, but the data path through fpx_comm_create and the library's own collectives
    (FPX_BENCH_FPX_COMM=1; RCCL refuses two ranks on one device, so its symbols come from tests/rccl_double).  group: the
    weak-scaling headline + the replica-axis row (32 acceptors per rank, 8-slice reduce-scatter, all-reduce(max) of the
    Nack rounds) + the all-gather of Chosen records of the `rccl` probe; replica: the strong-scaling line itself."""
    env = dict(os.environ, FPX_BENCH_SHARE_GPU="1", FPX_BENCH_BACKEND="gloo", FPX_BENCH_FPX_COMM="1", FPX_RCCL_LIB=_double(),
               FPX_BENCH_SLOTS_LOG2="16")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    extra = ["--replica-row-steps", "2"] if shard == "group" else ["--shard", "replica"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--test-hooks", "--gpus", "8", "--steps", "2", "--warmup", "1"] + extra,
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    n = 1 << 16
    assert d["n_gpus"] == 8 and d["steps"] == 2 and d["config"]["slots_per_step"] == n
    assert d["rccl_ranks"] == 8
    if shard == "group":
        assert d["scaling"] == "weak" and abs(d["value"] * d["ms_per_step"] * 1e-3 - 8 * n) < 1e-3 * 8 * n
        r = d["replica_axis"]
        assert "error" not in r, r
        assert r["rccl_ranks"] == 8 and r["steps"] == 2 and r["collective_avg_ms_max_over_ranks"] > 0
        assert abs(r["value"] * r["ms_per_step"] * 1e-3 - n) < 1e-3 * n
        assert d["rccl"] == {"ranks": 8, "allgather_of_chosen_records_ok": True}
    else:
        assert d["scaling"] == "strong" and d["collective"]["calls_timed"] == 2 and d["collective"]["avg_ms"] > 0
        assert abs(d["value"] * d["ms_per_step"] * 1e-3 - n) < 1e-3 * n


def test_preflight_of_the_first_multi_gpu_run():
    """`bench.py --gpus 8 --preflight` -- eight ranks creating headline-shaped contexts at once (their
    placement searches probing one device together, each within its budget), fpx_comm_create under a deadline, one
    replica-sharded step and one all-gather through the library's collectives (the RCCL double: the ranks share cuda:0),
    every rank's times in ONE line, exit status 0; and a communicator that cannot come up (the double pointed at a file
    that is no library) is reported with status != 0 instead of hanging"""
    env = dict(os.environ, FPX_BENCH_SHARE_GPU="1", FPX_BENCH_BACKEND="gloo", FPX_BENCH_FPX_COMM="1", FPX_RCCL_LIB=_double())
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--test-hooks", "--gpus", "8", "--preflight"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["preflight"] and d["ok"] and d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and not d["errors"], d
    assert len(d["placement_search_ms_per_rank"]) == 8 and all(0 < x < 5000 for x in d["placement_search_ms_per_rank"]), d
    assert all(x > 0 for x in d["collective_ms_per_rank"]) and all(0 < x < 60 for x in d["comm_create_s_per_rank"]), d
    env["FPX_RCCL_LIB"] = os.path.join(ROOT, "README.md")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--test-hooks", "--gpus", "2", "--preflight", "--comm-deadline", "20"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and out.returncode != 0, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads(lines[0])
    assert not d["ok"] and d["errors"], d
