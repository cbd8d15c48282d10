"""CPU-side checks of the drop-in boundary: libfpx.so loads, exports every symbol include/fpx.h
declares, validates configurations (host logic only -- no kernels run without a GPU), and fails
loudly instead of falling back when there is no device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fa():
    import frankenpaxos_amd

    if not os.path.exists(frankenpaxos_amd._lib.SO_PATH):
        frankenpaxos_amd.build()
    return frankenpaxos_amd


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "fpx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(fpx_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(fa):
    lib = C.CDLL(fa._lib.SO_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), "libfpx.so does not export %s" % name
    # and the python binding binds exactly the header's surface
    assert sorted(fa._lib.SIGNATURES) == names
    # the wire adapter's header (include/fpx_wire.h) is part of the same library
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "fpx_wire.h")).read(), flags=re.S)
    wire = sorted(set(re.findall(r"\b(fpx_wire_[a-z0-9_]+)\s*\(", hdr)))
    assert len(wire) >= 26
    for name in wire:
        assert hasattr(lib, name), "libfpx.so does not export %s" % name
    # ... and so is dependency-graph execution (include/fpx_depgraph.h)
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "fpx_depgraph.h")).read(), flags=re.S)
    dg = sorted(set(re.findall(r"\b(fpx_depgraph_[a-z0-9_]+)\s*\(", hdr)))
    from frankenpaxos_amd import depgraph
    assert dg == sorted(depgraph.SIGNATURES) and len(dg) == 9
    for name in dg:
        assert hasattr(lib, name), "libfpx.so does not export %s" % name


def test_header_compiles_as_c_and_cxx(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "fpx.h"\n#include "fpx_wire.h"\n#include "fpx_depgraph.h"\n'
                   'int main(void){fpx_config c; fpx_wire_epx_msg m; fpx_depgraph_config d; (void)c; (void)m; (void)d; '
                   'return FPX_OK + FPX_WIRE_OTHER + FPX_DG_TARJAN;}\n')
    inc = os.path.join(ROOT, "include")
    assert os.system("gcc -std=c99 -Wall -Werror -I%s -c %s -o %s" % (inc, src, tmp_path / "t.o")) == 0
    assert os.system("g++ -std=c++17 -Wall -Werror -I%s -x c++ -c %s -o %s" % (inc, src, tmp_path / "t2.o")) == 0


def test_config_struct_layout_matches_the_oracle(fa, oracle):
    a, b = fa.FpxConfig, oracle.Config
    assert [(n, t) for n, t in a._fields_] == [(n, t) for n, t in b._fields_]
    assert C.sizeof(a) == C.sizeof(b) == 60


CASES = [
    (dict(num_slots=64, num_replicas=3, f=1), 0),
    (dict(num_slots=0, num_replicas=3, f=1), 1),
    (dict(num_slots=64, num_replicas=0), 1),
    (dict(num_slots=64, num_replicas=257), 1),
    (dict(num_slots=64, num_replicas=3, f=3), 1),                      # quorum larger than the group
    (dict(num_slots=64, num_replicas=3, f=1, tally_ways=9), 1),
    (dict(num_slots=64, num_replicas=6, quorum_kind=2, grid_rows=2, grid_cols=3), 0),
    (dict(num_slots=64, num_replicas=6, quorum_kind=2, grid_rows=2, grid_cols=2), 1),  # Grid.scala:14-17
    (dict(num_slots=64, num_replicas=3, quorum_kind=7), 1),
    (dict(num_slots=64, num_replicas=128, f=127, replica_base=128, replicas_total=256), 0),
    (dict(num_slots=64, num_replicas=128, f=127, replica_base=130, replicas_total=256), 1),
    (dict(num_slots=64, num_replicas=3, f=1, num_groups=0), 1),
    (dict(num_slots=64, num_replicas=3, f=1, num_leaders=0), 1),
    (dict(num_slots=64, num_replicas=3, f=1, ballot_mode=2), 1),
]


@pytest.mark.parametrize("kw,want", CASES)
def test_config_check_matches_oracle(fa, oracle, kw, want):
    assert fa.lib().fpx_config_check(C.byref(fa.make_config(**kw))) == want
    assert oracle.lib().fpo_config_check(C.byref(oracle.make_config(**kw))) == want


def test_no_cpu_fallback(fa):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(fa.FpxError) as e:
        fa.Context(fa.make_config(num_slots=64, num_replicas=3, f=1))
    assert e.value.status == fa.FPX_ENODEVICE
    import numpy as np

    cfg = fa.make_config(num_slots=1, num_replicas=5, quorum_kind=1)
    with pytest.raises(fa.FpxError) as e:
        fa.quorum_eval(cfg, np.zeros((1, 4), np.uint64))
    assert e.value.status == fa.FPX_ENODEVICE
    with pytest.raises(fa.FpxError) as e:
        fa.PinnedArray((16,), np.int32)
    assert e.value.status == fa.FPX_ENODEVICE


def test_product_never_imports_the_oracle():
    """the package and the C++ sources must not reference oracle/ (SURVEY / task rule)"""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "frankenpaxos_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                if re.search(r"\boracle\b|fpo_|pyoracle", txt):
                    bad.append(os.path.join(dirpath, fn))
    assert not bad, bad


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """bench.py --gpus N must never report another world size under the N-GPU label"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], capture_output=True,
                         text=True, env=env, timeout=300)
    assert out.returncode != 0 and "--gpus 4 but WORLD_SIZE is 1" in out.stderr
    # ... and without a launcher around it, it launches its own ranks: one per GPU, 127.0.0.1 rendezvous
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FPX_BENCH_DRY_SPAWN"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--test-hooks", "--gpus", "8", "--steps", "5"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["spawn"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == ["--test-hooks", "--gpus", "8", "--steps", "5"] and cmd[-6].endswith("bench.py")
    # a hook variable that leaked into the environment does not change what the line measures -- without
    # --test-hooks the run is refused
    env["FPX_BENCH_SLOTS_LOG2"] = "16"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], capture_output=True, text=True,
                         env=env, timeout=300)
    assert out.returncode != 0 and "without --test-hooks" in out.stderr


def test_bench_deadline_helper():
    """bench.run_with_deadline: a value, an exception turned into an "error" field, and a call that never returns
    reported as such (the multi-GPU run uses it around the extra replica-axis row)"""
    import sys
    import time
    sys.path.insert(0, ROOT)
    import bench
    assert bench.run_with_deadline(lambda: {"x": 1}, 5, None) == ({"x": 1}, False)
    out, hung = bench.run_with_deadline(lambda: 1 // 0, 5, None)
    assert not hung and "ZeroDivisionError" in out["error"]
    out, hung = bench.run_with_deadline(lambda: time.sleep(3), 0.2, None)
    assert hung and "no answer" in out["error"]
