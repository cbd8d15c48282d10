"""the randomised stress scripts under profiles/microbench/ as (short) GPU tests: random geometries, batch orders, row
layouts and call mixes beside the oracle; the scripts print their mismatch count"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_mencius_multipaxos_stress_has_no_mismatches():
    env = dict(os.environ, SEEDS="24", GRAFT_REPO_ROOT=ROOT)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "microbench", "stress_mencius.py")],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "24 geometries, mismatches: 0" in run.stdout, run.stdout + run.stderr
