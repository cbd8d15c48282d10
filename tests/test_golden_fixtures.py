"""Committed regression fixtures (tests/golden/*.npz): OUTPUTS OF THE CPU ORACLE frozen by
tests/golden/make_golden.py -- NOT reference vectors (the JVM reference cannot run here, and it holds no
known-answer test for the vote / tally: SURVEY.md F9).  The oracle must keep reproducing them (CPU), and the
HIP path must match them bit for bit (GPU).  The vectors that DO come from the reference are in
tests/test_oracle_golden.py (quorums, round system, BufferMap, TopOne, popularItems, conflict index),
tests/test_epaxos.py (IntPrefixSet properties) and tests/golden/wire_vectors.json (protobuf bytes)."""
import os

import numpy as np
import pytest

from tests.golden import make_golden as G

NAMES = sorted(G.CASES)


def load(name):
    return np.load(os.path.join(os.path.dirname(G.__file__), name + ".npz"))


def compare(got, want):
    assert sorted(got) == sorted(want.files)
    for k in want.files:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_fixture(oracle, name):
    cfg_kw, script_kw = G.CASES[name]
    compare(G.run_case(oracle.System(oracle.make_config(**cfg_kw)), cfg_kw, script_kw), load(name))


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_matches_fixture(name):
    import frankenpaxos_amd as fa

    cfg_kw, script_kw = G.CASES[name]
    compare(G.run_case(fa.Context(fa.make_config(**cfg_kw)), cfg_kw, script_kw), load(name))
