"""The C++ host mirror (frankenpaxos_amd/host/fpx.hpp): compiles and links against libfpx.so
without a GPU (host logic); on the GPU box the transcribed reference unit tests + the 1000-command
MultiPaxos scenario (BASELINE.json configs[0]) run through it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "build")
EXE = os.path.join(BUILD, "host_mirror_test")


def build_driver():
    import frankenpaxos_amd

    if not os.path.exists(frankenpaxos_amd._lib.SO_PATH):
        frankenpaxos_amd.build()
    os.makedirs(BUILD, exist_ok=True)
    csrc = os.path.join(ROOT, "frankenpaxos_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "host_mirror_test.cpp"),
           "-o", EXE, "-L" + csrc, "-lfpx", "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return EXE


def test_host_mirror_compiles_and_links():
    assert os.path.exists(build_driver())


def test_host_mirror_dependency_graph_tests():
    """the dependency-graph part of the mirror is host code: the reference's depgraph tests run without a GPU"""
    exe = build_driver()
    out = subprocess.run([exe, "--host-only"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host-only tests passed" in out.stdout


@pytest.mark.gpu
def test_host_mirror_reference_unit_tests():
    exe = build_driver()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout
