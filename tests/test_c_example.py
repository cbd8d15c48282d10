"""examples/phase2_demo.c: the C ABI used from plain C (no Python, no C++).  It must compile with -Wall -Werror
as C11 and link against libfpx.so; without a device it reports FPX_ENODEVICE (exit 77), with one it runs
BASELINE.json configs[0] (1000 commands, all chosen; a stale leader is Nacked) and exits 0."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_caller_builds_links_and_fails_loudly_without_a_device():
    import torch

    import frankenpaxos_amd

    if not os.path.exists(frankenpaxos_amd._lib.SO_PATH):
        frankenpaxos_amd.build()
    csrc = os.path.join(ROOT, "frankenpaxos_amd", "csrc")
    out_dir = os.path.join(ROOT, "tests", "build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "phase2_demo")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "examples", "phase2_demo.c"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + csrc, "-lfpx", "-Wl,-rpath," + csrc,
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    if torch.cuda.is_available():
        assert run.returncode == 0, run.stdout + run.stderr
        assert "1000 of 1000 commands chosen" in run.stdout
    else:
        assert run.returncode == 77, run.stdout + run.stderr
        assert "no usable gfx950 device" in run.stdout


def test_plain_c_epaxos_caller():
    """examples/epaxos_demo.c: tick -> Accept phase of the slow path -> dependency graph -> execution, through
    include/fpx.h and include/fpx_depgraph.h alone"""
    import torch

    import frankenpaxos_amd

    if not os.path.exists(frankenpaxos_amd._lib.SO_PATH):
        frankenpaxos_amd.build()
    csrc = os.path.join(ROOT, "frankenpaxos_amd", "csrc")
    out_dir = os.path.join(ROOT, "tests", "build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "epaxos_demo")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "examples", "epaxos_demo.c"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + csrc, "-lfpx", "-Wl,-rpath," + csrc,
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    if torch.cuda.is_available():
        assert run.returncode == 0, run.stdout + run.stderr
        assert "400 of 400 commands executed" in run.stdout and " 5 blockers" in run.stdout
    else:
        assert run.returncode == 77, run.stdout + run.stderr
        assert "no usable gfx950 device" in run.stdout


def test_plain_c_mencius_caller():
    """examples/mencius_demo.c: leader groups with commands (their batches back to back), leader groups that skip with
    noop ranges, the replica log executing the band, a leader change in one group -- through include/fpx.h alone"""
    import torch

    import frankenpaxos_amd

    if not os.path.exists(frankenpaxos_amd._lib.SO_PATH):
        frankenpaxos_amd.build()
    csrc = os.path.join(ROOT, "frankenpaxos_amd", "csrc")
    out_dir = os.path.join(ROOT, "tests", "build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "mencius_demo")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "examples", "mencius_demo.c"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + csrc, "-lfpx", "-Wl,-rpath," + csrc,
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    if torch.cuda.is_available():
        assert run.returncode == 0, run.stdout + run.stderr
        assert "2048 of 2048 chosen" in run.stdout and "executed watermark 4096 of 4096, 4096 slots in the log" in run.stdout
    else:
        assert run.returncode == 77, run.stdout + run.stderr
        assert "no usable gfx950 device" in run.stdout


import pytest  # noqa: E402


@pytest.mark.gpu
def test_plain_c_callers_on_the_gpu():
    """the three examples again under `-m gpu`: on the MI355X box they run to the end and check their own results"""
    test_plain_c_caller_builds_links_and_fails_loudly_without_a_device()
    test_plain_c_epaxos_caller()
    test_plain_c_mencius_caller()
