"""csrc/fpx_fastdiv.hpp -- slot / L and row / A by multiplication, which the kernels use for every slot -> (leader group,
row, acceptor group) split (phys_slot, group_of_slot: multipaxos/ProxyLeader.scala:190, mencius/ProxyLeader.scala:169-176,
231-234) -- compiled with g++ from the same source and held against the machine's division: every divisor up to 4096, the
powers of two and their neighbours up to 2^30, 20 000 random divisors, each on the edges of the 31-bit dividend range and
on random dividends; and every slot of a 2^22-slot window for the leader-group counts of BASELINE.json's configs."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "build")


def test_division_by_multiplication_is_exact():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "fastdiv_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "tests", "fastdiv_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "fastdiv ok" in out.stdout
