"""Variants of the K5 kernels that round 4's timelines suggest and round 4 had no GPU minutes left to measure
(profiles/r04_k5.md, "What the timelines leave to try").  Each is a patch on a COPY of csrc/, built here into
profiles/microbench/build/libfpx_k5<name>.so; time them on one box against `base` and run the EPaxos tests on the one
that wins:

    python profiles/microbench/r05_k5_variants.py                                      # here: builds base + variants
    K5_VARIANTS="base fpr2" bash profiles/microbench/r04_k5_ab2.sh                     # on the GPU box: ms per tick
    FPX_LIB=$PWD/profiles/microbench/build/libfpx_k5fpr2.so python -m pytest tests/test_epaxos.py tests/test_epaxos_models.py -m gpu -q

  fpr2   k_kp_scatter adds its fingerprint words up through LDS columns beside the scan (as round 4's `fpr`, which lost
         15 us because its 64-bit atomics left in the middle of the kernel) and issues the atomics as the kernel's LAST
         instructions: the 4.5 us butterfly at the end of every scatter workgroup goes, the atomics stay where they cost nothing
"""
import os, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "frankenpaxos_amd", "csrc")
OUT = os.path.join(ROOT, "profiles", "microbench", "build")


def rep(s, old, new):
    assert s.count(old) == 1, (old[:60], s.count(old))
    return s.replace(old, new)


def fpr2(s):
    # 1. park the threads' sums in the staging area before the barrier behind the counts
    s = rep(s, "    at[u] = atomicAdd(&cnt[k], 1u);\n  }\n  __syncthreads();\n",
            "    at[u] = atomicAdd(&cnt[k], 1u);\n  }\n"
            "  constexpr int FW = 2 * (N + 1);\n"
            "  static_assert((size_t)FW * 8 * S::THREADS <= (size_t)S::SW * T::NI * 4 && FW * 32 <= S::THREADS, \"fingerprint columns fit the staging area\");\n"
            "  unsigned long long* fcol = reinterpret_cast<unsigned long long*>(stage);  // [FW][THREADS]\n"
            "#pragma unroll\n  for (int q = 0; q < FW; ++q) fcol[q * S::THREADS + threadIdx.x] = f[q];\n"
            "  __syncthreads();\n")
    # 2. column sums beside the scan (before the barrier that lets the records overwrite the staging area)
    s = rep(s, "      if (mine) cnt[threadIdx.x * E + j] = run, run += c[j];\n  }\n  __syncthreads();\n  int staged = 0;",
            "      if (mine) cnt[threadIdx.x * E + j] = run, run += c[j];\n  }\n"
            "  unsigned long long fword = 0ull;  // word threadIdx.x >> 5 of the workgroup, in the lanes with (threadIdx.x & 31) == 0\n"
            "  if (threadIdx.x < FW * 32) {\n"
            "    const int q = threadIdx.x >> 5, part = threadIdx.x & 31;\n"
            "#pragma unroll\n    for (int j = 0; j < S::THREADS / 32; ++j) fword += fcol[q * S::THREADS + part + 32 * j];\n"
            "#pragma unroll\n    for (int o = 16; o > 0; o >>= 1) fword += __shfl_xor(fword, o);\n"
            "  }\n  __syncthreads();\n  int staged = 0;")
    # 3. the atomics stay the kernel's last instructions
    i0 = s.index("  // the fingerprints: wavefront sums, then one 64-bit atomic per workgroup and word\n")
    i1 = s.index("// LOG: the command log is kept (its own instantiation")
    s = s[:i0] + "  if (threadIdx.x < FW * 32 && (threadIdx.x & 31) == 0) atomicAdd(&a.fp[threadIdx.x >> 5], fword);\n}\n\n" + s[i1:]
    return rep(s, "  __shared__ unsigned long long fsum[S::THREADS / 64][2 * (N + 1)];\n", "")


VARIANTS = {"base": lambda s: s, "fpr2": fpr2}


def build(name, patch):
    d = tempfile.mkdtemp(prefix="k5" + name)
    shutil.copytree(SRC, os.path.join(d, "csrc"), ignore=shutil.ignore_patterns("*.o", "*.so", ".*"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(d, "include"))
    for f in os.listdir(os.path.join(d, "csrc")):
        p = os.path.join(d, "csrc", f)
        if not os.path.isfile(p):
            continue
        s = open(p).read()
        if f == "fpx_epaxos_kp.hpp":
            s = patch(s)
        open(p, "w").write(s.replace("../../include/", "../include/"))
    os.makedirs(OUT, exist_ok=True)
    cmd = ("cd %s/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Wno-unused-result -c -o epx.o fpx_epaxos.hip && "
           "/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o %s/libfpx_k5%s.so %s/fpx_api.o epx.o %s/fpx_wire.o %s/fpx_depgraph.o -ldl"
           % (d, OUT, name, SRC, SRC, SRC))
    return subprocess.Popen(cmd, shell=True)


if __name__ == "__main__":
    names = sys.argv[1:] or list(VARIANTS)
    procs = [(n, build(n, VARIANTS[n])) for n in names]
    bad = [n for n, p in procs if p.wait() != 0]
    assert not bad, bad
    print(sorted(f for f in os.listdir(OUT) if f.endswith(".so")))
