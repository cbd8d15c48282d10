"""Builds the two DIAGNOSTIC libraries the timeline scripts load (results of a tick are wrong or missing: timing only):

  libfpx_k5tl.so   k_kp_scatter stamps wall_clock64 at its phase boundaries into the packed output (thread 0 of every
                   workgroup, eight stamps), k_epx_key2 returns at once          -> k5_scatter_timeline.py
  libfpx_k5tl2.so  thread 0 of every k_epx_key2 workgroup stamps behind every barrier of a key into rows behind the
                   packed output (the script allocates them)                     -> k5_key_timeline.py

Patches a COPY of csrc/ (the product sources carry no diagnostics), hipcc cross-compiles here, the .so files travel to
the GPU box with the snapshot:  python profiles/microbench/k5_timeline_build.py"""
import os, shutil, subprocess, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "frankenpaxos_amd", "csrc")
OUT = os.path.join(ROOT, "profiles", "microbench", "build")


def rep(s, old, new):
    assert s.count(old) == 1, (old[:60], s.count(old))
    return s.replace(old, new)


def scatter_stamps(s):
    s = rep(s, "  const int tile = unit * S::TILES;\n", "  const int tile = unit * S::TILES;\n  long long tl[8]; tl[0] = wall_clock64();\n")
    lines = s.split("\n")
    at = [i for i, l in enumerate(lines) if "const int tile = unit * S::TILES;" in l][0]
    bars = [i for i in range(at, len(lines)) if lines[i].strip() == "__syncthreads();"][:6]
    for b, t in ((bars[0], 1), (bars[1], 2), (bars[3], 3), (bars[4], 4)):  # tables, counts, scan, staged
        lines[b] += " tl[%d] = wall_clock64();" % t
    s = "\n".join(lines)
    s = rep(s, "  // the fingerprints: wavefront sums, then one 64-bit atomic per workgroup and word\n",
            "  tl[5] = wall_clock64();\n  __builtin_amdgcn_s_waitcnt(0);\n  tl[6] = wall_clock64();\n"
            "  // the fingerprints: wavefront sums, then one 64-bit atomic per workgroup and word\n")
    s = rep(s, "    atomicAdd(&a.fp[threadIdx.x], v);\n  }\n}",
            "    atomicAdd(&a.fp[threadIdx.x], v);\n  }\n  tl[7] = wall_clock64();\n  if (threadIdx.x == 0 && a.packed) for (int q = 0; q < 8; ++q) "
            "reinterpret_cast<long long*>(a.packed)[(size_t)blockIdx.x * 8 + q] = tl[q];\n}")
    return rep(s, "  auto give_up = [&]() {", "  if (a.m > 0) return;\n  auto give_up = [&]() {")


def key_stamps(s):
    lines = s.split("\n")
    i0 = [i for i, l in enumerate(lines) if "for (; k < st.num_keys; k += gridDim.x) {" in l][0]
    i1 = [i for i, l in enumerate(lines) if l.strip() == "cur = nxt;"][0]
    out = lines[:i0 + 1]
    out.append("    long long* dbg = reinterpret_cast<long long*>(a.packed + (size_t)a.m * a.stride) + ((size_t)blockIdx.x * 4 + (size_t)(k / gridDim.x)) * 16; int tn = 0;")
    out.append("    if (threadIdx.x == 0) dbg[tn++] = wall_clock64();")
    for l in lines[i0 + 1:i1]:
        out.append(l)
        if l.strip().startswith("__syncthreads();"):
            out.append("    if (threadIdx.x == 0 && tn < 16) dbg[tn++] = wall_clock64();")
    out.append("    if (threadIdx.x == 0) { for (; tn < 16; ++tn) dbg[tn] = 0; }")
    return "\n".join(out + lines[i1:])


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
    cmd = ("cd %s/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-result -c -o epx.o fpx_epaxos.hip && "
           "/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o %s/libfpx_k5%s.so %s/fpx_api.o epx.o %s/fpx_wire.o %s/fpx_depgraph.o -ldl"
           % (d, OUT, name, SRC, SRC, SRC))
    return subprocess.Popen(cmd, shell=True)


if __name__ == "__main__":
    procs = [build("tl", scatter_stamps), build("tl2", key_stamps)]
    assert all(p.wait() == 0 for p in procs)
    print(sorted(f for f in os.listdir(OUT) if "k5tl" in f))
