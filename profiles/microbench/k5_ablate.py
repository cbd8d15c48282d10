"""Ablation builds of the K5 key kernel: TIMING ONLY, the results are wrong.  The product sources carry no ablation
switches (hygiene): this script patches COPIES of csrc/ and builds one libfpx_<variant>.so each into
profiles/microbench/build/ (hipcc cross-compiles here; the .so files travel to the GPU box), and `run` times them with
k5v2_time.py.

  python profiles/microbench/k5_ablate.py build     # here
  python profiles/microbench/k5_ablate.py run       # on the GPU box
"""
import os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "frankenpaxos_amd", "csrc")
OUT = os.path.join(ROOT, "profiles", "microbench", "build")

# variant -> list of (file, old, new); every `old` must occur exactly once
V = {
    "base": [],
    "NOFIX": [("fpx_epaxos_kp.hpp", "for (uint32_t j = s0; j < s1; ++j) below += dst[j] < ec[cc] ? 1u : 0u;", "below = ea[cc];")],
    "NOSCAN": [("fpx_epaxos_kp.hpp", "if (q0 + cc * 64 < q1) scan_chunk<N>(valid, (fl >> 3) & 1, L, id1, cg, cs, ng, ns, dep);",
                "for (int l = 0; l < N; ++l) dep[l] = id1 + cg[l];")],
    "NODECIDE": [("fpx_epaxos_kp.hpp", "    for (int sl = threadIdx.x; sl < c; sl += T::THREADS) {\n      const int i = RF(0, sl), x = RF(1, sl), L = RF(2, sl) & 7;",
                  "    for (int sl = threadIdx.x; sl < c && a.m < 0; sl += T::THREADS) {\n      const int i = RF(0, sl), x = RF(1, sl), L = RF(2, sl) & 7;")],
    "NOOUT": [("fpx_epaxos_kp.hpp", "      for (int t = threadIdx.x; t < c * Q; t += T::THREADS) {", "      for (int t = threadIdx.x; t < c * Q && a.m < 0; t += T::THREADS) {")],
    "NOMINMAX": [("fpx_epaxos_kp.hpp", "        if (lane == 0 && mn <= mx) atomicMin(&rmin[q], mn), atomicMax(&rmax[q], mx);\n",
                  "        (void)mx, (void)mn;\n"),
                 ("fpx_epaxos_kp.hpp", "        const int mx = __builtin_amdgcn_readlane(wave_incl_max(hi[q]), 63);\n        const int mn = 0x7fffffff - __builtin_amdgcn_readlane(wave_incl_max(0x7fffffff - lo[q]), 63);\n",
                  "        const int mx = hi[q], mn = lo[q];\n        if (threadIdx.x == 0) rmin[q] = 0, rmax[q] = a.m - 1;\n")],
    "NOPUTS": [("fpx_epaxos_kp.hpp", "            atomicMax(&tot[(r * T::W + (int)((s0 + below) / (uint32_t)perq)) * 2 * N + ((fl >> 3) & 1) * N + (fl & 7)], RF(1, sl) + 1);",
                "            if (a.m < 0) atomicMax(&tot[fl & 7], RF(1, sl) + 1);")],
    "NOROWS": [("fpx_epaxos_kp.hpp", "          for (int l = 0; l < N; ++l) rows[sl * T::RSTR + ri * N + l] = dep[l];",
                "          for (int l = 0; l < N; ++l) if (dep[l] == -77) rows[sl * T::RSTR + ri * N + l] = dep[l];")],
}
# tuning variants (results stay correct)
V["HG4"] = [("fpx_epaxos_kp.hpp", "constexpr int KP_HG = 8;", "constexpr int KP_HG = 4;")]
V["HG16"] = [("fpx_epaxos_kp.hpp", "constexpr int KP_HG = 8;", "constexpr int KP_HG = 16;"),
             ("fpx_epaxos_kp.hpp", "__launch_bounds__(128 * KP_HG) k_kp_hist", "__launch_bounds__(64 * KP_HG) k_kp_hist"),
             ("fpx_epaxos_kp.hpp", "  const int sub = threadIdx.x >> 7, t = threadIdx.x & 127;", "  const int sub = threadIdx.x >> 6, t = threadIdx.x & 63;"),
             ]


def build():
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for name, patches in V.items():
        if len(sys.argv) > 2 and name not in sys.argv[2:]:
            continue
        d = tempfile.mkdtemp(prefix="k5abl_" + name)
        shutil.copytree(SRC, os.path.join(d, "csrc"), ignore=shutil.ignore_patterns("*.o", "*.so"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(d, "include"))
        for f, old, new in patches:
            p = os.path.join(d, "csrc", f)
            s = open(p).read()
            assert s.count(old) == 1, (name, f, old[:50], s.count(old))
            open(p, "w").write(s.replace(old, new))
        # the copy sits one level lower than csrc does in the repo: ../../include -> ../include
        for f in os.listdir(os.path.join(d, "csrc")):
            p = os.path.join(d, "csrc", f)
            if not os.path.isfile(p):
                continue
            s = open(p).read()
            if "../../include/" in s:
                open(p, "w").write(s.replace("../../include/", "../include/"))
        cmd = ("cd %s/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-result -Wno-unused-value -c -o epx.o fpx_epaxos.hip && "
               "/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o %s/libfpx_%s.so %s/fpx_api.o epx.o %s/fpx_wire.o %s/fpx_depgraph.o -ldl"
               % (d, OUT, name, SRC, SRC, SRC))
        procs.append((name, subprocess.Popen(cmd, shell=True)))
    for name, p in procs:
        assert p.wait() == 0, name
    print(sorted(os.listdir(OUT)))


def run():
    os.makedirs(os.path.join(ROOT, "gpurun_out", "k5abl"), exist_ok=True)
    out = open(os.path.join(ROOT, "gpurun_out", "k5abl", "variants.txt"), "w")
    for name in V:
        so = os.path.join(OUT, "libfpx_%s.so" % name)
        if not os.path.exists(so):
            continue
        env = dict(os.environ, FPX_LIB=so, K5_MODES="packed")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "microbench", "k5v2_time.py"), name], env=env,
                           capture_output=True, text=True, timeout=300)
        for line in r.stdout.splitlines():
            if "ms per tick" in line:
                print(line), out.write(line + "\n")
        if r.returncode:
            print(name, "rc", r.returncode, r.stderr[-300:])
        d = "/tmp/k5abl_prof_" + name
        subprocess.run("cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d %s -o k5 -- %s %s %s > /dev/null 2>&1"
                       % (d, sys.executable, os.path.join(ROOT, "profiles", "microbench", "k5v2_time.py"), name), shell=True, env=env, timeout=300)
        import csv, glob
        for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
            row = ["%s %.1f" % (("key2" if "k_epx_key2" in x["Name"] else "scatter" if "k_kp_scatter" in x["Name"] else "hist"), float(x["AverageNs"]) / 1e3)
                   for x in csv.DictReader(open(f)) if "k_epx_key2" in x["Name"] or "k_kp_" in x["Name"]]
            line = "%s kernels (us, rocprofv3 average): %s" % (name, "  ".join(sorted(row)))
            print(line), out.write(line + "\n")


if __name__ == "__main__":
    (build if (len(sys.argv) < 2 or sys.argv[1] == "build") else run)()
