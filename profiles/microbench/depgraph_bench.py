"""Host throughput of dependency-graph execution (include/fpx_depgraph.h) on what a K5 tick commits:
ticks of m commands (n = 5 replicas, 1024 keys, 50 % sets) from the CPU oracle, every command committed with its
(leader, number, dependencies) triple, then executeByComponent.  One core; no GPU involved (the library is host code)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle.pyoracle as oracle  # noqa: E402
from frankenpaxos_amd.depgraph import FPX_DG_TARJAN, FPX_DG_ZIGZAG, DependencyGraph  # noqa: E402
from tests.workloads import random_tick  # noqa: E402

oracle.build()
n, num_keys = 5, 1024
for m in [int(x) for x in os.environ.get("M", "16384,131072,1048576").split(",")]:
    rng = np.random.default_rng(3)
    ref = oracle.EPaxos(n, num_keys)
    nxt = [0] * n
    ticks = []
    for t in range(2):
        leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 64.0)
        out = ref.preaccept(leader, number, key, is_set, mask, rank)
        assert out[0] == 0
        ticks.append((leader, number, out))
    for kind, name in ((FPX_DG_ZIGZAG, "zigzag"), (FPX_DG_TARJAN, "tarjan")):
        if kind == FPX_DG_TARJAN and m > (1 << 14):
            continue   # visits every edge of the prefix-shaped dependency sets, like the reference's: quadratic in the tick
        g = DependencyGraph(n, kind=kind)
        t_commit = t_exec = 0.0
        done = 0
        for leader, number, out in ticks:
            # the slow path's union is what gets committed for commands off the fast path: outputs 2 (deps) / 4 (own end)
            deps, own_end = out[2], out[4]
            t0 = time.perf_counter()
            g.commit_epx(leader, number, deps, own_end)
            t1 = time.perf_counter()
            ex = g.execute_arrays()
            t2 = time.perf_counter()
            t_commit += t1 - t0
            t_exec += t2 - t1
            done += len(ex[0])
        total = t_commit + t_exec
        print("m = %8d  %-6s  commit %7.1f ms  execute %7.1f ms  -> %.2fe6 commands/s executed (%d of %d), 1 core"
              % (m, name, t_commit * 1e3, t_exec * 1e3, done / total / 1e6, done, 2 * m))
        g.close()
