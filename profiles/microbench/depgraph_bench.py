"""Host throughput of dependency-graph execution (include/fpx_depgraph.h) on what a K5 tick commits:
ticks of m commands (n = 5 replicas, 1024 keys, 50 % sets) from the CPU oracle, every command committed with its
(leader, number, dependencies) triple, then executeByComponent.  One core; no GPU involved (the library is host code)."""
import os
import resource
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
    for t in range(4):
        leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 64.0)
        out = ref.preaccept(leader, number, key, is_set, mask, rank)
        assert out[0] == 0
        ticks.append((leader, number, out))
    for kind, name in ((FPX_DG_ZIGZAG, "zigzag"), (FPX_DG_TARJAN, "tarjan")):
        if kind == FPX_DG_TARJAN and m > (1 << 14):
            continue   # visits every edge of the prefix-shaped dependency sets, like the reference's: quadratic in the tick
        g = DependencyGraph(n, kind=kind)
        for phase, part in (("first two ticks (the graph's pools grow: page faults)", ticks[:2]), ("next two ticks (pools warm)", ticks[2:])):
            t_commit = t_exec = 0.0
            done = 0
            ru0 = resource.getrusage(resource.RUSAGE_SELF)
            for leader, number, out in part:
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
            ru1 = resource.getrusage(resource.RUSAGE_SELF)
            user, sys_ = ru1.ru_utime - ru0.ru_utime, ru1.ru_stime - ru0.ru_stime
            print("m = %8d  %-6s  %-54s commit %7.1f ms  execute %7.1f ms  -> %.2fe6 commands/s executed (%d of %d), 1 core; "
                  "user %.0f ms + system %.0f ms (page faults of fresh result arrays and growing pools): %.1fe6 commands/s of user time"
                  % (m, name, phase, t_commit * 1e3, t_exec * 1e3, done / total / 1e6, done, 2 * m, user * 1e3, sys_ * 1e3, done / max(user, 1e-9) / 1e6))
        g.close()
