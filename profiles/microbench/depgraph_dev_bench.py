"""Device dependency-graph execution (fpx_epx_execute_dev, csrc/fpx_depgraph_dev.hpp) on what a K5 tick commits:
BASELINE.json configs[3] -- n = 5 replicas, 1024 keys, a 2^20-command tick pre-accepted on the GPU (K5), every command
committed with its (leader, number, dependencies) triple, then executed.  Wall time of the call (it ends with the counts
on the host), best and median of REPS, for FIFO channels and for reordering ones (own-column explicit ids take part).

    python profiles/microbench/depgraph_dev_bench.py [log2 m]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frankenpaxos_amd.epaxos import EPaxos
from tests import workloads as W
from tests.workloads import random_tick

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
REPS = 10
n, num_keys, m = 5, 1024, 1 << lg
dev = torch.device("cuda:0")
for fifo in (True, False):
    epx = EPaxos(n, num_keys)
    rng = np.random.default_rng(45)
    nxt = [0] * n
    leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 64.0, fifo=fifo)
    key = (W.splitmix64_at(np.arange(m, dtype=np.uint64)) % np.uint64(num_keys)).astype(np.int32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dl, dn = t(leader), t(number)
    packed = torch.zeros((m, epx.packed_stride()), dtype=torch.int32, device=dev)
    epx.preaccept_packed_dev(dl, dn, t(key), t(is_set), t(mask), t(rank), packed)
    assert epx.sync() == 0
    order, comp = torch.zeros(m, dtype=torch.int32, device=dev), torch.zeros(m, dtype=torch.int32, device=dev)
    first, count = np.zeros(n, np.int32), np.asarray(nxt, np.int32)
    times = []
    for rep in range(REPS + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ne, nc, nh = epx.execute_dev(dl, dn, packed, first, count, order, comp)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    times = sorted(times[1:])
    assert ne == m and not nh
    print("m = 2^%d  %-10s %8d commands  %7d components  best %.3f ms  median %.3f ms  -> %.2fe9 / %.2fe9 commands/s executed"
          % (lg, "fifo" if fifo else "reordered", ne, nc, times[0] * 1e3, times[len(times) // 2] * 1e3,
             ne / times[0] / 1e9, ne / times[len(times) // 2] / 1e9))
