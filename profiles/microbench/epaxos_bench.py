"""K5 throughput on BASELINE.json configs[3] (measurement aid): EPaxos n = 5, 2^20 commands per tick,
keys = splitmix64(i) % 1024, 50 % sets, every replica sees the tick in the global order perturbed by a
replica-specific skew.  Inputs resident in HBM; prints commands/s and the fast-path fraction."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from frankenpaxos_amd.epaxos import EPaxos
from tests import workloads as W

n, K, m = 5, 1024, 1 << 20
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
epx = EPaxos(n, K)
epx.set_stream(torch.cuda.current_stream().cuda_stream)
ticks = []
nxt = np.zeros(n, np.int64)
for t in range(6):
    leader = rng.integers(0, n, m).astype(np.int32)
    number = np.zeros(m, np.int32)
    for l in range(n):
        idx = np.nonzero(leader == l)[0]
        number[idx] = nxt[l] + np.arange(len(idx))
        nxt[l] += len(idx)
    key = (W.splitmix64_at(np.arange(t * m, (t + 1) * m, dtype=np.uint64)) % np.uint64(K)).astype(np.int32)
    is_set = (rng.random(m) < 0.5).astype(np.uint8)
    drop = rng.integers(0, n - 1, m)
    others = np.stack([(leader + 1 + j) % n for j in range(n - 1)], 1)
    mask = np.zeros(m, np.uint8)
    for j in range(n - 1):
        mask |= np.where(drop != j, (1 << others[:, j]).astype(np.uint8), 0).astype(np.uint8)
    rank = np.zeros((n, m), np.int32)
    for r in range(n):
        rank[r, np.argsort(np.arange(m) + rng.normal(0, 200.0, m), kind="stable")] = np.arange(m)
    ticks.append([torch.from_numpy(x).to(dev) for x in (leader, number, key, is_set, mask, rank)])
fast = torch.empty(m, dtype=torch.uint8, device=dev)
deps = torch.empty((m, n), dtype=torch.int32, device=dev)
epx.preaccept_dev(*ticks[0], fast, deps, None)
assert epx.sync() == 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(1, 6):
    epx.preaccept_dev(*ticks[t], fast, deps, None)
assert epx.sync() == 0
dt = (time.perf_counter() - t0) / 5
print("EPaxos K5: n=%d keys=%d  %d commands/tick  %.3f ms/tick  %.3e commands/s  fast-path fraction %.3f"
      % (n, K, m, dt * 1e3, m / dt, float(fast.float().mean())))
