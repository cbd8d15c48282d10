"""times the K5 tick (BASELINE.json configs[3] workload) on the library FPX_LIB names: packed lines and the four arrays"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from frankenpaxos_amd.epaxos import EPaxos
from tests import workloads as W
from tests.workloads import random_tick

n, num_keys, m = int(os.environ.get("K5_N", "5")), 1024, 1 << 20
dev = torch.device("cuda:0")
epx = EPaxos(n, num_keys)
epx.set_stream(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(4)
nxt = [0] * n
T = int(os.environ.get("K5_T", "8"))
ticks = []
for t in range(T):
    leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 64.0)
    key = (W.splitmix64_at(np.arange(t * m, (t + 1) * m, dtype=np.uint64)) % np.uint64(num_keys)).astype(np.int32)
    ticks.append([torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (leader, number, key, is_set, mask, rank)])
torch.cuda.synchronize()
packed = torch.zeros((m, epx.packed_stride()), dtype=torch.int32, device=dev)
fast = torch.zeros(m, dtype=torch.uint8, device=dev)
deps, ldeps = (torch.zeros((m, n), dtype=torch.int32, device=dev) for _ in range(2))
own = torch.zeros((m, 2), dtype=torch.int32, device=dev)
for mode in os.environ.get("K5_MODES", "packed,arrays").split(","):
    ms = []
    for t in range(T):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if mode == "packed":
            epx.preaccept_packed_dev(*ticks[t], packed)
        else:
            epx.preaccept_dev(*ticks[t], fast, deps, ldeps, own_values_end=own)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    print("%s %s: ms per tick %s  (mean of the last 5: %.4f)" % (sys.argv[1] if len(sys.argv) > 1 else "", mode, " ".join("%.3f" % x for x in ms), float(np.mean(ms[3:]))), flush=True)
for mode in os.environ.get("K5_MODES", "packed,arrays").split(","):
    if mode != "packed":
        continue
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 5
    for rep in range(R):
        for t in range(T):
            epx.preaccept_packed_dev(*ticks[t], packed)
    torch.cuda.synchronize()
    print("%s packed, %d ticks back to back: %.4f ms per tick (wall)" % (sys.argv[1] if len(sys.argv) > 1 else "", R * T, (time.perf_counter() - t0) * 1e3 / (R * T)), flush=True)
    if os.environ.get("K5_ROTATE_OUT"):
        # the same ticks, every one into an output buffer of its own (T x 64 MB: more than the 256 MB memory-side cache holds)
        outs = [torch.zeros((m, epx.packed_stride()), dtype=torch.int32, device=dev) for _ in range(T)]
        for t in range(T):
            epx.preaccept_packed_dev(*ticks[t], outs[t])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(R):
            for t in range(T):
                epx.preaccept_packed_dev(*ticks[t], outs[t])
        torch.cuda.synchronize()
        print("%s packed, %d ticks back to back, %d output buffers in turn: %.4f ms per tick (wall)" % (sys.argv[1] if len(sys.argv) > 1 else "", R * T, T, (time.perf_counter() - t0) * 1e3 / (R * T)), flush=True)
rc = epx.sync()
assert rc == 0 or "FPX_LIB" in os.environ, rc
