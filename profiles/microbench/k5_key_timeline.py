"""DIAGNOSTIC (FPX_LIB = a build of profiles/microbench: thread 0 of every k_epx_key2 workgroup stamps wall_clock64 behind
every barrier of a key into rows behind the packed output): where a key's ~16 us go.  One tick of BASELINE configs[3]."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from frankenpaxos_amd.epaxos import EPaxos
from tests.workloads import random_tick
n, num_keys, m = 5, 1024, 1 << 20
dev = torch.device("cuda:0")
rng = np.random.default_rng(4)
tick = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in random_tick(rng, n, num_keys, m, [0] * n, 64.0)]
for rep in range(3):
    epx = EPaxos(n, num_keys)
    epx.set_stream(torch.cuda.current_stream().cuda_stream)
    stride = epx.packed_stride()
    full = torch.zeros((m + 8192, stride), dtype=torch.int32, device=dev)
    epx.preaccept_packed_dev(*tick, full[:m])
    torch.cuda.synchronize()
    t = full[m:].reshape(-1).view(torch.int64)[: 256 * 4 * 16].cpu().numpy().reshape(256, 4, 16).astype(np.float64)
    cnt = (t > 0).sum(axis=2)
    print("rep", rep, "stamps per key:", np.unique(cnt))
    t0 = t[:, 0, 0].min()
    names = ["key start", "unpacked", "counted", "bucket starts", "placed", "fixed up (sorted)", "first chunk scanned", "scanned", "decided", "key done"]
    d = np.diff(t, axis=2) / 100.0   # us per phase (100 MHz)
    for ki in range(4):
        print("  key %d of a workgroup: starts at %.1f us (median); phases in us, median over 256 workgroups:" % (ki, np.median(t[:, ki, 0] - t0) / 100.0))
        print("    " + "  ".join("%s %.2f" % (names[q + 1], np.median(d[:, ki, q])) for q in range(9)))
    print("  a key start to start: %.2f us; the kernel's last stamp at %.1f us" % (np.median(np.diff(t[:, :, 0], axis=1)) / 100.0, (t.max() - t0) / 100.0))
