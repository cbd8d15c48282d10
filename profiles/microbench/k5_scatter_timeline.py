"""DIAGNOSTIC (FPX_LIB = a build of profiles/microbench: k_kp_scatter stamps wall_clock64 at its phase boundaries into the
packed output, k_epx_key2 returns at once): where a scatter workgroup's 30 us go.  One tick of BASELINE configs[3]."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from frankenpaxos_amd.epaxos import EPaxos
from tests import workloads as W
from tests.workloads import random_tick
n, num_keys, m = 5, 1024, 1 << 20
dev = torch.device("cuda:0")
rng = np.random.default_rng(4)
tick = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in random_tick(rng, n, num_keys, m, [0] * n, 64.0)]
for rep in range(3):
    epx = EPaxos(n, num_keys)
    epx.set_stream(torch.cuda.current_stream().cuda_stream)
    packed = torch.zeros((m, epx.packed_stride()), dtype=torch.int32, device=dev)
    epx.preaccept_packed_dev(*tick, packed)
    torch.cuda.synchronize()
    t = packed.view(torch.int64).flatten()[: 512 * 8].cpu().numpy().reshape(512, 8).astype(np.float64)
    t = (t - t[:, 0].min()) / 100.0   # us (100 MHz)
    names = ["start", "tables", "loads+count", "scan", "staged", "stores issued", "stores done", "end"]
    print("rep", rep, "| us since the first workgroup started: min / median / max over the 512 workgroups")
    for q, nm in enumerate(names):
        print("  %-14s %7.2f %7.2f %7.2f" % (nm, t[:, q].min(), np.median(t[:, q]), t[:, q].max()))
