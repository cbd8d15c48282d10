"""How long does one fused step (k_phase2 + k_finalize) take as a function of the batch size, R = 3, fresh slots every
call?  HIP events around each call on the context's stream; the median of the calls of one size."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import frankenpaxos_amd as fa  # noqa: E402

dev = torch.device("cuda:0")
sizes = [int(x) for x in os.environ["SIZES"].split(",")] if os.environ.get("SIZES") else [1 << k for k in (10, 12, 14, 16, 18, 20)]
reps = 9
mode = int(os.environ.get("BALLOT", "1"))
S = sum(sizes) * reps
ctx = fa.Context(fa.make_config(num_slots=S, num_replicas=3, num_groups=1, ballot_mode=mode, tally_ways=4, f=1,
                                flags=fa.FPX_F_TRUSTED))
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
assert ctx.acceptor_phase1a(0, 0)[0] == 0
ctx.flush_promises()
base = 0
for n in sizes:
    ts = []
    for r in range(reps):
        slot = torch.arange(base, base + n, dtype=torch.int32, device=dev)
        base += n
        rnd, val = torch.zeros_like(slot), slot * 3
        ch = torch.zeros(n, dtype=torch.uint8, device=dev)
        cr, cv = torch.zeros_like(slot), torch.zeros_like(slot)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ctx.phase2_fused_dev(slot, rnd, val, None, ch, cr, cv)
        b.record()
        torch.cuda.synchronize()
        assert bool(ch.all())
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print("n = %8d   fused step %7.1f us (median of %d; min %.1f)   %.2f ns per slot" % (n, ts[reps // 2], reps, ts[0], ts[reps // 2] * 1e3 / n))
