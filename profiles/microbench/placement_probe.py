"""Does the physical placement of the cell slab explain the two speeds of the headline kernel (0.565 vs 0.60 ms
per 2^20-slot step; the FIRST process on a fresh box lands in the slow mode, r02)?  One process: create the
bench's 23-window context, time 20 steps, destroy it, repeat -- optionally with a large throw-away allocation
first.  Measurement aid."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import frankenpaxos_amd as fa

S, R, F, W = 1 << 20, 256, 127, 23
dev = torch.device("cuda:0")


def one(tag):
    ctx = fa.Context(fa.make_config(num_slots=W * S, num_replicas=R, f=F, ballot_mode=1, tally_ways=4,
                                    flags=fa.FPX_F_TRUSTED))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.acceptor_phase1a(0, 0)
    ctx.flush_promises()
    ch = torch.zeros(S, dtype=torch.uint8, device=dev)
    steps = []
    for w in range(W):
        slot = torch.arange(w * S, (w + 1) * S, dtype=torch.int32, device=dev)
        steps.append((slot, torch.zeros_like(slot), slot * 5 + 1))
    for w in range(3):
        ctx.phase2_fused_dev(*steps[w], None, ch, None, None)
    assert ctx.sync() == 0
    ctx.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for w in range(3, W):
        ctx.phase2_fused_dev(*steps[w], None, ch, None, None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (W - 3)
    n, ms = ctx.profile_read()
    print("%-28s %.4f ms/step  kernel %.4f ms  (%.3e slots/s)" % (tag, dt * 1e3, ms / n, S / dt), flush=True)
    ctx.close()
    del steps, ch
    torch.cuda.empty_cache()


one("1st context of the process")
one("2nd context")
big = torch.empty(200 << 30, dtype=torch.uint8, device=dev)   # touch-free throw-away allocation
del big
torch.cuda.empty_cache()
one("after a 200 GiB alloc+free")
one("4th context")
