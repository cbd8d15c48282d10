"""The parity / adversarial stream of SURVEY.md 8(d) at full size, timed (measurement aid): S = 2^20 slots x
256 acceptors, 64 epochs, leader changes with 25 % of the acceptors pre-promised (stale Phase2a's get Nacked),
5 % re-proposals after a change, target masks = random subsets of U[q-8, R] acceptors.  Fused K3 through the
device entry points, one launch per epoch (an epoch carries one round), Phase1a between epochs; the batches are
resident in HBM, the per-epoch launches are validated (no FPX_F_TRUSTED).  Prints proposals/s and chosen slots."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import frankenpaxos_amd as fa
from tests import workloads as W

S, R, Q = 1 << 20, 256, 128
dev = torch.device("cuda:0")
for seed in (1,):
    t0 = time.perf_counter()
    script = W.adversarial_script(S, R, Q, seed, epochs=64, fused=True)
    gen = time.perf_counter() - t0
    for ballot in (0, 1):
        ctx = fa.Context(fa.make_config(num_slots=S, num_replicas=R, f=Q - 1, ballot_mode=ballot, tally_ways=8,
                                        flags=fa.FPX_F_SCATTERED_TARGETS))
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ops = []
        proposals = 0
        for op in script:
            if op[0] == "phase1a":
                ops.append(op)
            else:
                _, slot, rr, val, tgt = op
                n = len(slot)
                proposals += n
                ops.append(("fused", torch.from_numpy(slot).to(dev), torch.from_numpy(rr).to(dev),
                            torch.from_numpy(val).to(dev), torch.from_numpy(tgt.view(np.int64)).to(dev),
                            torch.zeros(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.int32, device=dev),
                            torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for op in ops:
            if op[0] == "phase1a":
                _, g, rnd, wm, tgt = op
                assert ctx.acceptor_phase1a(g, rnd, wm, tgt)[0] == 0
            else:
                ctx.phase2_fused_dev(*op[1:])
        assert ctx.sync() == 0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        chosen = sum(int(op[5].sum().item()) for op in ops if op[0] == "fused")
        nacked = sum(int((op[8] >= 0).sum().item()) for op in ops if op[0] == "fused")
        print("ballot model %d seed %d: %d proposals in %d launches (+%d Phase1a), %.3f ms  %.3e proposals/s; "
              "%d chosen, %d Nacked (script generation on the CPU %.1f s)"
              % (ballot, seed, proposals, sum(1 for o in ops if o[0] == "fused"),
                 sum(1 for o in ops if o[0] == "phase1a"), dt * 1e3, proposals / dt, chosen, nacked, gen))
        del ctx, ops
