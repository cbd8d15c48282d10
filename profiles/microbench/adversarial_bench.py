"""The parity / adversarial stream of SURVEY.md 8(d) at full size, timed (measurement aid): S = 2^20 slots x
256 acceptors, 64 epochs, leader changes with 25 % of the acceptors pre-promised (stale Phase2a's get Nacked),
5 % re-proposals after a change, target masks = random subsets of U[q-8, R] acceptors.  Everything is resident
in HBM and asynchronous: fused K3 through fpx_phase2_fused_dev (one launch per epoch: an epoch carries one round),
Phase1a through fpx_acceptor_phase1a_dev -- no host round trip inside the stream.  Three ways of driving it:
  eager      one C-ABI call per op, validated launches
  trusted    FPX_F_TRUSTED (the stream satisfies the run contract by construction: no validation kernels)
  graph      the whole stream captured once into a HIP graph on the context's stream and replayed
Prints proposals/s and chosen slots; chosen / Nacked counts must not depend on the driver."""
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
seed = 1
t0 = time.perf_counter()
script = W.adversarial_script(S, R, Q, seed, epochs=64, fused=True, subsets=W.fast_subsets)
gen = time.perf_counter() - t0
for ballot in (0, 1):
    for mode in ("eager", "trusted", "graph"):
        flags = fa.FPX_F_SCATTERED_TARGETS | (0 if mode == "eager" else fa.FPX_F_TRUSTED)
        ctx = fa.Context(fa.make_config(num_slots=S, num_replicas=R, f=Q - 1, ballot_mode=ballot, tally_ways=8,
                                        flags=flags))
        side = torch.cuda.Stream()
        ops = []
        proposals = 0
        for op in script:
            if op[0] == "phase1a":
                _, g, rnd, wm, tgt = op
                ops.append(("phase1a", g, rnd, wm,
                            None if tgt is None else torch.from_numpy(np.ascontiguousarray(tgt).view(np.int64)).to(dev)))
            else:
                _, slot, rr, val, tgt = op
                n = len(slot)
                proposals += n
                ops.append(("fused", torch.from_numpy(slot).to(dev), torch.from_numpy(rr).to(dev),
                            torch.from_numpy(val).to(dev), torch.from_numpy(tgt.view(np.int64)).to(dev),
                            torch.zeros(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.int32, device=dev),
                            torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)))

        def run():
            for op in ops:
                if op[0] == "phase1a":
                    ctx.acceptor_phase1a_dev(op[1], op[2], op[3], op[4])
                else:
                    ctx.phase2_fused_dev(*op[1:])

        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            ctx.set_stream(side.cuda_stream)
            run()                                   # warm-up: scratch buffers reach their size
            assert ctx.sync() == 0
            chosen0 = sum(int(op[5].sum().item()) for op in ops if op[0] == "fused")
            ctx.reset()
            assert ctx.sync() == 0
            if mode == "graph":
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    run()
                ctx.reset()
                assert ctx.sync() == 0
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                graph.replay()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            else:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            assert ctx.sync() == 0
        chosen = sum(int(op[5].sum().item()) for op in ops if op[0] == "fused")
        nacked = sum(int((op[8] >= 0).sum().item()) for op in ops if op[0] == "fused")
        assert chosen == chosen0
        print("ballot model %d %-8s: %d proposals in %d launches (+%d Phase1a), %.3f ms  %.3e proposals/s; "
              "%d chosen, %d Nacked (script generation on the CPU %.1f s)"
              % (ballot, mode, proposals, sum(1 for o in ops if o[0] == "fused"),
                 sum(1 for o in ops if o[0] == "phase1a"), dt * 1e3, proposals / dt, chosen, nacked, gen), flush=True)
        ctx.set_stream(None)
        del ctx, ops
