"""config 2's step (65 536 slots x 3 acceptors, fused) by ballot model and by the stream it is launched on"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import frankenpaxos_amd as fa
from bench import steady_values_torch

dev = torch.device("cuda:0")
n, R, f, K = 65536, 3, 1, 40
for ballot in (0, 1):
    for which in ("null", "side", "own"):
        ctx = fa.Context(fa.make_config(num_slots=(K + 4) * n, num_replicas=R, f=f, tally_ways=4, ballot_mode=ballot, flags=fa.FPX_F_TRUSTED))
        side = torch.cuda.Stream()
        steps = []
        for w in range(K + 4):
            slot = torch.arange(w * n, (w + 1) * n, dtype=torch.int32, device=dev)
            steps.append((slot, torch.zeros_like(slot), steady_values_torch(slot), torch.zeros(n, dtype=torch.uint8, device=dev),
                          torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)))
        torch.cuda.synchronize()
        if which == "null":
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        elif which == "side":
            ctx.set_stream(side.cuda_stream)
        assert ctx.acceptor_phase1a(0, 0)[0] == 0
        ctx.flush_promises()
        for i in range(4):
            ctx.phase2_fused_dev(steps[i][0], steps[i][1], steps[i][2], None, *steps[i][3:])
        assert ctx.sync() == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(4, 4 + K):
            ctx.phase2_fused_dev(steps[i][0], steps[i][1], steps[i][2], None, *steps[i][3:])
        assert ctx.sync() == 0
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        ok = all(bool(s[3].all()) for s in steps)
        print("ballot model %d, %-4s stream: %.4f ms per step (%.3e slots/s); all chosen: %s" % (ballot, which, dt * 1e3, n / dt, ok), flush=True)
        if which != "own":
            ctx.set_stream(None)
        ctx.close()
