"""Fused K3 at the reference's everyday group sizes (f = 1, 2, 3: R = 3, 5, 7 acceptors), dense delivery,
2^22 fresh slots per step, inputs resident in HBM (measurement aid).  At small R a slot row is one 16- or
32-byte cell, so the proxy leader's tally table (16-byte key row + value) weighs as much as the votes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import frankenpaxos_amd as fa

B, WINDOWS, STEPS = 1 << 22, 6, 5
dev = torch.device("cuda:0")
for ballot in (0, 1):
    for R in [int(x) for x in os.environ.get("SMALL_R", "3,5,7,16,64").split(",")]:
        f = (R - 1) // 2 if R % 2 else R // 2 - 1
        ctx = fa.Context(fa.make_config(num_slots=B * WINDOWS, num_replicas=R, f=f, ballot_mode=ballot,
                                        flags=fa.FPX_F_TRUSTED))
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.acceptor_phase1a(0, 0)
        rnd = torch.zeros(B, dtype=torch.int32, device=dev)
        ch = torch.empty(B, dtype=torch.uint8, device=dev)
        cr = torch.empty(B, dtype=torch.int32, device=dev)
        cv = torch.empty(B, dtype=torch.int32, device=dev)
        slots = [torch.arange(w * B, (w + 1) * B, dtype=torch.int32, device=dev) for w in range(WINDOWS)]
        val = slots[0] ^ 0x5A5A5A
        ctx.phase2_fused_dev(slots[0], rnd, val, None, ch, cr, cv, None)
        assert ctx.sync() == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for w in range(1, 1 + STEPS):
            ctx.phase2_fused_dev(slots[w], rnd, val, None, ch, cr, cv, None)
        assert ctx.sync() == 0
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / STEPS
        assert int(ch.sum().item()) == B
        RS = (R + 3) // 4 * 4
        alg = (3 if ballot else 2) * RS * 4 + 12 + 9      # cells (padded row) + proposal + chosen record
        print("ballot model %d  R = %3d (f = %2d)  %.3f ms per 2^22 slots  %.3e slots/s  (%d B/slot of cells+records = %.2f TB/s)"
              % (ballot, R, f, dt * 1e3, B / dt, alg, alg * B / dt / 1e12))
        del ctx
