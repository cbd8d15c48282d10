"""The replica-axis path of one GPU, without the exchange (measurement aid for DESIGN.md section 7): K2-open +
K1 on this GPU's acceptor range + K2 tally of a full 256-bit bitmap, 2^20 fresh slots per step, per-slot ballot
model.  R_local = 256 (one GPU holds the whole group), 128, 64, 32 (the shard of 2, 4, 8 GPUs).  The all-reduce of
the 32 MiB of bitmaps between K1 and K2 is not part of this script (one GPU)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import frankenpaxos_amd as fa

B, WINDOWS, STEPS = 1 << 20, 6, 5
dev = torch.device("cuda:0")
for R_local in (256, 128, 64, 32):
    ctx = fa.Context(fa.make_config(num_slots=B * WINDOWS, num_replicas=R_local, f=127, ballot_mode=fa.FPX_BALLOT_PER_SLOT,
                                    replica_base=0, replicas_total=256, flags=fa.FPX_F_TRUSTED))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.acceptor_phase1a(0, 0)
    rnd = torch.zeros(B, dtype=torch.int32, device=dev)
    vb = torch.empty((B, 4), dtype=torch.int64, device=dev)
    full = torch.full((B, 4), -1, dtype=torch.int64, device=dev)   # what the all-reduce would deliver
    ch = torch.empty(B, dtype=torch.uint8, device=dev)
    cr = torch.empty(B, dtype=torch.int32, device=dev)
    cv = torch.empty(B, dtype=torch.int32, device=dev)
    slots = [torch.arange(w * B, (w + 1) * B, dtype=torch.int32, device=dev) for w in range(WINDOWS)]
    val = slots[0] ^ 0x5A5A5A

    def step(w):
        ctx.proxy_open_dev(slots[w], rnd, val, None)
        ctx.acceptor_phase2a_dev(slots[w], rnd, val, None, vb, None, None)
        ctx.proxy_phase2b_dev(slots[w], rnd, full if R_local < 256 else vb, ch, cr, cv)

    step(0)
    assert ctx.sync() == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for w in range(1, 1 + STEPS):
        step(w)
    assert ctx.sync() == 0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    assert int(ch.sum().item()) == B
    print("R_local = %3d of 256: open + K1 + K2 = %.3f ms per 2^20 slots  (%.3e slots/s per GPU before the exchange)"
          % (R_local, dt * 1e3, B / dt))
    del ctx
