"""Kernel-level timings of the rows around the fused step (measurement aid; run under rocprofv3 --kernel-trace --stats):
f1 replica log ingest + executed watermark on 2^20 Chosen per call, f2 Phase-1 recovery scan over 2^20 slots x 256
acceptors, K4 Mencius noop range over 2^20 slots (3 leader groups x 2 acceptor groups x 3 acceptors)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import frankenpaxos_amd as fa

dev = torch.device("cuda:0")
S = 1 << 20


def timed(label, fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print("%-58s %.3f ms" % (label, (time.perf_counter() - t0) / reps * 1e3))


# ---- f1 + f2 on the headline grid
ctx = fa.Context(fa.make_config(num_slots=8 * S, num_replicas=256, f=127, flags=fa.FPX_F_TRUSTED))
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.acceptor_phase1a(0, 0)
rnd = torch.zeros(S, dtype=torch.int32, device=dev)
ch = torch.empty(S, dtype=torch.uint8, device=dev)
cr = torch.empty(S, dtype=torch.int32, device=dev)
cv = torch.empty(S, dtype=torch.int32, device=dev)
w = [0]


def fused_then_log():
    slot = torch.arange(w[0] * S, (w[0] + 1) * S, dtype=torch.int32, device=dev)
    w[0] += 1
    ctx.phase2_fused_dev(slot, rnd, slot, None, ch, cr, cv, None)
    ctx.replica_chosen_dev(slot, cv, ch)   # the chosen records feed the replica log on the device


timed("K3 + f1 (2^20 Chosen -> log, executed watermark), per step", fused_then_log, reps=6)
assert ctx.sync() == 0
wm, nc = ctx.replica_state()
assert wm == nc == 7 * S, (wm, nc)
q = np.zeros((1, 4), np.uint64)
q[0, :2] = np.uint64(0xFFFFFFFFFFFFFFFF)  # Phase1b's of acceptors 0..127
timed("f2 safe values of 2^20 slots x 128 of 256 Phase1b's (host call)", lambda: ctx.leader_phase1b_scan(0, q, S), reps=3)
del ctx

# ---- K4: a Mencius leader group skips 2^20 of its slots
L, A = 3, 2
ctx = fa.Context(fa.make_config(num_slots=3 * S + 16, num_replicas=3, num_groups=A, num_leader_groups=L, f=1, tally_ways=8))
r = [0]


def noop_range():
    r[0] += 1
    st, vb, nb, nr = ctx.acceptor_phase2a_noop_range(1, 1 + 3 * S, r[0])
    assert st == 0 and nr == -1


timed("K4 noop range over 2^20 slots of one leader group (host call)", noop_range, reps=3)
