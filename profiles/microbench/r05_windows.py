"""Per-window kernel times of the headline step: one context of W windows of 2^20 slots x 256 acceptors,
three laps over the window ring (lap 0 fresh rows, laps 1-2 re-proposals in a higher round), the vote kernel's own
dispatch-packet events launch by launch (fpx_profile_read_launches).  Run twice: FPX_PLACEMENT_CHUNKS=0 (one allocation,
rounds 2-4) and default (chunks paired by measurement).  FPX_DEBUG=1 prints the placement probes on stderr."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import frankenpaxos_amd as fa
from bench import steady_values_torch, SLOTS_PER_STEP as N

W = int(sys.argv[1]) if len(sys.argv) > 1 else 25
ballot = fa.FPX_BALLOT_PER_SLOT if (len(sys.argv) <= 2 or sys.argv[2] == "per_slot") else fa.FPX_BALLOT_ACCEPTOR
dev = torch.device("cuda:0")
ctx = fa.Context(fa.make_config(num_slots=W * N, num_replicas=256, f=127, ballot_mode=ballot, tally_ways=4, flags=fa.FPX_F_TRUSTED))
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
assert ctx.acceptor_phase1a(0, 0)[0] == 0
ctx.flush_promises()
print("placement:", ctx.placement_stats(), flush=True)
ch = torch.zeros(N, dtype=torch.uint8, device=dev)
cr = torch.zeros(N, dtype=torch.int32, device=dev)
cv = torch.zeros(N, dtype=torch.int32, device=dev)
ins = []
for w in range(W):
    slot = torch.arange(w * N, (w + 1) * N, dtype=torch.int32, device=dev)
    ins.append((slot, steady_values_torch(slot)))
ctx.profile_enable(True)
for lap in range(3):
    rnd = torch.full((N,), lap, dtype=torch.int32, device=dev)
    for w in range(W):
        ctx.phase2_fused_dev(ins[w][0], rnd, ins[w][1], None, ch, cr, cv)
    ms = ctx.profile_read_launches()
    assert ctx.sync() == 0 and bool(ch.all())
    a = np.array(ms)
    print("lap %d per window [ms]: %s" % (lap, " ".join("%.4f" % x for x in a)))
    print("lap %d: min %.4f median %.4f max %.4f mean %.4f sigma %.4f (%.2f %%)" % (lap, a.min(), np.median(a), a.max(), a.mean(), a.std(), 100 * a.std() / a.mean()), flush=True)
ctx.close()
