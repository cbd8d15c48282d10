"""How does the fused step's time depend on the BYTES a row moves when the rows stay the same in number?  R = 256,
2^20 slots per step, every message targets a run of L acceptors (start = 16 * (slot % (256 / 16)), cyclic), f = L - 1
so that the run is a quorum.  If the time does not follow L the walk is bound per ROW, not per byte."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import frankenpaxos_amd as fa

S, R = 1 << 20, 256
dev = torch.device("cuda:0")
ballot = int(sys.argv[1]) if len(sys.argv) > 1 else 0
windows = 5

def pack(bits):
    w = bits.view(-1, 4, 64).to(torch.int64)
    sh = torch.arange(64, device=dev, dtype=torch.int64)
    lo = (w[..., :63] << sh[:63]).sum(-1)
    return lo | (w[..., 63] << 63)

for L in (256, 128, 64, 32, 16):
    ctx = fa.Context(fa.make_config(num_slots=windows * S, num_replicas=R, f=L - 1, ballot_mode=ballot, flags=fa.FPX_F_TRUSTED))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.acceptor_phase1a(0, 0)
    ctx.flush_promises()
    s = torch.arange(S, device=dev)[:, None]
    j = torch.arange(256, device=dev)[None, :]
    tgt = pack(((j - 16 * (s % 16)) % 256) < L)
    steps = []
    for w in range(windows):
        slot = torch.arange(w * S, (w + 1) * S, dtype=torch.int32, device=dev)
        steps.append((slot, torch.zeros(S, dtype=torch.int32, device=dev), slot * 7))
    ch = torch.zeros(S, dtype=torch.uint8, device=dev)
    ctx.phase2_fused_dev(*steps[0], tgt, ch, None, None)
    assert ctx.sync() == 0 and bool(ch.all())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for w in range(1, windows):
        ctx.phase2_fused_dev(*steps[w], tgt, ch, None, None)
    assert ctx.sync() == 0
    dt = (time.perf_counter() - t0) / (windows - 1)
    print("ballot model %d  run of %3d acceptors  %.3f ms/step  %.3e slots/s  %.2f TB/s of cells" % (ballot, L, dt * 1e3, S / dt, S * L * (8 + 4 * ballot) / dt / 1e12), flush=True)
    ctx.close()
