"""Thrifty delivery (the reference's default: ProxyLeader.handlePhase2a sends each Phase2a to a random
f+1 of the 2f+1 acceptors of the group, ProxyLeader.scala:190-191) through the fused step: R = 255
acceptors (f = 127), 2^20 slots per step, per-slot target bitmaps resident in HBM.  Measurement aid:
prints slots/s for (a) random f+1 subsets, (b) a rotating contiguous f+1 run, (c) dense delivery."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import frankenpaxos_amd as fa

S, R, F = 1 << 20, 255, 127
dev = torch.device("cuda:0")
ballot = int(sys.argv[1]) if len(sys.argv) > 1 else 0
scattered = len(sys.argv) > 2 and sys.argv[2] == "scattered"  # FPX_F_SCATTERED_TARGETS hint
only = sys.argv[3] if len(sys.argv) > 3 else "all"      # random | rotating | dense | all
phase = sys.argv[4] if len(sys.argv) > 4 else "both"    # first | both   (PMC runs: one kind of launch per process)
windows = 6


def pack(bits):  # bool [n, 256] -> int64 [n, 4]
    w = bits.view(-1, 4, 64).to(torch.int64)
    sh = torch.arange(64, device=dev, dtype=torch.int64)
    lo = (w[..., :63] << sh[:63]).sum(-1)
    return lo | (w[..., 63] << 63)


def random_masks(n):
    out = []
    for c in range(0, n, 1 << 16):
        m = min(1 << 16, n - c)
        r = torch.rand(m, R, device=dev)
        kth = r.kthvalue(F + 1, dim=1, keepdim=True).values
        bits = torch.zeros(m, 256, dtype=torch.bool, device=dev)
        bits[:, :R] = r <= kth
        out.append(pack(bits))
    return torch.cat(out)


def rotating_masks(n):
    s = torch.arange(n, device=dev)[:, None]
    j = torch.arange(256, device=dev)[None, :]
    bits = (((j - s) % R) < (F + 1)) & (j < R)
    return pack(bits)


def aligned_masks(n):
    """rotating sector-aligned windows of f + 1 acceptors (start a multiple of 16, no wrap over the end of the row):
    what GpuProxyLeader sends by default -- k_phase2's packed walk"""
    s = torch.arange(n, device=dev)[:, None]
    j = torch.arange(256, device=dev)[None, :]
    start = 16 * (s % ((R - (F + 1)) // 16 + 1))
    bits = (j >= start) & (j < start + F + 1)
    return pack(bits)


for name, maker in (("random f+1 subsets", random_masks), ("rotating f+1 run", rotating_masks), ("aligned f+1 run", aligned_masks),
                    ("dense", None)):
    if only != "all" and not name.startswith(only):
        continue
    ctx = fa.Context(fa.make_config(num_slots=windows * S, num_replicas=R, f=F, ballot_mode=ballot,
                                    flags=fa.FPX_F_TRUSTED | (fa.FPX_F_SCATTERED_TARGETS if scattered else 0)))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.acceptor_phase1a(0, 0)
    ctx.flush_promises()
    print('ctx ok', name, flush=True)
    tgt = maker(S) if maker else None
    torch.cuda.synchronize(); print('masks ok', flush=True)
    steps = []
    for w in range(windows):
        slot = torch.arange(w * S, (w + 1) * S, dtype=torch.int32, device=dev)
        steps.append((slot, torch.zeros(S, dtype=torch.int32, device=dev), slot * 7))
    ch = torch.zeros(S, dtype=torch.uint8, device=dev)
    ctx.phase2_fused_dev(*steps[0], tgt, ch, None, None)
    st0 = ctx.sync(); print('first step', st0, int(ch.sum()), flush=True)
    assert st0 == 0 and bool(ch.all())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for w in range(1, windows):
        ctx.phase2_fused_dev(*steps[w], tgt, ch, None, None)
    assert ctx.sync() == 0
    dt = (time.perf_counter() - t0) / (windows - 1)
    print("ballot model %d  %-20s first proposal  %.3f ms/step  %.3e slots/s" % (ballot, name, dt * 1e3, S / dt))
    if phase == "first":
        ctx.close()
        continue
    # the same windows re-proposed in round 1 to a DIFFERENT f+1: the rows now hold votes, partial cells take
    # the 4-byte (or, with the scattered hint, load / blend / store) path
    tgt2 = torch.roll(tgt, 12345, 0) if tgt is not None else None
    one = torch.ones(S, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for w in range(1, windows):
        ctx.phase2_fused_dev(steps[w][0], one, steps[w][2], tgt2, ch, None, None)
    assert ctx.sync() == 0
    dt = (time.perf_counter() - t0) / (windows - 1)
    print("ballot model %d  %-20s re-proposal     %.3f ms/step  %.3e slots/s" % (ballot, name, dt * 1e3, S / dt))
    ctx.close()
