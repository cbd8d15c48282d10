"""bench.py --config {2,3,4,5}: the other BASELINE.json configs as bench lines (same JSON schema as the headline:
roofline with that workload's algorithmic bytes, cpu_baseline from the oracle).  The headline metric's grid
(2^20 slots x 256 acceptors) is bench.py's default and lives there.

  2  MultiPaxos f=1, 64k slots x 3 acceptors               fused K3, one step = 65 536 fresh slots
  3  Compartmentalized MultiPaxos, 16 groups of 2x2 grids   fused K3, one step = 2^20 fresh slots (slot % 16 -> group)
  4  EPaxos, 5 replicas                                     K5, one step = one tick of 2^20 fresh commands, 1024 keys
  thrifty  MultiPaxos, R = 255, f = 127                     fused K3, every Phase2a to a rotating window of f + 1 acceptors (the
                                                            reference's default delivery, ProxyLeader.scala:190-191)
  acceptor_model  the headline grid (2^20 x 256, threshold 128) under FPX_BALLOT_ACCEPTOR: one promised round per acceptor, the
                  reference's actual acceptor (multipaxos/Acceptor.scala:95), 2064 B per slot, write-bound
  thrifty_random  as `thrifty`, but every Phase2a goes to a RANDOM f + 1 of the 255 acceptors: what the reference literally
                  draws (Random.shuffle(group).take(f + 1), ProxyLeader.scala:190-191)
  host_path       the headline step through HOST pointers (fpx_phase2_fused_submit / _wait, page-locked arrays, 3 calls in
                  flight): H2D proposals + fused step + D2H chosen records per 2^20 x 256 call
  adversarial     SURVEY.md 8(d)'s parity / adversarial stream, seed 1, at full size (64 epochs, leader changes, Nacks,
                  re-proposals, random target subsets), device-resident, proposals/s; every output == the oracle afterwards
  5  Mencius, 256 leader groups x 3 acceptors, 4M slots     one step = a band of 2^22 slots: the leader groups that have
                                                            commands propose them (fused K3), the others skip their
                                                            slots with one noop range each (fused K4); N > 1: leader
                                                            groups are sharded over the ranks (strong scaling)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0


def _splitmix_values(slot):
    from bench import steady_values_torch
    return steady_values_torch(slot)


class RegionTimer:
    """ONE pair of HIP events around the timed steps, on the stream the library launches on (the torch current stream,
    handed to the context): device time per step = elapsed / steps, the gaps between a step's kernels included.  (A pair
    of events per step put two marker packets between the steps: 8 % of a config-4 tick, 15 % of a config-5 band,
    profiles/r04_events.md.)"""

    def __init__(self):
        self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def start(self):
        self.a.record()

    def stop(self):
        self.b.record()

    def total_ms(self):
        torch.cuda.synchronize()
        return self.a.elapsed_time(self.b)


# ------------------------------------------------------------------------------------------------------------------
# configs 2, 3: the fused MultiPaxos step on other shapes
# ------------------------------------------------------------------------------------------------------------------
def multipaxos_setup(fa, dev, local_rank, ballot_mode, cfg, K, Wm):
    shapes = {
        "2": dict(slots=65536, R=3, groups=1, kw=dict(f=1, quorum_kind=fa.FPX_Q_THRESHOLD),
                  name="MultiPaxos f=1: fused Phase-2 step, 65 536 fresh slots x 3 acceptors per step"),
        "3": dict(slots=1 << 20, R=4, groups=16, kw=dict(quorum_kind=fa.FPX_Q_GRID, grid_rows=2, grid_cols=2),
                  name="Compartmentalized MultiPaxos: 16 acceptor groups of 2x2 grids (slot % 16 -> group), fused "
                       "Phase-2 step, 2^20 fresh slots per step"),
        "acceptor_model": dict(slots=1 << 20, R=256, groups=1, kw=dict(f=127, quorum_kind=fa.FPX_Q_THRESHOLD),
                               name="MultiPaxos Phase-2 fused step on the headline grid (2^20 fresh slots x 256 acceptors per step, "
                                    "threshold 128) with ONE promised round per acceptor (FPX_BALLOT_ACCEPTOR): the reference's "
                                    "actual acceptor state, multipaxos/Acceptor.scala:95"),
    }[cfg]
    if cfg == "acceptor_model":
        ballot_mode = fa.FPX_BALLOT_ACCEPTOR
    n, R, G = shapes["slots"], shapes["R"], shapes["groups"]
    windows = K + Wm
    ctx = fa.Context(fa.make_config(num_slots=windows * n, num_replicas=R, num_groups=G, ballot_mode=ballot_mode,
                                    tally_ways=4, device=local_rank, flags=fa.FPX_F_TRUSTED, **shapes["kw"]))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for g in range(G):
        assert ctx.acceptor_phase1a(g, 0)[0] == 0
    ctx.flush_promises()
    steps = []
    for w in range(windows):
        slot = torch.arange(w * n, (w + 1) * n, dtype=torch.int32, device=dev)
        steps.append((slot, torch.zeros_like(slot), _splitmix_values(slot), torch.zeros(n, dtype=torch.uint8, device=dev),
                      torch.full((n,), -7, dtype=torch.int32, device=dev), torch.full((n,), -7, dtype=torch.int32, device=dev)))

    def step(i):
        slot, rnd, val, ch, cr, cv = steps[i]
        ctx.phase2_fused_dev(slot, rnd, val, None, ch, cr, cv)

    def verify(lo, hi):
        done = 0
        for i in range(lo, hi):
            slot, rnd, val, ch, cr, cv = steps[i]
            assert bool(ch.all()) and bool((cv == val).all()) and bool((cr == 0).all()), "step %d" % i
            done += int(ch.sum().item())
        return done

    cells = (3 if ballot_mode == 1 else 2) * 4 * R        # ballot read (per_slot) + voteRound + voteValue written
    bps = cells + 12 + 9 + 20                             # + proposal, chosen record, tally key row (16 read + 4 written)
    if cfg == "acceptor_model":
        bps = 8 + 8 * R + 8                               # SURVEY.md 8(d)'s faithful-scalar model: 2064 B per slot

    def cpu(fa_cfg_mode=ballot_mode):
        from oracle import pyoracle
        from tests import workloads as W
        pyoracle.build()
        S = min(n, 1 << 18)
        ref = pyoracle.System(pyoracle.make_config(num_slots=S, num_replicas=R, num_groups=G, ballot_mode=ballot_mode,
                                                   **shapes["kw"]))
        for g in range(G):
            ref.acceptor_phase1a(g, 0)
        slot, rnd, val = W.steady_stream(S)
        t0 = time.perf_counter()
        out = ref.phase2_fused(slot, rnd, val)
        dt = time.perf_counter() - t0
        assert out[0] == 0 and int(out[1].sum()) == S
        return {"value": S / dt, "unit": "slots/s", "cores": 1, "kind": "port",
                "sample": "oracle/fpx_oracle.c fpo_phase2_fused, %d slots of the same workload, 1 thread" % S}

    # (steps of ~10 - 40 us are timed by ONE pair of events around the timed region, like configs 4 and 5, and want a few
    # hundred of them: over 20 steps the region is a quarter of a millisecond and measures the GPU waking up after the fence
    # -- 0.018 - 0.032 ms per 65 536 x 3 step over 20 steps, 0.0121 (round per acceptor) / 0.0129 (ballot per cell) over 200;
    # profiles/r05_small_steps.md.  A captured HIP graph of the steps is no faster: the step is its two dependent launches.)
    return dict(ctx=ctx, step=step, verify=verify, units=n, unit="slots/s", bytes_per_unit=bps, region_timed=(cfg in ("2", "3")),
                workload=shapes["name"], kernel="k_phase2 (fused K3)", profile=lambda: ctx.profile_read(),
                metric=("committed log slots/sec (BASELINE.json configs[%d])" % (int(cfg) - 1)) if cfg.isdigit() else
                       "committed log slots/sec at 1M slots x 256 replicas, one promised round per acceptor (FPX_BALLOT_ACCEPTOR)", cpu=cpu,
                extra={"slots_per_step": n, "replicas": R, "acceptor_groups": G,
                       "ballot_model": "per_slot" if ballot_mode == 1 else "acceptor"})


# ------------------------------------------------------------------------------------------------------------------
# thrifty: the reference's DEFAULT delivery -- every Phase2a goes to f + 1 of the group's 2f + 1 acceptors
# (multipaxos/ProxyLeader.scala:190-191).  R = 255 = 2f + 1 (the reference's legal group size), f = 127; the f + 1 are a
# window of neighbouring acceptors that rotates from slot to slot in steps of 16 (what jni/Native.scala's
# GpuProxyLeader sends; any f + 1 will do) -- k_phase2's packed walk, two rows per wavefront step.
# ------------------------------------------------------------------------------------------------------------------
def thrifty_setup(fa, dev, local_rank, ballot_mode, K, Wm, random_targets=False):
    n, R, F = 1 << 20, 255, 127
    windows = K + Wm
    ctx = fa.Context(fa.make_config(num_slots=windows * n, num_replicas=R, f=F, ballot_mode=ballot_mode, tally_ways=4,
                                    device=local_rank,
                                    flags=fa.FPX_F_TRUSTED | (fa.FPX_F_SCATTERED_TARGETS if random_targets else 0)))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    assert ctx.acceptor_phase1a(0, 0)[0] == 0
    ctx.flush_promises()
    s = torch.arange(n, device=dev)[:, None]
    j = torch.arange(256, device=dev)[None, :]
    start = 16 * (s % ((R - (F + 1)) // 16 + 1))
    sh = torch.arange(64, device=dev, dtype=torch.int64)

    def pack(b):  # bool [k, 256] -> int64 [k, 4]
        w = b.view(-1, 4, 64).to(torch.int64)
        return ((w[..., :63] << sh[:63]).sum(-1) | (w[..., 63] << 63)).contiguous()

    if random_targets:
        # Random.shuffle(group).take(f + 1): a uniformly random f + 1 of the R acceptors per slot (seeded: the reference's
        # own draw is the unseeded global RNG); acceptor 254's column is read back by verify()
        gen = torch.Generator(device=dev)
        gen.manual_seed(0xF9A405)
        parts = []
        for c0 in range(0, n, 1 << 16):
            r = torch.rand((min(1 << 16, n - c0), R), device=dev, generator=gen)
            kth = r.kthvalue(F + 1, dim=1, keepdim=True).values
            b = torch.zeros((r.shape[0], 256), dtype=torch.bool, device=dev)
            b[:, :R] = r <= kth
            parts.append(pack(b))
        tgt = torch.cat(parts).contiguous()
        in_last = ((tgt[:, 3] >> 62) & 1).to(torch.bool).cpu().numpy()   # is acceptor 254 a target of slot s (every window)
        del parts, r, b
    else:
        tgt = pack((j >= start) & (j < start + F + 1))
    steps = []
    for w in range(windows):
        slot = torch.arange(w * n, (w + 1) * n, dtype=torch.int32, device=dev)
        steps.append((slot, torch.zeros_like(slot), _splitmix_values(slot), torch.zeros(n, dtype=torch.uint8, device=dev),
                      torch.full((n,), -7, dtype=torch.int32, device=dev), torch.full((n,), -7, dtype=torch.int32, device=dev)))

    def step(i):
        slot, rnd, val, ch, cr, cv = steps[i]
        ctx.phase2_fused_dev(slot, rnd, val, tgt, ch, cr, cv)

    def verify(lo, hi):
        done = 0
        for i in range(lo, hi):
            slot, rnd, val, ch, cr, cv = steps[i]
            assert bool(ch.all()) and bool((cv == val).all()) and bool((cr == 0).all()), "step %d" % i
            done += int(ch.sum().item())
        if random_targets:   # acceptor 254 voted (round 0) exactly in the slots whose mask names it
            col = np.asarray(ctx.read_acceptor(0, 254)[2])[lo * n:hi * n].reshape(-1, n)
            assert bool(((col == 0) == in_last[None, :]).all()) and bool(((col == -1) == ~in_last[None, :]).all())
            return done
        # the votes are where the targets were, nowhere else (one acceptor inside and one outside every window)
        vr, vv = ctx.read_acceptor(0, 127)[2], ctx.read_acceptor(0, 0)[2]
        assert bool((np.asarray(vr)[lo * n:hi * n] == 0).all())              # acceptor 127 is in every window
        first = np.asarray(vv)[lo * n:hi * n].reshape(-1, 8)                 # acceptor 0: only in the windows that start at 0
        assert bool((first[:, 0] == 0).all()) and bool((first[:, 1:] == -1).all())
        return done

    # per slot: 128 voters x (voteRound 4 + voteValue 4) written, 32 B target mask + 12 B proposal read, 9 B chosen
    # record written (VERDICT r03's 1077 B model); PER_SLOT adds the 128 ballots read (4 B each)
    bps = 128 * 8 + 32 + 12 + 9 + (128 * 4 if ballot_mode == 1 else 0)

    def cpu():
        from oracle import pyoracle
        from tests import workloads as W
        pyoracle.build()
        S = 1 << 16
        ref = pyoracle.System(pyoracle.make_config(num_slots=S, num_replicas=R, f=F, ballot_mode=ballot_mode))
        ref.acceptor_phase1a(0, 0)
        slot, rnd, val = W.steady_stream(S)
        t = W.bits_from_bool((W.fast_subsets if random_targets else W.run_subsets)(np.random.default_rng(1), S, R, F + 1, F + 1))
        t0 = time.perf_counter()
        out = ref.phase2_fused(slot, rnd, val, t)
        dt = time.perf_counter() - t0
        assert out[0] == 0
        return {"value": S / dt, "unit": "slots/s", "cores": 1, "kind": "port",
                "sample": "oracle/fpx_oracle.c fpo_phase2_fused with f + 1 target windows, %d slots, 1 thread" % S}

    if random_targets:
        return dict(ctx=ctx, step=step, verify=verify, units=n, unit="slots/s", bytes_per_unit=bps,
                    workload="MultiPaxos thrifty delivery as the reference literally draws it: fused Phase-2 step, 2^20 fresh slots "
                             "per step, each Phase2a to a uniformly RANDOM f + 1 = 128 of the 255 acceptors "
                             "(Random.shuffle(group).take(f + 1), ProxyLeader.scala:190-191)",
                    kernel="k_phase2<64, 2, *, fused> (FPX_F_SCATTERED_TARGETS: fresh rows are written as whole cells blended with "
                           "-1; a random half of a row touches every 64-byte sector, so the traffic is the dense step's)",
                    profile=lambda: ctx.profile_read(),
                    metric="committed log slots/sec, thrifty delivery to a random f + 1 (ProxyLeader.scala:190-191)", cpu=cpu,
                    extra={"slots_per_step": n, "replicas": R, "f": F, "targets_per_slot": F + 1,
                           "ballot_model": "per_slot" if ballot_mode == 1 else "acceptor",
                           "byte_model": "the 1077 (+512) B of `thrifty`: what the VOTERS' cells need; the kernel moves whole rows "
                                         "(~2 x), see roofline.traffic"})
    return dict(ctx=ctx, step=step, verify=verify, units=n, unit="slots/s", bytes_per_unit=bps,
                workload="MultiPaxos thrifty delivery (the reference's default): fused Phase-2 step, 2^20 fresh slots per step, "
                         "each Phase2a to f + 1 = 128 neighbouring acceptors of 255 (a window rotating in steps of 16)",
                kernel="k_phase2<64, 3, *, fused> (packed walk: two rows per wavefront step) behind it the row-at-a-time walk "
                       "for chunks that are no runs (none here)", profile=lambda: ctx.profile_read(),
                metric="committed log slots/sec, thrifty delivery (ProxyLeader.scala:190-191)", cpu=cpu,
                extra={"slots_per_step": n, "replicas": R, "f": F, "targets_per_slot": F + 1,
                       "ballot_model": "per_slot" if ballot_mode == 1 else "acceptor"})


# ------------------------------------------------------------------------------------------------------------------
# config 4: EPaxos pre-accept ticks
# ------------------------------------------------------------------------------------------------------------------
def epaxos_setup(fa, dev, local_rank, K, Wm):
    from frankenpaxos_amd.epaxos import EPaxos
    from tests import workloads as W
    from tests.workloads import random_tick

    n, num_keys, m = 5, 1024, 1 << 20
    epx = EPaxos(n, num_keys, device=local_rank)
    epx.set_stream(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(4)
    nxt = [0] * n
    ticks = []
    for t in range(K + Wm):
        leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 64.0)
        key = (W.splitmix64_at(np.arange(t * m, (t + 1) * m, dtype=np.uint64)) % np.uint64(num_keys)).astype(np.int32)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        # one packed line per command (fpx_epx_preaccept_packed_dev): deps | leader_deps | own_values_end | fast
        ticks.append((d(leader), d(number), d(key), d(is_set), d(mask), d(rank),
                      torch.full((m, epx.packed_stride()), -7, dtype=torch.int32, device=dev)))
    def step(i):
        leader, number, key, is_set, mask, rank, packed = ticks[i]
        epx.preaccept_packed_dev(leader, number, key, is_set, mask, rank, packed)

    def prewarm():
        """The timed region is a few milliseconds: behind two or three warm-up ticks the clocks are still rising (0.115 ms per
        tick by events behind 3 ticks, 0.108 behind 40: profiles/r06_k5.md).  Drawing 40 more ticks on the host would cost
        20 s, so the warm-up ticks are run several times over: a tick's effect on the conflict indexes is a max per
        (replica, key, leader) -- idempotent --, its outputs land in its own buffer, and the oracle check of the first
        timed tick below still replays every tick exactly once."""
        for _ in range(12):
            for i in range(Wm):
                step(i)

    def verify(lo, hi):
        """the first timed tick against the oracle on EVERY output (the oracle replays the ticks before it: the conflict
        indexes carry over), the others by their path counts"""
        from oracle import pyoracle
        assert epx.sync() == 0
        pyoracle.build()
        ref = pyoracle.EPaxos(n, num_keys)
        h = lambda t: t.cpu().numpy()
        for i in range(lo + 1):
            want = ref.preaccept(*[h(x) for x in ticks[i][:6]])
            assert want[0] == 0
        fast, deps, ldeps, own = (h(x) for x in epx.unpack(ticks[lo][6]))
        assert (fast == want[1]).all() and (deps == want[2]).all() and (ldeps == want[3]).all() and (own == want[4]).all(), \
            "tick %d differs from the oracle" % lo
        done = 0
        for i in range(lo, hi):
            fast, _, _, own = epx.unpack(ticks[i][6])
            nf = int(fast.sum().item())
            assert 0 < nf < m and int(fast.max().item()) == 1, "tick %d: %d fast-path commits" % (i, nf)
            assert bool((own == 0).all())               # FIFO channels: no own-column holes
            done += m                                   # every command is decided (fast commit or Accept phase)
        return done

    def cpu():
        from oracle import pyoracle
        pyoracle.build()
        mm = 1 << 18
        ref = pyoracle.EPaxos(n, num_keys)
        args = random_tick(np.random.default_rng(5), n, num_keys, mm, [0] * n, 64.0)
        t0 = time.perf_counter()
        out = ref.preaccept(*args)
        dt = time.perf_counter() - t0
        assert out[0] == 0
        return {"value": mm / dt, "unit": "commands/s", "cores": 1, "kind": "port",
                "sample": "oracle/fpx_oracle_epaxos.c fpo_epx_preaccept, one tick of 2^18 commands, 1 thread"}

    # per command, what has to cross HBM: inputs (leader, number, key 12 B, is_set + mask 2 B, rank 5 x 4 B) read once, one
    # 64-byte output line written once (the conflict rows never leave the chip) = 98 B.  The 32-byte record the
    # partition pass writes and the key kernel reads back is the design's own traffic, not part of the model.
    bpc = 34 + 64
    return dict(ctx=epx, step=step, verify=verify, prewarm=prewarm, units=m, unit="commands/s", bytes_per_unit=bpc,
                workload="EPaxos n = 5: one tick = 2^20 fresh single-key commands (1024 keys, Bernoulli get/set) through "
                         "the pre-accept phase of all replicas: conflict scan in every replica's delivery order, "
                         "fast-path test, slow-path union, commit into every conflict index",
                kernel="K5 tick, second form (k_kp_hist, k_kp_scatter<5>: one record per command into its key's segment; "
                       "k_epx_key2<5>: per key on chip -- order by rank per replica, scans, decisions, index update)", region_timed=True,
                metric="EPaxos commands decided/sec (BASELINE.json configs[3])", cpu=cpu,
                extra={"commands_per_tick": m, "replicas": n, "keys": num_keys,
                       "byte_model": "34 B inputs + 64 B packed output line per command; round 2's model (243 B) also counted "
                                     "the 4 conflict rows of 20 B each way, which no longer cross HBM"})


# ------------------------------------------------------------------------------------------------------------------
# 4_execute: what a configs[3] tick commits, through dependency-graph execution on the device
# ------------------------------------------------------------------------------------------------------------------
def epaxos_execute_setup(fa, dev, local_rank, K, Wm, fifo=True):
    from frankenpaxos_amd.epaxos import EPaxos
    from tests import workloads as W
    from tests.workloads import random_tick

    n, num_keys, m = 5, 1024, 1 << 20
    epx = EPaxos(n, num_keys, device=local_rank)
    epx.set_stream(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(45)
    nxt = [0] * n
    leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 64.0, fifo=fifo)
    key = (W.splitmix64_at(np.arange(m, dtype=np.uint64)) % np.uint64(num_keys)).astype(np.int32)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dl, dn = d(leader), d(number)
    packed = torch.zeros((m, epx.packed_stride()), dtype=torch.int32, device=dev)
    epx.preaccept_packed_dev(dl, dn, d(key), d(is_set), d(mask), d(rank), packed)     # the tick (K5): every command is decided
    assert epx.sync() == 0
    order = torch.full((m,), -1, dtype=torch.int32, device=dev)
    comp = torch.full((m,), -1, dtype=torch.int32, device=dev)
    first, count = np.zeros(n, np.int32), np.asarray(nxt, np.int32)
    results = []

    def step(i):
        results.append(epx.execute_dev(dl, dn, packed, first, count, order, comp))

    def verify(lo, hi):
        from tests.test_epaxos import check_execution_order
        for ne, nc, nh in results:
            assert ne == m and nh == 0 and 0 < nc <= m, (ne, nc, nh)
        fast, deps, ldeps, own = (x.cpu().numpy() for x in epx.unpack(packed))
        o = order.cpu().numpy()
        # a valid execution order of the whole tick: every instance once, no component before one it depends on
        check_execution_order(n, leader, number, deps, own[:, 0], leader[o], number[o], np.bincount(comp.cpu().numpy()))
        return (hi - lo) * m

    def cpu():
        from frankenpaxos_amd import depgraph as P
        mm = 1 << 17
        fast, deps, ldeps, own = (x.cpu().numpy() for x in epx.unpack(packed))
        sel = np.zeros(m, bool)
        for L in range(n):                                   # a dense prefix of every column: 2^17 instances in all
            idx = np.nonzero(leader == L)[0]
            sel[idx[number[idx] < mm // n]] = True
        g = P.DependencyGraph(n, kind=P.FPX_DG_ZIGZAG)
        t0 = time.perf_counter()
        g.commit_epx(leader[sel], number[sel], np.minimum(deps[sel], mm // n), own[sel] * 0)
        out = g.execute_arrays()
        dt = time.perf_counter() - t0
        return {"value": int(sel.sum()) / dt, "unit": "commands/s", "cores": 1, "kind": "port",
                "sample": "csrc/fpx_depgraph.cpp (the library's HOST graph, a C++ restatement of ZigzagTarjanDependencyGraph.scala held to "
                          "the reference's test vectors), the first %d instances of the tick, 1 thread" % int(sel.sum())}

    # per command, what must cross HBM: the instance (8 B) and its packed line (64 B) in, its place and component out (8 B)
    return dict(ctx=epx, step=step, verify=verify, units=m, unit="commands/s", bytes_per_unit=80,
                workload="EPaxos n = 5: what one tick of 2^20 single-key commands (1024 keys) commits -- every command with its agreed "
                         "dependencies -- through dependency-graph execution ON THE DEVICE (fpx_epx_execute_dev: strongly connected "
                         "components in reverse topological order, depgraph/TarjanDependencyGraph.scala:225-276); one step = the whole "
                         "tick executed (the same tick every step), the call ends with the counts on the host",
                kernel="closure rounds on 16-byte rows (k_dp_relax + k_dp_carry), k_dp_keys, two LSD radix sorts, k_dp_count_starts, k_dp_emit "
                       "(csrc/fpx_depgraph_pk.hpp)", region_timed=True,
                metric="EPaxos commands executed/sec (dependency-graph execution of a BASELINE.json configs[3] tick's commits)", cpu=cpu,
                extra={"commands_per_tick": m, "replicas": n, "keys": num_keys, "channels": "fifo" if fifo else "reordering",
                       "byte_model": "8 B instance + 64 B packed line in, 8 B (position, component) out per command; the closure rounds' "
                                     "traffic (128 B per vertex and round, 4 - 7 rounds) is the algorithm's own"})


# ------------------------------------------------------------------------------------------------------------------
# config 5: Mencius bands -- commands from half of the leader groups, noop ranges from the other half
# ------------------------------------------------------------------------------------------------------------------
def mencius_setup(fa, dev, local_rank, rank, world, K, Wm):
    L_total, R, band_total = 256, 3, 1 << 22
    if L_total % world:
        raise SystemExit("--config 5 shards 256 leader groups: --gpus must divide 256")
    L = L_total // world                 # this rank's leader groups (a Mencius deployment of its own: slot % L)
    band = band_total // world           # its slots per step
    rows = band // L
    windows = K + Wm
    ctx = fa.Context(fa.make_config(num_slots=windows * band, num_replicas=R, num_groups=1, num_leader_groups=L, f=1,
                                    tally_ways=4, device=local_rank, flags=fa.FPX_F_TRUSTED))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for lg in range(L):
        assert ctx.acceptor_phase1a(lg, 0)[0] == 0
    steps = []
    lgs = torch.arange(L, device=dev)
    for w in range(windows):
        active = (lgs + w) % 2 == 0                       # the leader groups with commands alternate
        base = w * band
        r = torch.arange(rows, device=dev, dtype=torch.int64)
        if os.environ.get("FPX_CFG5_ORDER") == "slot":      # one batch in slot order over all proposing leader groups
            slot = (base + r[:, None] * L + lgs[active][None, :]).reshape(-1).to(torch.int32)
        else:                                               # the proposing leader groups' batches back to back, each in slot order
            slot = (base + r[None, :] * L + lgs[active][:, None]).reshape(-1).to(torch.int32)
        idle = lgs[~active]
        start = (base + idle).to(torch.int32)
        end = (base + (rows - 1) * L + idle + 1).to(torch.int32)
        steps.append((slot, torch.zeros_like(slot), _splitmix_values(slot),
                      torch.zeros(slot.numel(), dtype=torch.uint8, device=dev),
                      torch.full((slot.numel(),), -7, dtype=torch.int32, device=dev),
                      start.contiguous(), end.contiguous(), torch.zeros_like(start),
                      torch.zeros(start.numel(), dtype=torch.uint8, device=dev)))
    def step(i):
        slot, rnd, val, ch, cv, start, end, rr, rch = steps[i]
        # ONE call per step (fpx_mencius_band_fused_dev).  The leader groups with commands and those with ranges alternate,
        # never both in a step, and the call says so (`independent`): the step is then two launches -- the vote kernel with
        # the range chain as its first workgroup, the ranges' fill with the vote kernel's fold of maxima in its grid -- instead
        # of four (profiles/r05_cfg5.md).  FPX_CFG5_SERIAL=1: the two halves one after the other, as rounds 2 - 4 ran them
        ctx.mencius_band_fused_dev(slot, rnd, val, None, ch, None, cv, None, start, end, rr, None, None, None, None, None, rch,
                                   independent=os.environ.get("FPX_CFG5_SERIAL") != "1")

    def verify(lo, hi):
        done = 0
        for i in range(lo, hi):
            slot, rnd, val, ch, cv, start, end, rr, rch = steps[i]
            assert bool(ch.all()) and bool((cv == val).all()), "step %d: commands" % i
            assert bool(rch.all()), "step %d: noop ranges" % i
            done += int(ch.sum().item()) + int(rch.sum().item()) * rows     # a chosen range commits its `rows` slots
        return done

    def cpu():
        from oracle import pyoracle
        from tests import workloads as W
        pyoracle.build()
        S = 1 << 19
        ref = pyoracle.System(pyoracle.make_config(num_slots=S, num_replicas=R, num_groups=1, num_leader_groups=256, f=1))
        for lg in range(256):
            ref.acceptor_phase1a(lg, 0)
        lg = np.arange(256)
        rws = S // 256
        slot = (np.arange(rws)[:, None] * 256 + lg[lg % 2 == 0][None, :]).reshape(-1).astype(np.int32)
        idle = lg[lg % 2 == 1]
        t0 = time.perf_counter()
        a = ref.phase2_fused(slot, np.zeros(len(slot), np.int32), W.steady_values(slot))
        b = ref.noop_ranges_fused(idle.astype(np.int32), ((rws - 1) * 256 + idle + 1).astype(np.int32), np.zeros(128, np.int32))
        dt = time.perf_counter() - t0
        assert a[0] == 0 and b[0] == 0 and a[1].all() and b[5].all()
        return {"value": S / dt, "unit": "slots/s", "cores": 1, "kind": "port",
                "sample": "oracle/fpx_oracle.c: one band of 2^19 slots, 256 leader groups (128 propose commands through "
                          "fpo_phase2_fused, 128 skip through fpo_noop_ranges_fused), 1 thread"}

    # per slot: voteRound + voteValue of 3 acceptors written (24 B); command slots add proposal 12 + chosen 5 + tally key
    # row 20 B (half of the slots): 24 + 18.5
    bps = 24 + 0.5 * (12 + 5 + 20)
    return dict(ctx=ctx, step=step, verify=verify, units=band, unit="slots/s", bytes_per_unit=bps,
                workload="Mencius: %d leader groups x 3 acceptors on this GPU (256 in the job), one step = a band of "
                         "%d slots: half of the leader groups propose commands in their slots (fused K3; %s), the "
                         "others skip theirs with one noop range each (fused K4)"
                         % (L, band, "one batch in slot order across the leader groups" if os.environ.get("FPX_CFG5_ORDER") == "slot"
                            else "the proposing leader groups' batches back to back, each in slot order"),
                kernel="k_phase2_band (fused K3 + the range chain) + k_ranges_fill_lg_fin (K4 fill + the fold of the vote kernel's maxima), one call per step (fpx_mencius_band_fused_dev)", region_timed=True,
                metric="committed log slots/sec (BASELINE.json configs[4])", cpu=cpu,
                extra={"slots_per_step_per_gpu": band, "leader_groups_per_gpu": L, "replicas": R,
                       "ranges_per_step_per_gpu": L // 2},
                scaling="strong")


# ------------------------------------------------------------------------------------------------------------------
# host_path: SURVEY.md 8(d) (ii) -- the headline step end to end through the C ABI's HOST-pointer entry point
# ------------------------------------------------------------------------------------------------------------------
def host_path_line(args, fa, dev, local_rank):
    """fpx_phase2_fused_submit / _wait on page-locked arrays (fpx_host_alloc), up to 3 calls in flight: per call 12 B per
    slot of proposals go up (copy engine), the fused 2^20 x 256 step runs and writes its 9 B per slot of Chosen records
    straight into the caller's arrays (profiles/r06_host_path.md).  PCIe-inclusive, never bench.py's `value`."""
    import ctypes as C
    from tests import workloads as W
    K, Wm = args.steps, args.warmup
    B, R, F = 1 << 20, 256, 127
    ballot_mode = fa.FPX_BALLOT_PER_SLOT if args.ballot == "per_slot" else fa.FPX_BALLOT_ACCEPTOR
    ctx = fa.Context(fa.make_config(num_slots=B * (K + Wm), num_replicas=R, f=F, ballot_mode=ballot_mode, tally_ways=4,
                                    device=local_rank))
    assert ctx.acceptor_phase1a(0, 0)[0] == 0
    ctx.flush_promises()
    L = fa.lib()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    keep, batches = [], []
    for k in range(K + Wm):
        objs = [fa.PinnedArray((B,), dt) for dt in (np.int32, np.int32, np.int32, np.uint8, np.int32, np.int32)]
        keep.append(objs)
        sl, rd, vl, och, ocr, ocv = [x.array for x in objs]
        sl[:] = np.arange(k * B, (k + 1) * B, dtype=np.int32)
        rd[:] = 0
        vl[:] = W.steady_values(sl)
        ocr[:] = -7
        ocv[:] = -7
        batches.append((sl, rd, vl, och, ocr, ocv))
    tick = C.c_int32()

    def pump(lo, hi):
        inflight = []
        for sl, rd, vl, och, ocr, ocv in batches[lo:hi]:
            if len(inflight) == 3:
                assert L.fpx_phase2_fused_wait(ctx._h, inflight.pop(0)) == 0
            assert L.fpx_phase2_fused_submit(ctx._h, B, p(sl), p(rd), p(vl), None, p(och), p(ocr), p(ocv), None, C.byref(tick)) == 0
            inflight.append(tick.value)
        while inflight:
            assert L.fpx_phase2_fused_wait(ctx._h, inflight.pop(0)) == 0

    pump(0, Wm)
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    pump(Wm, Wm + K)
    elapsed = time.perf_counter() - t0
    launches, kernel_ms = ctx.profile_read()
    assert ctx.sync() == 0
    done = 0
    for sl, rd, vl, och, ocr, ocv in batches[Wm:]:
        assert int(och.sum()) == B and bool((ocv == vl).all()) and bool((ocr == 0).all())
        done += B
    ctx.close()
    bps = 3088 if ballot_mode == 1 else 2064
    per = elapsed / K
    return {
        "metric": "committed log slots/sec END TO END through host pointers (SURVEY.md 8d (ii); PCIe-inclusive, never `value`)",
        "value": done / elapsed, "unit": "slots/s", "n_gpus": 1, "steps": K, "warmup": Wm, "ms_per_step": per * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "the headline step through fpx_phase2_fused_submit / _wait: 2^20 fresh slots x 256 acceptors per call, "
                               "proposals (12 B per slot) and Chosen records (9 B per slot) in page-locked HOST arrays, 3 calls in flight",
                   "baseline_config": "host_path", "ballot_model": args.ballot, "slots_per_step": B, "replicas": R,
                   "verified": "every timed call checked after the timed region: every slot chosen in round 0 with its proposed value",
                   "pcie_bytes_per_slot": 21, "pcie_GBs": 21 * B / per / 1e9},
        "roofline": {"bound": "hbm", "kernel": "k_phase2<64, 0, *, fused>: inputs staged in HBM by the copy engine, records written "
                                               "straight into the caller's page-locked arrays; k_validate, k_finalize, k_status_snap around it",
                     "achieved": bps * B / per / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bps * B / per / 1e9 / HBM_PEAK_GBS,
                     "traffic": traffic_of("host_path", args.ballot), "traffic_round": traffic_entry(traffic_key("host_path", args.ballot))[1],
                     "algorithmic_bytes_per_unit": bps, "units_per_launch": B,
                     "avg_kernel_ms": kernel_ms / max(launches, 1), "launches_timed": launches,
                     "kernel_time_source": "`achieved` = algorithmic HBM bytes / WALL time per call (the PCIe transfers included); "
                                           "avg_kernel_ms = the vote kernel alone (fpx_profile_*)",
                     "note": "NOT bound by PCIe (21 B per slot each call = pcie_GBs of a ~55 GB/s link): the call is the fused step "
                             "(+3 % for its posted writes to host memory) + forced validation + finalize + status snapshot and the "
                             "dependent-launch gaps between them; profiles/r06_host_path.md"},
    }


# ------------------------------------------------------------------------------------------------------------------
# adversarial: SURVEY.md 8(d)'s parity stream at full size, timed; every output against the oracle afterwards
# ------------------------------------------------------------------------------------------------------------------
def adversarial_line(args, fa, dev, local_rank, seed=1):
    from oracle import pyoracle
    from tests import workloads as W
    S, R, Q = 1 << 20, 256, 128
    ballot_mode = fa.FPX_BALLOT_PER_SLOT if args.ballot == "per_slot" else fa.FPX_BALLOT_ACCEPTOR
    script = W.adversarial_script(S, R, Q, seed, epochs=64, fused=True, subsets=W.fast_subsets)
    kw = dict(num_slots=S, num_replicas=R, f=Q - 1, ballot_mode=ballot_mode, tally_ways=8)
    # the stream satisfies the run contract by construction (an epoch carries one round): no validation kernels
    ctx = fa.Context(fa.make_config(device=local_rank, flags=fa.FPX_F_SCATTERED_TARGETS | fa.FPX_F_TRUSTED, **kw))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d = lambda a, view=None: torch.from_numpy(np.ascontiguousarray(a) if view is None else np.ascontiguousarray(a).view(view)).to(dev)
    ops, proposals = [], 0
    for op in script:
        if op[0] == "phase1a":
            _, g, rnd, wm, tgt = op
            ops.append(("phase1a", g, rnd, wm, None if tgt is None else d(tgt, np.int64)))
        else:
            _, slot, rr, val, tgt = op
            n = len(slot)
            proposals += n
            ops.append(("fused", d(slot), d(rr), d(val), d(tgt, np.int64), torch.zeros(n, dtype=torch.uint8, device=dev),
                        torch.full((n,), -7, dtype=torch.int32, device=dev), torch.full((n,), -7, dtype=torch.int32, device=dev),
                        torch.full((n,), -7, dtype=torch.int32, device=dev)))

    def run():
        for op in ops:
            if op[0] == "phase1a":
                ctx.acceptor_phase1a_dev(op[1], op[2], op[3], op[4])
            else:
                ctx.phase2_fused_dev(*op[1:])

    reps = max(1, args.steps // 4)          # a "step" here is the WHOLE stream (64 epochs); a few repetitions, the last one verified
    run()                                   # warm-up: scratch buffers reach their size
    assert ctx.sync() == 0
    times, t_enq = [], []
    for _ in range(reps):
        ctx.reset()
        assert ctx.sync() == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        t_enq.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        assert ctx.sync() == 0
    dt = sorted(times)[len(times) // 2]
    enqueue_ms = sorted(t_enq)[len(t_enq) // 2] * 1e3   # host time to enqueue one pass (92 calls): the stream must not be bound by it
    # AFTER the timed region: the oracle replays the script message by message; every output of every epoch must agree
    pyoracle.build()
    ref = pyoracle.System(pyoracle.make_config(**kw))
    t_or = time.perf_counter()
    want = W.run_script(ref, script)
    t_or = time.perf_counter() - t_or
    chosen = nacked = 0
    fused_ops = [op for op in ops if op[0] == "fused"]
    k = 0
    for w_out in want:
        if w_out[0] != "fused":
            continue
        _, st, ch, cr, cv, nr = w_out
        op = fused_ops[k]
        k += 1
        assert st == 0
        g_ch, g_cr, g_cv, g_nr = (op[j].cpu().numpy() for j in (5, 6, 7, 8))
        np.testing.assert_array_equal(g_ch, ch)
        m = ch.astype(bool)
        np.testing.assert_array_equal(g_cr[m], cr[m])
        np.testing.assert_array_equal(g_cv[m], cv[m])
        np.testing.assert_array_equal(g_nr, nr)
        chosen += int(ch.sum())
        nacked += int((nr >= 0).sum())
    np.testing.assert_array_equal(ctx.state_digest(), ref.state_digest())
    ctx.close()
    # per proposal: the dense model's 3088 (2064) B + the 32-byte target mask; random target subsets make most rows a
    # read-modify-write of partially voted cells, which the model does not count
    bps = (3088 if ballot_mode == 1 else 2064) + 32
    return {
        "metric": "proposals/sec on SURVEY.md 8(d)'s adversarial stream (seed %d) at 1M slots x 256 replicas" % seed,
        "value": proposals / dt, "unit": "proposals/s", "n_gpus": 1, "steps": reps, "warmup": 1, "ms_per_step": dt * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "the parity / adversarial stream of SURVEY.md 8(d), seed %d, 2^20 slots x 256 acceptors: 64 epochs "
                               "(one fused launch each), leader changes with 25 %% of the acceptors pre-promised (stale Phase2a's "
                               "Nacked), 5 %% re-proposals, target masks = random subsets of U[q - 8, R] acceptors; device-resident, "
                               "one step = the whole stream" % seed,
                   "baseline_config": "adversarial", "ballot_model": args.ballot, "proposals": proposals,
                   "fused_launches": len(fused_ops), "phase1a_calls": len(ops) - len(fused_ops), "chosen": chosen, "nacked": nacked,
                   "host_enqueue_ms_per_pass": enqueue_ms,
                   "verified": "AFTER the timed region: chosen flag / round / value and Nack round of every proposal of every epoch "
                               "and the whole-state digest == the CPU oracle replaying the script message by message (%.1f s)" % t_or},
        "roofline": {"bound": "hbm", "kernel": "k_phase2_fin<64, 2, *, fused> x 64 (each with the fold of the launch before) + k_p1a_fast x 28",
                     "achieved": bps * proposals / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": bps * proposals / dt / 1e9 / HBM_PEAK_GBS, "traffic": traffic_of("adversarial", args.ballot),
                     "traffic_round": traffic_entry(traffic_key("adversarial", args.ballot))[1],
                     "algorithmic_bytes_per_unit": bps, "units_per_launch": proposals / max(1, len(fused_ops)),
                     "avg_kernel_ms": dt * 1e3, "launches_timed": reps,
                     "kernel_time_source": "wall clock around the whole stream (perf_counter between synchronisations), median of the repetitions",
                     "note": "about 21 500 proposals per launch: each launch is a read-modify-write of partially voted rows (~3.9 KB of "
                             "traffic per proposal) + 4.8 us of launch floor; a Phase1a is one 5 us launch (profiles/r06_phase1a.md)"},
    }


# ------------------------------------------------------------------------------------------------------------------
# config 1: SURVEY.md 8(d)'s first rung -- the oracle alone, on the host
# ------------------------------------------------------------------------------------------------------------------
def config1_entry(reps=21):
    """BASELINE.json configs[0] as SURVEY.md 8(d) states it: MultiPaxos f = 1 -- one acceptor group of 3, quorum 2, 2 replicas --
    1000 commands through the CPU oracle behind its strict FIFO message pump (the stand-in for the reference's in-process
    Transport: every Phase2a / Phase2b is one queued message, handled one at a time), both replicas' logs executing the
    Chosen records; microseconds per slot on this host.  No GPU involved: this is the CPU reference case of the ladder."""
    from oracle import pyoracle
    from tests import workloads as W
    pyoracle.build()
    n = 1000
    slot, rnd, val = W.steady_stream(n)
    times = []
    for _ in range(reps):
        ref = pyoracle.System(pyoracle.make_config(num_slots=n, num_replicas=3, f=1))
        ref.acceptor_phase1a(0, 0)
        logs = [pyoracle.Log(), pyoracle.Log()]
        t0 = time.perf_counter()
        st, ch, cr, cv, nr = ref.phase2_fifo_pump(slot, rnd, val)
        for lg in logs:
            for s_, v_ in zip(slot.tolist(), cv.tolist()):
                lg.chosen(s_, v_)
        times.append(time.perf_counter() - t0)
        assert st == 0 and bool(ch.all()) and bool((cv == val).all()) and bool((cr == 0).all())
        assert all(lg.executed_watermark == n for lg in logs)
    dt = sorted(times)[len(times) // 2]
    return {"value": float("%.5g" % (n / dt)), "unit": "slots/s", "us_per_slot": float("%.4g" % (dt / n * 1e6)), "steps": reps,
            "commands": n, "host_only": True, "cores": 1, "kind": "port", "verified": True}


def traffic_entry(key):
    """(HBM bytes per step, the round they were measured in, where) from the PMC passes committed under profiles/ --
    profiles/traffic.json, every key tagged with its round -- or (None, None, None) if that workload was never profiled"""
    import json
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key)
        return (e["bytes"], e["round"], e.get("source")) if e else (None, None, None)
    except Exception:
        return (None, None, None)


def traffic_key(config, ballot="per_slot"):
    return ("config%s" % config) if str(config)[:1].isdigit() else ("%s_%s" % (config, ballot))


def traffic_of(config, ballot="per_slot"):
    return traffic_entry(traffic_key(config, ballot))[0]


def run(args, fa, dist, dev, rank, world, local_rank, all_reduce):
    K, Wm = args.steps, args.warmup
    ballot_mode = fa.FPX_BALLOT_PER_SLOT if args.ballot == "per_slot" else fa.FPX_BALLOT_ACCEPTOR
    if args.config == "host_path":
        return host_path_line(args, fa, dev, local_rank) if rank == 0 else None
    if args.config == "adversarial":
        return adversarial_line(args, fa, dev, local_rank) if rank == 0 else None
    if args.config in ("2", "3", "acceptor_model"):
        w = multipaxos_setup(fa, dev, local_rank, ballot_mode, args.config, K, Wm)
    elif args.config == "4":
        w = epaxos_setup(fa, dev, local_rank, K, Wm)
    elif args.config == "4_execute":
        w = epaxos_execute_setup(fa, dev, local_rank, K, Wm)
    elif args.config in ("thrifty", "thrifty_random"):
        w = thrifty_setup(fa, dev, local_rank, ballot_mode, K, Wm, random_targets=args.config == "thrifty_random")
    else:
        w = mencius_setup(fa, dev, local_rank, rank, world, K, Wm)
    ctx = w["ctx"]

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if "prewarm" in w:
        w["prewarm"]()
    for i in range(Wm):
        w["step"](i)
    assert ctx.sync() == 0
    # kernel time: a step of one vote kernel is timed by the library (fpx_profile_*: events on the kernel's own dispatch
    # packet); a step of several kernels (configs 4, 5) by one pair of events around the timed steps
    region = RegionTimer() if w.get("region_timed") else None
    if hasattr(ctx, "profile_enable"):
        ctx.profile_enable(region is None)
    fence()
    t0 = time.perf_counter()
    if region:
        region.start()
    for i in range(Wm, Wm + K):
        w["step"](i)
    if region:
        region.stop()
    fence()
    elapsed = time.perf_counter() - t0
    launches, kernel_ms = (K, region.total_ms()) if region else w["profile"]()
    assert ctx.sync() == 0
    done = w["verify"](Wm, Wm + K)
    assert done == K * w["units"], (done, K * w["units"])
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_reduce(t, dist.ReduceOp.MAX)
        elapsed = float(t.item())
        t = torch.tensor([done], dtype=torch.int64, device=dev)
        all_reduce(t, dist.ReduceOp.SUM)
        done = int(t.item())
    if rank != 0:
        return None
    avg_kernel_s = (kernel_ms / max(launches, 1)) * 1e-3
    achieved = w["bytes_per_unit"] * w["units"] / avg_kernel_s / 1e9 if launches else None
    line = {
        "metric": w["metric"], "value": done / elapsed, "unit": w["unit"], "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": w.get("scaling", "weak"),
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": dict({"workload": w["workload"], "baseline_config": int(args.config) if args.config.isdigit() else args.config,
                        "verified": "every timed step checked after the timed region" +
                                    (": first timed tick == the CPU oracle on every output, all ticks by path counts"
                                     if args.config == "4" else ": every instance executed once, in an order in which no component "
                                     "precedes one it depends on" if args.config == "4_execute" else ": every slot chosen with its proposed value")},
                       **w["extra"]),
        "roofline": {
            "bound": "hbm", "kernel": w["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic_of(args.config, args.ballot),
            "traffic_round": traffic_entry(traffic_key(args.config, args.ballot))[1],
            "traffic_source": "profiles/traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this command, "
                              "all kernels of one step summed; every key carries the round it was measured in), not measured in this run",
            "algorithmic_bytes_per_unit": w["bytes_per_unit"], "units_per_launch": w["units"],
            "avg_kernel_ms": kernel_ms / max(launches, 1), "launches_timed": launches,
            "kernel_time_source": ("one pair of HIP events around the timed steps on the launch stream / steps: all kernels of "
                                   "a step and the gaps between them" if region else
                                   "HIP events on the vote kernel's own dispatch packet (hipExtLaunchKernelGGL; fpx_profile_*)"),
            "note": "small-row workloads (16-byte rows) run at 3.6 - 4.0 TB/s of actual traffic at best and are bound by "
                    "dependent-step latency below ~10^6 slots per launch: the fraction of the HBM peak is reported for "
                    "the contract, the absolute rate is the figure of merit",
        },
    }
    if args.config == "2":
        line["roofline"]["bound_in_practice"] = "launch latency: a 65 536-slot x 3 step is one ~6 us vote kernel and a ~4 us " \
                                                "k_finalize with a dependent-launch gap behind each (profiles/r03_small_n.txt; a " \
                                                "captured HIP graph of the steps runs no faster: r05_small_step_graph.py); " \
                                                "the config is BASELINE.json's bring-up / bit-exactness case, not a bandwidth case"
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = w["cpu"]()
    if hasattr(ctx, "close"):
        ctx.close()
    return line
