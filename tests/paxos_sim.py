"""A small randomized MultiPaxos simulation in the style of the reference's SimulatedSystem tests
(shared/src/test/scala/frankenpaxos/simulator/{Simulator,SimulatedSystem}.scala and
shared/src/test/scala/frankenpaxos/multipaxos/MultiPaxos.scala): all actors share one in-memory
message pool (FakeTransport), the harness delivers / drops / duplicates messages in random order and
starts leader changes at random, and after every step the safety invariant is checked.

The acceptors and the proxy leader are the system under test, driven ONE MESSAGE AT A TIME through
the batch API (n = 1, single-acceptor target masks), so the same schedule can be replayed on the CPU
oracle and on the GPU library and their traces compared.  The two leaders (Phase 1, safe-value
selection: multipaxos/Leader.scala:306-329, 504-577) and the replicas' logs live in this harness.

Invariant (multipaxos/MultiPaxos.scala:291-305 "logs are prefix compatible"): no two Chosen messages
for one slot ever carry different values.
"""
import random

import numpy as np

from tests import workloads as W


def one_bit(j):
    v = np.zeros((1, 4), np.uint64)
    v[0, j >> 6] = np.uint64(1) << np.uint64(j & 63)
    return v


class Leader:
    def __init__(self, index):
        self.index = index
        self.round = -1
        self.phase = "idle"
        self.phase1bs = {}


def simulate(be, seed, R=3, f=1, S=6, steps=600, num_leaders=2, drop=0.05, dup=0.05):
    """returns (trace, chosen) ; raises AssertionError on a safety violation"""
    rng = random.Random(seed)
    pool = []
    leaders = [Leader(i) for i in range(num_leaders)]
    chosen = {}
    trace = []
    max_round = [-1]
    i32 = lambda x: np.array([x], np.int32)

    def start_round(ld):
        ld.round = W.next_classic_round(num_leaders, ld.index, max(max_round[0], ld.round))
        max_round[0] = max(max_round[0], ld.round)
        ld.phase = "p1"
        ld.phase1bs = {}
        for r in range(R):
            pool.append(("p1a", ld.index, ld.round, r))

    def deliver(m):
        kind = m[0]
        if kind == "p1a":
            _, li, rnd, r = m
            st, pb, nb = be.acceptor_phase1a(0, rnd, 0, one_bit(r)[0])
            ok = bool(pb.any())
            trace.append(("p1a", r, rnd, ok))
            if ok:
                _, _, vr, vv, _ = be.read_acceptor(0, r)
                pool.append(("p1b", li, rnd, r, vr.copy(), vv.copy()))
            else:
                pr, _ = be.read_scalars()
                pool.append(("nack", li, int(pr[0, r])))
        elif kind == "p1b":
            _, li, rnd, r, vr, vv = m
            ld = leaders[li]
            if ld.round != rnd or ld.phase != "p1":
                return
            ld.phase1bs[r] = (vr, vv)
            if len(ld.phase1bs) < R - f:  # a read quorum of the f+1 write-quorum system
                return
            ld.phase = "p2"
            for slot in range(S):
                best_round, best_val = -1, None
                for (avr, avv) in ld.phase1bs.values():
                    if avr[slot] > best_round:
                        best_round, best_val = int(avr[slot]), int(avv[slot])
                val = best_val if best_round >= 0 else li * 1000000 + rnd * 1000 + slot
                pool.append(("p2a_pl", slot, rnd, val))
        elif kind == "p2a_pl":
            _, slot, rnd, val = m
            st, new = be.proxy_open(i32(slot), i32(rnd), i32(val))
            trace.append(("open", slot, rnd, val, int(new[0])))
            if new[0]:
                for r in range(R):
                    pool.append(("p2a", slot, rnd, val, r))
        elif kind == "p2a":
            _, slot, rnd, val, r = m
            st, vb, nb, nr = be.acceptor_phase2a(i32(slot), i32(rnd), i32(val), one_bit(r))
            voted = bool(vb.any())
            trace.append(("p2a", r, slot, rnd, val, voted, int(nr[0])))
            if voted:
                pool.append(("p2b", r, slot, rnd))
            else:
                # Acceptor.scala:197: the Nack goes to leaders(roundSystem.leader(phase2a.round))
                pool.append(("nack", rnd % num_leaders, int(nr[0])))
        elif kind == "p2b":
            _, r, slot, rnd = m
            st, ch, cr, cv = be.proxy_phase2b(i32(slot), i32(rnd), one_bit(r))
            assert st == 0
            trace.append(("p2b", r, slot, rnd, int(ch[0]), int(cv[0])))
            if ch[0]:
                chosen.setdefault(slot, set()).add(int(cv[0]))
                assert len(chosen[slot]) == 1, "SAFETY VIOLATION in slot %d: %r" % (slot, chosen[slot])
        elif kind == "nack":
            _, li, rnd = m
            max_round[0] = max(max_round[0], rnd)
            ld = leaders[li]
            if rnd > ld.round:
                ld.phase = "idle"

    start_round(leaders[0])
    for _ in range(steps):
        x = rng.random()
        if x < 0.04 or not pool:
            start_round(leaders[rng.randrange(num_leaders)])
            continue
        k = rng.randrange(len(pool))
        m = pool[k]
        y = rng.random()
        if y < drop:
            pool.pop(k)
            continue
        if y >= drop + dup:
            pool.pop(k)  # otherwise: delivered AND left in the pool (a duplicate)
        deliver(m)
    return trace, chosen
