"""Mencius-shaped safety check of the oracle (CPU): single-slot proposals, noop ranges, thrifty / partial
delivery and leader changes with Phase-1 recovery (Acceptor.handlePhase1a -> Leader safe values -> re-proposal)
interleaved at random; a slot that was chosen once -- as a command or as a Noop of a range -- must never be
chosen with another value.  Exercises K4 (mencius/Acceptor.scala:237-291, mencius/ProxyLeader.scala:255-411),
f2 (the recovery scan) and the fused step together, the way a Mencius leader group uses them."""
import numpy as np
import pytest

L, A, R, F = 3, 2, 3, 1          # leader groups, acceptor groups per leader group, acceptors per group, f
S = 3 * 64
NOOP = -1


def _bits(idx):
    w = np.zeros(4, np.uint64)
    for i in idx:
        w[i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    return w


def _run(oracle, seed):
    rng = np.random.default_rng(seed)
    be = oracle.System(oracle.make_config(num_slots=S, num_replicas=R, num_groups=A, num_leader_groups=L, f=F,
                                          num_leaders=2, tally_ways=8))
    chosen = {}
    rounds = [0] * L
    # what the current leader of each leader group is bound to: its own proposals of this round and the safe
    # values of its Phase 1 -- a leader proposes at most one value per slot and round
    known = [dict() for _ in range(L)]
    for lg in range(L):                                   # every leader group's first leader runs Phase 1
        for ag in range(A):
            assert be.acceptor_phase1a(lg * A + ag, 0)[0] == 0
    next_value = [1000]

    def learn(slot, value):
        assert chosen.setdefault(slot, value) == value, (seed, slot, chosen[slot], value)

    def owned(lg, lo, hi):
        first = lo + ((lg - lo) % L)
        return np.arange(first, hi, L, dtype=np.int32)

    for step in range(160):
        lg = int(rng.integers(0, L))
        kind = rng.integers(0, 10)
        if kind < 5:                                      # the leader proposes commands in some of its slots
            mine = owned(lg, 0, S)
            slot = np.sort(rng.choice(mine, size=int(rng.integers(1, 12)), replace=False)).astype(np.int32)
            slot = np.array([s for s in slot if int(s) not in known[lg]], np.int32)
            if len(slot) == 0:
                continue
            val = np.arange(next_value[0], next_value[0] + len(slot), dtype=np.int32)
            next_value[0] += len(slot)
            known[lg].update({int(s_): int(v_) for s_, v_ in zip(slot, val)})
            tgt = np.stack([_bits(rng.choice(R, size=int(rng.integers(1, R + 1)), replace=False)) for _ in slot])
            st, ch, cr, cv, nr = be.phase2_fused(slot, np.full(len(slot), rounds[lg], np.int32), val, tgt)
            assert st in (0,), st
            for s, c, v in zip(slot, ch, cv):
                if c:
                    learn(int(s), int(v))
        elif kind < 8:                                    # the leader skips a stretch of its slots
            lo = lg + L * int(rng.integers(0, S // L - 8))
            hi = min(S, lo + L * int(rng.integers(1, 8)))
            if any(int(s) in known[lg] for s in owned(lg, lo, hi)):
                continue                                  # it only skips slots it has not used
            st, new = be.proxy_open_noop_range(lo, hi, rounds[lg])
            assert st == 0 and new
            known[lg].update({int(s_): NOOP for s_ in owned(lg, lo, hi)})
            tgts = np.stack([_bits(rng.choice(R, size=int(rng.integers(1, R + 1)), replace=False)) for _ in range(A)])
            st, vb, nb, nr = be.acceptor_phase2a_noop_range(lo, hi, rounds[lg], tgts)
            assert st == 0
            st, done = be.proxy_phase2b_noop_range(lo, hi, rounds[lg], vb)
            assert st == 0
            if done:
                for s in owned(lg, lo, hi):
                    learn(int(s), NOOP)
        else:                                             # leader change in this leader group
            new_round = rounds[lg] + 1
            quorum = np.zeros((L * A, 4), np.uint64)
            ok = True
            for ag in range(A):
                q = rng.choice(R, size=F + 1, replace=False)
                st, pb, nb = be.acceptor_phase1a(lg * A + ag, new_round, 0, _bits(q))
                ok = ok and st == 0
                quorum[lg * A + ag] = _bits(q)
            assert ok
            rounds[lg] = new_round
            st, mx, sr, sv = be.leader_phase1b_scan(0, quorum, S)
            assert st == 0
            slots = np.array([s for s in range(min(S, mx + 1)) if s % L == lg and sr[s] >= 0], np.int32)
            # the new leader is bound only by what its Phase 1 saw; every other slot is free again
            known[lg] = {int(s_): int(sv[s_]) for s_ in slots}
            if len(slots):
                vals = sv[slots].astype(np.int32)
                st, ch, cr, cv, nr = be.phase2_fused(slots, np.full(len(slots), new_round, np.int32), vals)
                assert st == 0
                for s, c, v in zip(slots, ch, cv):
                    if c:
                        learn(int(s), int(v))
    return chosen


def test_mencius_mixed_streams_never_change_a_chosen_slot(oracle):
    total = noops = 0
    for seed in range(30):
        chosen = _run(oracle, seed)
        total += len(chosen)
        noops += sum(1 for v in chosen.values() if v == NOOP)
    assert total > 1500 and noops > 200


def test_mencius_safety_harness_negative_control(oracle):
    """the check is not vacuous: a new leader that ignores its Phase 1 and proposes a fresh value in a slot
    that holds a chosen Noop does get a second value chosen there"""
    tripped = 0
    for seed in range(30):
        rng = np.random.default_rng(seed)
        be = oracle.System(oracle.make_config(num_slots=S, num_replicas=R, num_groups=A, num_leader_groups=L, f=F,
                                              num_leaders=2, tally_ways=8))
        for g in range(L * A):
            be.acceptor_phase1a(g, 0)
        lo, hi = 1, 1 + L * 10
        assert be.proxy_open_noop_range(lo, hi, 0) == (0, 1)
        st, vb, nb, nr = be.acceptor_phase2a_noop_range(lo, hi, 0)
        assert be.proxy_phase2b_noop_range(lo, hi, 0, vb) == (0, 1)          # Noop chosen in slots 1, 4, ..., 28
        for ag in range(A):
            be.acceptor_phase1a(1 * A + ag, 1)
        slot = np.array([1 + L * int(rng.integers(0, 10))], np.int32)
        st, ch, cr, cv, nr = be.phase2_fused(slot, np.array([1], np.int32), np.array([777], np.int32))
        if ch[0] and cv[0] != NOOP:
            tripped += 1                                                       # a second value in a chosen slot
    assert tripped == 30
