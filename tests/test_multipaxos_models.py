"""oracle/fpx_oracle.c (flat arrays, 256-bit vote sets, (slot, round)-keyed tally ways) against
oracle/multipaxos_maps.py (Acceptor objects with a states map, a proxy leader with Map[(slot, round), Pending],
quorum systems as sets): random adversarial streams -- competing Phase1a's, stale rounds Nacked, equal rounds voting
again, re-proposals, duplicate Phase2a's and Phase2b's, votes after Done, Phase2b's nobody asked for -- under every
quorum kind.  Every reply, then every acceptor's round / maxVotedSlot / votes.  CPU only."""
import numpy as np
import pytest

from oracle import multipaxos_maps as model

COVERAGE = []
Q_THRESHOLD, Q_MAJORITY, Q_GRID, Q_UNANIMOUS = 0, 1, 2, 3


def bits_to_indices(words):
    return [j for j in range(256) if (int(words[j >> 6]) >> (j & 63)) & 1]


CASES = [
    # R, groups, f, kind, rows, cols
    (5, 1, 2, Q_THRESHOLD, 0, 0),
    (3, 4, 1, Q_THRESHOLD, 0, 0),          # non-flexible MultiPaxos: slot % 4 picks the acceptor group
    (5, 1, 0, Q_MAJORITY, 0, 0),
    (6, 1, 0, Q_GRID, 2, 3),
    (4, 1, 0, Q_UNANIMOUS, 0, 0),
    (4, 3, 0, Q_GRID, 2, 2),               # BASELINE config #3's shape: independent 2x2 grids
    (7, 2, 3, Q_THRESHOLD, 0, 0),
]


@pytest.mark.parametrize("R,G,f,kind,rows,cols", CASES)
def test_flat_oracle_and_map_model_agree(oracle, R, G, f, kind, rows, cols):
    S = 96
    ref = oracle.System(oracle.make_config(num_slots=S, num_replicas=R, num_groups=G, f=f, quorum_kind=kind,
                                           grid_rows=rows, grid_cols=cols, tally_ways=8))
    accs = [[model.Acceptor(g, i) for i in range(R)] for g in range(G)]
    if kind == Q_THRESHOLD:
        qs = None
    elif kind == Q_MAJORITY:
        qs = lambda g: model.SimpleMajority(range(R))
    elif kind == Q_UNANIMOUS:
        qs = lambda g: model.UnanimousWrites(range(R))
    else:
        qs = lambda g: model.Grid([[r * cols + c for c in range(cols)] for r in range(rows)])
    proxy = model.ProxyLeader(f, qs)
    rng = np.random.default_rng(R * 100 + G * 10 + kind)
    i32 = lambda x: np.array([x], np.int32)
    seen = dict(vote=0, nack=0, chosen=0, after_done=0, dup_open=0, revote=0, fatal=0)
    open_keys = []
    for step in range(500):
        what = int(rng.integers(0, 10))
        rnd = int(rng.integers(0, 5))
        if what == 0:                                   # a competing leader's Phase1a
            g = int(rng.integers(0, G))
            t = [i for i in range(R) if rng.random() < 0.5] or [0]
            st, pb, nb = ref.acceptor_phase1a(g, rnd, 0, oracle.bits_of(t))
            want = {i: accs[g][i].handle_phase1a(rnd) for i in t}
            assert st == 0 and bits_to_indices(pb) == [i for i in t if want[i][0] == "phase1b"]
            assert bits_to_indices(nb) == [i for i in t if want[i][0] == "nack"]
        elif what == 1:                                 # a Phase2b for a (slot, round) nobody proposed
            slot = int(rng.integers(0, S))
            st, ch, cr, cv = ref.proxy_phase2b(i32(slot), i32(rnd + 9), oracle.bits_of([0]).reshape(1, 4))
            assert st == 2 and proxy.handle_phase2b(slot, rnd + 9, slot % G, 0) == "fatal"
            seen["fatal"] += 1
        elif what == 2 and open_keys:                   # a late / duplicate Phase2b for a known key
            slot, r0 = open_keys[int(rng.integers(0, len(open_keys)))]
            i = int(rng.integers(0, R))
            st, ch, cr, cv = ref.proxy_phase2b(i32(slot), i32(r0), oracle.bits_of([i]).reshape(1, 4))
            was_done = proxy.states[(slot, r0)] == "done"
            got = proxy.handle_phase2b(slot, r0, slot % G, i)
            assert st == 0 and bool(ch[0]) == isinstance(got, tuple)
            seen["after_done"] += was_done
            seen["chosen"] += isinstance(got, tuple)
        else:                                           # a Phase2a (new, a re-proposal, or a duplicate)
            if open_keys and rng.random() < 0.3:
                slot, r0 = open_keys[int(rng.integers(0, len(open_keys)))]
                rnd = r0 if rng.random() < 0.5 else rnd
            else:
                slot = int(rng.integers(0, S))
            g = slot % G
            value = slot * 7 + 1                         # a re-proposal carries the same value (Paxos-safe)
            st, new = ref.proxy_open(i32(slot), i32(rnd), i32(value))
            assert st == 0 and bool(new[0]) == proxy.handle_phase2a(slot, rnd, value)
            seen["dup_open"] += not new[0]
            if new[0]:
                open_keys.append((slot, rnd))
            t = [i for i in range(R) if rng.random() < 0.75] or [int(rng.integers(0, R))]
            st, vb, nb, nr = ref.acceptor_phase2a(i32(slot), i32(rnd), i32(value), oracle.bits_of(t).reshape(1, 4))
            assert st == 0
            before = {i: accs[g][i].states.get(slot) for i in t}
            want = {i: accs[g][i].handle_phase2a(slot, rnd, value) for i in t}
            voters = [i for i in t if want[i][0] == "phase2b"]
            nackers = [i for i in t if want[i][0] == "nack"]
            assert bits_to_indices(vb[0]) == voters and bits_to_indices(nb[0]) == nackers, step
            assert nr[0] == (max(want[i][1] for i in nackers) if nackers else -1)
            seen["vote"] += len(voters)
            seen["nack"] += len(nackers)
            seen["revote"] += sum(1 for i in voters if before[i] is not None and before[i][0] == rnd)
            order = list(voters)
            rng.shuffle(order)
            for i in order + order[:1]:                  # one by one, the first one twice
                st, ch, cr, cv = ref.proxy_phase2b(i32(slot), i32(rnd), oracle.bits_of([i]).reshape(1, 4))
                got = proxy.handle_phase2b(slot, rnd, g, i)
                assert st == 0 and bool(ch[0]) == isinstance(got, tuple), (step, i)
                if isinstance(got, tuple):
                    assert cr[0] == rnd and cv[0] == got[1]
                    seen["chosen"] += 1
    # a new leader's recovery (Leader.scala:306-329, 543-566) from the Phase1b's of random acceptor subsets
    for watermark in (0, 5, S // 2, S - 3, S + 4):
        who = [[i for i in range(R) if rng.random() < 0.6] or [0] for _ in range(G)]
        masks = np.stack([oracle.bits_of(w) for w in who])
        st, mx, sr, sv = ref.leader_phase1b_scan(watermark, masks, S)
        want_max, want = model.leader_recovery([[model.phase1b(accs[g][i], watermark) for i in who[g]] for g in range(G)],
                                               watermark, G)
        assert st == 0 and mx == want_max
        assert list(zip(sr.tolist(), sv.tolist())) == want
    vr, vv, _ = ref.read_state()
    pr, mv = ref.read_scalars()
    for g in range(G):
        for i in range(R):
            a = accs[g][i]
            assert pr[g][i] == a.round and mv[g][i] == a.max_voted_slot
            for slot in range(S):
                if slot in a.states:
                    assert slot % G == g and (vr[slot][i], vv[slot][i]) == a.states[slot]
                elif slot % G == g:
                    assert vr[slot][i] == -1
    COVERAGE.append(seen)


def test_the_multipaxos_scenarios_reached_every_branch():
    total = {k: sum(c[k] for c in COVERAGE) for k in COVERAGE[0]} if COVERAGE else {}
    assert COVERAGE and all(v > 0 for v in total.values()), sorted(total.items())
