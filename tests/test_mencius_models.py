"""oracle/fpx_oracle.c (flat arrays, 256-bit vote sets, hashed range table) against oracle/mencius_maps.py (one object
per acceptor with a states map, a proxy leader with Map[(start, end, round), State]): random streams of single-slot
Phase2a / Phase2b, noop ranges, competing Phase1a's, duplicate opens, length-one ranges that collide with single-slot
keys, Phase2b's nobody asked for -- every reply and the final state of every acceptor.  CPU only."""
import numpy as np
import pytest

from oracle import mencius_maps as model

COVERAGE = []


def bits_to_indices(words):
    return [j for j in range(256) if (int(words[j >> 6]) >> (j & 63)) & 1]


@pytest.mark.parametrize("L,A,R,f,seed", [(3, 2, 3, 1, 1), (2, 1, 3, 1, 2), (4, 3, 5, 2, 3), (1, 2, 3, 1, 4), (5, 2, 4, 1, 5)])
def test_flat_oracle_and_map_model_agree(oracle, L, A, R, f, seed):
    S = 240
    ref = oracle.System(oracle.make_config(num_slots=S, num_replicas=R, num_groups=A, num_leader_groups=L, f=f, tally_ways=8))
    mod = model.Mencius(L, A, R, f)
    rng = np.random.default_rng(seed)
    i32 = lambda x: np.array([x], np.int32)
    seen = dict(vote=0, nack=0, range_vote=0, range_nack=0, chosen=0, chosen_range=0, fatal=0, dup_open=0, swallowed=0)

    def targets():
        t = [i for i in range(R) if rng.random() < 0.7]
        return t or [int(rng.integers(0, R))]

    for step in range(400):
        kind = int(rng.integers(0, 6))
        rnd = int(rng.integers(0, 5))
        if kind == 0:                                   # a competing leader's Phase1a at some acceptors of one group
            lg, ag = int(rng.integers(0, L)), int(rng.integers(0, A))
            t = targets()
            st, pb, nb = ref.acceptor_phase1a(lg * A + ag, rnd, 0, oracle.bits_of(t))
            assert st == 0
            want = {i: mod.acceptors[lg][ag][i].handle_phase1a(rnd) for i in t}
            assert bits_to_indices(pb) == [i for i in t if want[i][0] == "phase1b"]
            assert bits_to_indices(nb) == [i for i in t if want[i][0] == "nack"]
        elif kind in (1, 2):                            # one single-slot Phase2a: proxy leader, acceptors, Phase2b's back
            slot, value = int(rng.integers(0, S)), int(rng.integers(0, 1000))
            lg, ag = mod.group_of(slot)
            st, new = ref.proxy_open(i32(slot), i32(rnd), i32(value))
            assert st == 0 and bool(new[0]) == mod.proxy.handle_phase2a(slot, rnd, value)
            seen["dup_open"] += not new[0]
            t = targets()
            st, vb, nb, nr = ref.acceptor_phase2a(i32(slot), i32(rnd), i32(value), oracle.bits_of(t).reshape(1, 4))
            assert st == 0
            want = {i: mod.acceptors[lg][ag][i].handle_phase2a(slot, rnd, value) for i in t}
            voters = [i for i in t if want[i][0] == "phase2b"]
            nackers = [i for i in t if want[i][0] == "nack"]
            assert bits_to_indices(vb[0]) == voters and bits_to_indices(nb[0]) == nackers
            assert nr[0] == (max(want[i][1] for i in nackers) if nackers else -1)
            seen["vote"] += len(voters)
            seen["nack"] += len(nackers)
            # the Phase2b's arrive one by one
            for i in voters:
                st, ch, cr, cv = ref.proxy_phase2b(i32(slot), i32(rnd), oracle.bits_of([i]).reshape(1, 4))
                got = mod.proxy.handle_phase2b(slot, rnd, i)
                assert st == (2 if got == "fatal" else 0)
                if isinstance(got, tuple):
                    assert ch[0] == 1 and cr[0] == rnd and cv[0] == got[1]
                    seen["chosen"] += 1
                else:
                    assert ch[0] == 0
                    seen["swallowed"] += got is None and not new[0]
        elif kind in (3, 4):                            # a noop range of one leader group
            lg = int(rng.integers(0, L))
            start = lg + L * int(rng.integers(0, S // L - 14))
            end = start + (1 if kind == 4 and rng.random() < 0.4 else 1 + int(rng.integers(0, 12 * L)))
            st, new = ref.proxy_open_noop_range(start, end, rnd)
            assert st == 0 and bool(new) == mod.proxy.handle_phase2a_noop_range(start, end, rnd)
            seen["dup_open"] += not new
            tm = np.zeros((A, 4), np.uint64)
            tg = [targets() for _ in range(A)]
            for ag in range(A):
                tm[ag] = oracle.bits_of(tg[ag])
            st, vb, nb, nr = ref.acceptor_phase2a_noop_range(start, end, rnd, tm)
            assert st == 0
            nack_rounds = []
            for ag in range(A):
                want = {i: mod.acceptors[lg][ag][i].handle_phase2a_noop_range(start, end, rnd) for i in tg[ag]}
                voters = [i for i in tg[ag] if want[i][0] != "nack"]
                nackers = [i for i in tg[ag] if want[i][0] == "nack"]
                assert bits_to_indices(vb[ag]) == voters and bits_to_indices(nb[ag]) == nackers, (step, ag)
                nack_rounds += [want[i][1] for i in nackers]
                seen["range_vote"] += len(voters)
                seen["range_nack"] += len(nackers)
                for i in voters:
                    one = np.zeros((A, 4), np.uint64)
                    one[ag] = oracle.bits_of([i])
                    st, ch = ref.proxy_phase2b_noop_range(start, end, rnd, one)
                    got = mod.proxy.handle_phase2b_noop_range(start, end, rnd, ag, i)
                    assert st == (2 if got == "fatal" else 0) and bool(ch) == isinstance(got, tuple), (step, ag, i)
                    seen["chosen_range"] += isinstance(got, tuple)
            assert nr == (max(nack_rounds) if nack_rounds else -1)
        else:                                           # a Phase2b / Phase2bNoopRange nobody asked for
            slot = int(rng.integers(0, S))
            if rng.random() < 0.5:
                st, ch, cr, cv = ref.proxy_phase2b(i32(slot), i32(rnd + 7), oracle.bits_of([0]).reshape(1, 4))
                assert st == 2 and mod.proxy.handle_phase2b(slot, rnd + 7, 0) == "fatal"
            else:
                one = np.zeros((A, 4), np.uint64)
                one[0] = oracle.bits_of([0])
                st, ch = ref.proxy_phase2b_noop_range(slot, slot + 3, rnd + 7, one)
                assert st == 2 and mod.proxy.handle_phase2b_noop_range(slot, slot + 3, rnd + 7, 0, 0) == "fatal"
            seen["fatal"] += 1
    # every acceptor: round, max voted slot, votes slot by slot
    vr, vv, _ = ref.read_state()
    pr, mv = ref.read_scalars()
    for lg in range(L):
        for ag in range(A):
            for i in range(R):
                acc = mod.acceptors[lg][ag][i]
                assert pr[lg * A + ag][i] == acc.round
                assert mv[lg * A + ag][i] == (max(acc.states) if acc.states else -1)
                for slot, (r0, v0) in acc.states.items():
                    assert mod.group_of(slot) == (lg, ag)
                    assert (vr[slot][i], vv[slot][i]) == (r0, v0)
    for slot in range(S):
        lg, ag = mod.group_of(slot)
        for i in range(R):
            if slot not in mod.acceptors[lg][ag][i].states:
                assert vr[slot][i] == -1
    COVERAGE.append(seen)


def test_the_mencius_scenarios_reached_every_branch():
    total = {k: sum(c[k] for c in COVERAGE) for k in COVERAGE[0]} if COVERAGE else {}
    assert COVERAGE and all(v > 0 for v in total.values()), sorted(total.items())
