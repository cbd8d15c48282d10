"""K4: Mencius noop ranges:
mencius/Acceptor.scala:237-291, mencius/ProxyLeader.scala:255-303, 355-411."""
import numpy as np
import pytest

from tests import workloads as W

L, A, R = 3, 2, 3  # leader groups, acceptor groups per leader group, acceptors per group (f = 1)
KW = dict(num_slots=256, num_replicas=R, num_groups=A, num_leader_groups=L, f=1, tally_ways=8)


def test_oracle_noop_range_by_hand(oracle):
    s = oracle.System(oracle.make_config(**KW))
    # leader group 1 owns slots 1, 4, 7, 10, ...; acceptor group of slot = (slot / 3) % 2:
    # 1 -> 0, 4 -> 1, 7 -> 0, 10 -> 1, 13 -> 0
    # acceptor (lg 1, ag 0, idx 2) was promised round 5 by someone else
    assert s.acceptor_phase1a(1 * A + 0, 5, 0, oracle.bits_of([2]))[0] == 0
    st, vb, nb, nr = s.acceptor_phase2a_noop_range(4, 14, 2)
    assert st == 0 and nr == 5
    assert oracle.indices_of(vb[0]) == [0, 1] and oracle.indices_of(nb[0]) == [2]   # ag 0
    assert oracle.indices_of(vb[1]) == [0, 1, 2] and oracle.indices_of(nb[1]) == []  # ag 1
    vr, vv, _ = s.read_state()
    assert vr[7].tolist() == [2, 2, -1] and vr[13].tolist() == [2, 2, -1]       # ag 0 slots in [4, 14)
    assert vr[4].tolist() == [2, 2, 2] and vr[10].tolist() == [2, 2, 2]        # ag 1 slots
    assert (vv[[4, 7, 10, 13]][vr[[4, 7, 10, 13]] == 2] == -1).all()            # Noop
    assert (vr[[1, 5, 6, 8, 16]] == -1).all()                                   # outside / other leader groups
    pr, mv = s.read_scalars()
    assert pr[2].tolist() == [2, 2, 5] and pr[3].tolist() == [2, 2, 2] and (pr[[0, 1, 4, 5]] == -1).all()
    assert mv[2].tolist() == [13, 13, -1] and mv[3].tolist() == [10, 10, 10]
    # proxy leader: a quorum (f + 1 = 2) from EVERY acceptor group
    assert s.proxy_phase2b_noop_range(4, 14, 2, vb)[0] == 2                      # never opened: fatal
    assert s.proxy_open_noop_range(4, 14, 2) == (0, 1)
    assert s.proxy_open_noop_range(4, 14, 2) == (0, 0)                           # known: ignored
    one = np.zeros((A, 4), np.uint64)
    one[0] = oracle.bits_of([0, 1])
    assert s.proxy_phase2b_noop_range(4, 14, 2, one) == (0, 0)                   # ag 1 has no votes yet
    one[0] = 0
    one[1] = oracle.bits_of([2])
    assert s.proxy_phase2b_noop_range(4, 14, 2, one) == (0, 0)                   # ag 1: 1 < 2
    one[1] = oracle.bits_of([2, 0])
    assert s.proxy_phase2b_noop_range(4, 14, 2, one) == (0, 1)                   # ChosenNoopRange
    assert s.proxy_phase2b_noop_range(4, 14, 2, one) == (0, 0)                   # Done: ignored


def test_oracle_length_one_range_collides_with_single_slot_key(oracle):
    """SlotRound(slot, slot + 1, round) is the key of both (mencius/ProxyLeader.scala:86-90)"""
    s = oracle.System(oracle.make_config(**KW))
    i32 = lambda x: np.array([x], np.int32)
    assert s.proxy_open(i32(7), i32(0), i32(55))[1][0] == 1
    assert s.proxy_open_noop_range(7, 8, 0) == (0, 0)          # single-slot Phase2a pending: ignored
    votes = np.zeros((A, 4), np.uint64)
    votes[:] = oracle.bits_of([0, 1, 2])
    assert s.proxy_phase2b_noop_range(7, 8, 0, votes) == (0, 0)  # swallowed by the PendingPhase2a
    assert s.proxy_open_noop_range(10, 11, 3) == (0, 1)
    assert s.proxy_open(i32(10), i32(3), i32(1))[1][0] == 0    # the range owns the key
    st, ch, cr, cv = s.proxy_phase2b(i32(10), i32(3), W.bits_from_bool(np.ones((1, 3), bool)))
    assert st == 0 and ch[0] == 0                              # PendingPhase2aNoopRange: ignored
    assert s.proxy_phase2b_noop_range(10, 11, 3, votes) == (0, 1)


@pytest.fixture(scope="module")
def fa():
    import frankenpaxos_amd

    frankenpaxos_amd.lib()
    return frankenpaxos_amd


@pytest.mark.gpu
def test_noop_ranges_match_oracle(fa, oracle):
    gpu = fa.Context(fa.make_config(**KW))
    ref = oracle.System(oracle.make_config(**KW))
    rng = np.random.default_rng(5)
    S = KW["num_slots"]
    i32 = lambda x: np.array(x, np.int32)
    for step in range(120):
        kind = rng.integers(0, 6)
        lg = int(rng.integers(0, L))
        start = lg + L * int(rng.integers(0, S // L - 2))
        end = min(S, start + L * int(rng.integers(0, 12)) + int(rng.integers(0, 2)) * (1 - L))
        end = max(end, start)
        rnd = int(rng.integers(0, 4))
        if kind == 0:    # a competing leader's Phase1a on some acceptors
            g = int(rng.integers(0, L * A))
            t = W.bits_from_bool(W.random_subsets(rng, 1, R, 1, R))[0]
            assert gpu.acceptor_phase1a(g, rnd, 0, t)[0] == ref.acceptor_phase1a(g, rnd, 0, t)[0]
        elif kind in (1, 2):
            tm = W.bits_from_bool(W.random_subsets(rng, A, R, 1, R)) if kind == 2 else None
            a = gpu.acceptor_phase2a_noop_range(start, end, rnd, tm)
            b = ref.acceptor_phase2a_noop_range(start, end, rnd, tm)
            assert a[0] == b[0] and a[3] == b[3]
            np.testing.assert_array_equal(a[1], b[1])
            np.testing.assert_array_equal(a[2], b[2])
            assert gpu.proxy_open_noop_range(start, end, rnd) == ref.proxy_open_noop_range(start, end, rnd)
            half = a[1].copy()
            half[:, 0] &= np.uint64(rng.integers(0, 8))
            for votes in (half, a[1], a[1]):
                assert gpu.proxy_phase2b_noop_range(start, end, rnd, votes) == \
                    ref.proxy_phase2b_noop_range(start, end, rnd, votes)
        elif kind == 3:  # ordinary single-slot traffic interleaved, including colliding keys
            slots = i32(sorted(set(rng.integers(0, S, 20).tolist()) | {start}))
            rr = np.full(len(slots), rnd, np.int32)
            W.assert_same_outputs(W.run_script(gpu, [("fused", slots, rr, slots * 3, None)]),
                                  W.run_script(ref, [("fused", slots, rr, slots * 3, None)]))
        elif kind == 4:  # length-one ranges
            assert gpu.proxy_open_noop_range(start, start + 1, rnd) == ref.proxy_open_noop_range(start, start + 1, rnd)
            votes = np.zeros((A, 4), np.uint64)
            votes[:] = oracle.bits_of([0, 1, 2])
            assert gpu.proxy_phase2b_noop_range(start, start + 1, rnd, votes) == \
                ref.proxy_phase2b_noop_range(start, start + 1, rnd, votes)
            vb = W.bits_from_bool(np.ones((1, 3), bool))
            a = gpu.proxy_phase2b(i32([start]), i32([rnd]), vb)
            b = ref.proxy_phase2b(i32([start]), i32([rnd]), vb)
            assert a[0] == b[0] and a[1][0] == b[1][0]
        else:            # Phase2bNoopRange for a key nobody opened: fatal
            votes = np.zeros((A, 4), np.uint64)
            votes[0] = oracle.bits_of([1])
            a = gpu.proxy_phase2b_noop_range(start, end + 1000 if end + 1000 <= S else end, 9, votes)
            b = ref.proxy_phase2b_noop_range(start, end + 1000 if end + 1000 <= S else end, 9, votes)
            assert a == b
    W.assert_same_state(gpu, ref, tally_slots=range(0, S, 7))


@pytest.mark.gpu
def test_noop_range_arguments(fa):
    gpu = fa.Context(fa.make_config(**KW))
    assert gpu.acceptor_phase2a_noop_range(5, 4, 0)[0] == fa.FPX_EINVAL
    assert gpu.acceptor_phase2a_noop_range(0, 257, 0)[0] == fa.FPX_EINVAL
    assert gpu.proxy_open_noop_range(0, 4, -1)[0] == fa.FPX_EINVAL
    per_slot = fa.Context(fa.make_config(ballot_mode=1, **KW))
    assert per_slot.acceptor_phase2a_noop_range(0, 4, 0)[0] == fa.FPX_EINVAL


# ------------------------------------------------ the receiver: Replica.handleChosenNoopRange ------
def test_oracle_replica_chosen_noop_range_by_hand(oracle):
    """mencius/Replica.scala:464-485, including its early `return` on a slot that is already chosen"""
    s = oracle.System(oracle.make_config(**KW))            # 3 leader groups: stride 3
    assert s.replica_chosen_noop_range(1, 11) == (0, 0, 4)  # slots 1, 4, 7, 10 <- Noop; hole at 0
    vals, pres = s.replica_read_log(0, 12)
    assert pres.tolist() == [0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0] and (vals[pres == 1] == -1).all()
    assert s.replica_chosen([0, 2, 3], [50, 52, 53])[1:] == (5, 7)   # prefix 0..4 executes
    # leader group 2 skips 2, 5, 8, 11 -- but 2 is already chosen: the handler returns at once,
    # nothing is put and executeLog does not run
    assert s.replica_chosen_noop_range(2, 12) == (0, 5, 7)
    assert s.replica_read_log(5, 1)[1].tolist() == [0]
    # from 5 on: 5, 8, 11 are put; executeLog: 5 -> watermark 6 (6 is missing)
    assert s.replica_chosen_noop_range(5, 12) == (0, 6, 10)
    # a range that hits a chosen slot in the middle: 13 is put, 16 is present -> return; 19 is NOT put,
    # and executeLog is skipped although nothing blocks it
    assert s.replica_chosen([16, 6], [66, 56])[1:] == (9, 12)
    assert s.replica_chosen([9], [59])[1:] == (12, 13)
    assert s.replica_chosen([12], [62])[1:] == (13, 14)
    assert s.replica_chosen_noop_range(13, 22) == (0, 13, 15)       # 13 put, watermark NOT advanced
    assert s.replica_read_log(19, 1)[1].tolist() == [0]
    assert s.replica_chosen_noop_range(30, 30) == (0, 14, 15)       # empty range: executeLog only
    assert s.replica_chosen_noop_range(-1, 4)[0] == 1 and s.replica_chosen_noop_range(0, 257)[0] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("stride", [1, 3])
def test_replica_chosen_noop_range_matches_oracle(fa, oracle, stride):
    kw = dict(KW, num_leader_groups=stride, num_slots=4096)
    gpu = fa.Context(fa.make_config(**kw))
    ref = oracle.System(oracle.make_config(**kw))
    rng = np.random.default_rng(17 + stride)
    S = kw["num_slots"]
    for step in range(300):
        if rng.random() < 0.5:
            n = int(rng.integers(1, 40))
            slot = rng.integers(0, min(S, 64 + step * 14), n).astype(np.int32)
            val = rng.integers(0, 1000, n).astype(np.int32)
            assert gpu.replica_chosen(slot, val) == ref.replica_chosen(slot, val)
        else:
            start = int(rng.integers(0, min(S - 1, 32 + step * 14)))
            end = min(S, start + int(rng.integers(0, 200)))
            assert gpu.replica_chosen_noop_range(start, end) == ref.replica_chosen_noop_range(start, end)
    a, b = gpu.replica_read_log(0, S), ref.replica_read_log(0, S)
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[0][a[1] == 1], b[0][b[1] == 1])
    assert gpu.replica_chosen_noop_range(0, S + 1)[0] == fa.FPX_EINVAL
