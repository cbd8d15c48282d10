"""Hand-computed micro-traces for the two functions the reference does NOT pin with golden vectors
(SURVEY.md section 8c): Acceptor.handlePhase2a and ProxyLeader.handlePhase2b.  Every expected
value below was derived by reading the Scala, not by running the oracle.  CPU only.

Reference: shared/src/main/scala/frankenpaxos/multipaxos/{Acceptor.scala:184-220, ProxyLeader.scala:175-258}
"""
import numpy as np
import pytest

from tests import workloads as W


def sys3(oracle, **kw):
    """f = 1: one group of 3 acceptors, threshold quorum 2"""
    return oracle.System(oracle.make_config(num_slots=16, num_replicas=3, f=1, **kw))


def test_acceptor_vote_nack_revote(oracle):
    s = sys3(oracle)
    # Acceptor.scala:95 round = -1: the first Phase2a in round 0 is accepted (0 < -1 is false)
    assert s.acceptor_handle_phase2a(0, 0, slot=4, round_=0, value=40) == (True, 0)
    p, mvs, vr, vv, _ = s.read_acceptor(0, 0)
    assert (p, mvs, vr[4], vv[4]) == (0, 4, 0, 40)
    # :192 "phase2a.round < round": an EQUAL round re-votes and overwrites (:205-208)
    assert s.acceptor_handle_phase2a(0, 0, 4, 0, 41) == (True, 0)
    assert s.read_acceptor(0, 0)[3][4] == 41
    # a higher round moves the acceptor's round for ALL its slots (the scalar of :95)
    assert s.acceptor_handle_phase2a(0, 0, 2, 3, 77) == (True, 3)
    p, mvs, vr, vv, _ = s.read_acceptor(0, 0)
    assert (p, mvs, vr[2], vv[2]) == (3, 4, 3, 77)  # maxVotedSlot stays 4 (:209 max)
    # a stale round in ANOTHER slot is Nacked with the acceptor's round (:192-200); no state change
    assert s.acceptor_handle_phase2a(0, 0, 9, 2, 99) == (False, 3)
    p, mvs, vr, vv, _ = s.read_acceptor(0, 0)
    assert (p, mvs, vr[9], vv[9]) == (3, 4, -1, -1)
    # the other acceptors are untouched
    assert s.read_acceptor(0, 1)[:2] == (-1, -1)


def test_acceptor_per_slot_ballots(oracle):
    """PER_SLOT: the ballot lives in the cell (epaxos/Replica.scala:1443-1448, 1488-1497)"""
    s = sys3(oracle, ballot_mode=1)
    assert s.acceptor_handle_phase2a(0, 0, 2, 3, 77) == (True, 3)
    # a lower round in ANOTHER slot is accepted: that cell's ballot is still -1
    assert s.acceptor_handle_phase2a(0, 0, 9, 2, 99) == (True, 2)
    # ... but is Nacked in the SAME slot, with that cell's ballot
    assert s.acceptor_handle_phase2a(0, 0, 2, 1, 55) == (False, 3)
    _, mvs, vr, vv, bl = s.read_acceptor(0, 0)
    assert (mvs, vr[2], vv[2], bl[2], vr[9], bl[9]) == (9, 3, 77, 3, 2, 2)


def test_phase1a_moves_the_round(oracle):
    s = sys3(oracle)
    assert s.acceptor_handle_phase1a(0, 1, 5) == (True, 5)  # Acceptor.scala:166
    assert s.acceptor_handle_phase1a(0, 1, 5) == (True, 5)  # equal round: promised again (:155 is <)
    assert s.acceptor_handle_phase1a(0, 1, 4) == (False, 5)  # stale: Nack(round)
    assert s.acceptor_handle_phase2a(0, 1, 0, 4, 1) == (False, 5)
    assert s.acceptor_handle_phase2a(0, 1, 0, 5, 1) == (True, 5)


def test_proxy_leader_tally(oracle):
    s = sys3(oracle)
    # ProxyLeader.scala:220-225: a Phase2b for a (slot, round) never opened is fatal
    assert s.proxy_handle_phase2b(0, slot=1, round_=0)[0] == -1
    assert s.proxy_handle_phase2a(1, 0, 500) is True      # :213 Pending
    assert s.proxy_handle_phase2a(1, 0, 501) is False     # :177-184 already known: ignored
    assert s.proxy_handle_phase2b(2, 1, 0) == (0, -1)     # one vote: size 1 < f+1 = 2 (:238)
    assert s.proxy_handle_phase2b(2, 1, 0) == (0, -1)     # duplicate vote: map key collapses (:237)
    assert s.read_tally(1) == [(0, 0, 500, (4, 0, 0, 0))]
    assert s.proxy_handle_phase2b(0, 1, 0) == (1, 500)    # quorum: Chosen(slot, pending.phase2a.value)
    assert s.read_tally(1) == [(0, 1, -1, (0, 0, 0, 0))]  # :256 Done
    assert s.proxy_handle_phase2b(1, 1, 0) == (2, -1)     # :227-232 after Done: ignored
    # a re-proposal in a higher round is a NEW tally keyed by (slot, round) (:87,135)
    assert s.proxy_handle_phase2b(0, 1, 2)[0] == -1
    assert s.proxy_handle_phase2a(1, 2, 500) is True
    assert s.proxy_handle_phase2b(1, 1, 2) == (0, -1)
    assert s.proxy_handle_phase2b(2, 1, 2) == (1, 500)
    assert [t[:2] for t in s.read_tally(1)] == [(0, 1), (2, 1)]


def test_old_round_tally_stays_pending_and_can_still_complete(oracle):
    """ProxyLeader.states keeps every (slot, round): a superseded tally is not cancelled."""
    s = sys3(oracle)
    s.proxy_handle_phase2a(3, 0, 7)
    assert s.proxy_handle_phase2b(0, 3, 0) == (0, -1)
    s.proxy_handle_phase2a(3, 1, 7)                       # re-proposed in round 1
    assert s.proxy_handle_phase2b(1, 3, 0) == (1, 7)      # a late round-0 vote completes round 0
    assert s.proxy_handle_phase2b(1, 3, 1) == (0, -1)
    assert [t[:2] for t in s.read_tally(3)] == [(0, 1), (1, 0)]


def test_fused_batch_trace(oracle):
    """one fused batch, by hand: acceptor 2 was promised round 1 by a competing leader"""
    s = sys3(oracle)
    tgt_only2 = W.bits_from_bool(np.array([[False, False, True]]))[0]
    assert s.acceptor_phase1a(0, 1, 0, tgt_only2)[0] == 0
    slot = np.array([0, 1, 2, 1], np.int32)
    rnd = np.array([0, 0, 0, 0], np.int32)
    val = np.array([10, 11, 12, 99], np.int32)
    tgt = W.bits_from_bool(np.array([[True, True, True],     # 0,1 vote, 2 nacks -> chosen
                                     [True, False, True],    # 0 votes, 2 nacks  -> pending
                                     [False, False, True],   # only a nack       -> pending
                                     [True, True, True]]))   # duplicate (1, 0): ignored, not forwarded
    st, ch, cr, cv, nr = s.phase2_fused(slot, rnd, val, tgt)
    assert st == 0
    assert ch.tolist() == [1, 0, 0, 0]
    assert cr.tolist() == [0, -1, -1, -1]
    assert cv.tolist() == [10, -1, -1, -1]
    assert nr.tolist() == [1, 1, 1, -1]
    vr, vv, _ = s.read_state()
    assert vr.tolist()[:3] == [[0, 0, -1], [0, -1, -1], [-1, -1, -1]]
    assert vv.tolist()[:3] == [[10, 10, -1], [11, -1, -1], [-1, -1, -1]]
    pr, mv = s.read_scalars()
    assert pr.tolist() == [[0, 0, 1]] and mv.tolist() == [[1, 0, -1]]
    assert s.read_tally(1) == [(0, 0, 11, (1, 0, 0, 0))]
    # the missing vote arrives later through K2
    st, ch, cr, cv = s.proxy_phase2b(np.array([1], np.int32), np.array([0], np.int32),
                                     W.bits_from_bool(np.array([[False, True, False]])))
    assert (st, ch[0], cr[0], cv[0]) == (0, 1, 0, 11)


@pytest.mark.parametrize("ballot_mode", [0, 1])
@pytest.mark.parametrize("R,kw", [(3, dict(f=1)), (7, dict(quorum_kind=1)),
                                  (6, dict(quorum_kind=2, grid_rows=2, grid_cols=3)),
                                  (4, dict(quorum_kind=3))])
def test_fifo_pump_equals_sequential_delivery(oracle, ballot_mode, R, kw):
    """FakeTransport-style FIFO drain (all Phase2a's first) == per-message delivery: the chosen set
    and the final state do not depend on the interleaving when slots are distinct."""
    S = 512
    mk = lambda: oracle.System(oracle.make_config(num_slots=S, num_replicas=R, ballot_mode=ballot_mode, **kw))
    a, b = mk(), mk()
    rng = np.random.default_rng(R)
    slot = rng.permutation(S).astype(np.int32)
    rnd = np.zeros(S, np.int32)
    val = (slot * 7 + 1).astype(np.int32)
    tgt = W.bits_from_bool(W.random_subsets(rng, S, R, 1, R))
    for be in (a, b):
        be.acceptor_phase1a(0, 2, 0, W.bits_from_bool(np.array([[True] + [False] * (R - 1)]))[0])
    out_a = a.phase2_fused(slot, rnd + 1, val, tgt)
    out_b = b.phase2_fifo_pump(slot, rnd + 1, val, tgt)
    for x, y in zip(out_a, out_b):
        np.testing.assert_array_equal(x, y)
    W.assert_same_state(a, b, tally_slots=range(0, S, 13))


def test_mencius_slot_to_group_map(oracle):
    """mencius/ProxyLeader.scala:169-176,231-234 ; mencius/Leader.scala slotSystem = round robin"""
    cfg = oracle.make_config(num_slots=64, num_replicas=3, num_groups=2, num_leader_groups=3, f=1)
    s = oracle.System(cfg)
    for slot in range(64):
        lg, ag = slot % 3, (slot // 3) % 2
        assert s.group_of_slot(slot) == lg * 2 + ag
    # multipaxos: slot % numAcceptorGroups (multipaxos/ProxyLeader.scala:190)
    s = oracle.System(oracle.make_config(num_slots=64, num_replicas=3, num_groups=5, f=1))
    assert [s.group_of_slot(x) for x in range(12)] == [x % 5 for x in range(12)]


def test_flat_oracle_agrees_with_the_reference_shaped_restatement(oracle):
    """Two independent restatements of the same handlers -- the flat-array oracle the parity tests use
    and oracle/fpx_faithful.cpp, which keeps the reference's data-structure shapes (a sorted map per
    acceptor, a hash map of Pending/Done keyed by (slot, round), one heap message per Phase2a / Phase2b
    through a FIFO transport) -- must choose the same values on the steady stream."""
    import ctypes as C
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "oracle", "libfpx_faithful.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "libfpx_faithful.so"],
                              stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    L.fpo_faithful_run.restype = C.c_int64
    L.fpo_faithful_run.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_int64)]
    for S, R, f in ((500, 3, 1), (300, 7, 3), (64, 255, 127)):
        slot, rnd, val = W.steady_stream(S)
        cs = C.c_int64()
        n = L.fpo_faithful_run(S, R, f, val.ctypes.data, C.byref(cs))
        sys_ = oracle.System(oracle.make_config(num_slots=S, num_replicas=R, f=f))
        assert sys_.acceptor_phase1a(0, 0)[0] == 0
        st, ch, cr, cv, nr = sys_.phase2_fused(slot, rnd, val)
        assert st == 0 and n == int(ch.sum()) == S
        assert cs.value == int(cv[ch == 1].astype(np.int64).sum())


def test_oracle_phase1a_loop_order_is_immaterial(oracle):
    """the batch Phase1a of a big PER_SLOT window walks slots outside / acceptors inside; the per-acceptor
    handler (Acceptor.scala:148-182 restated in fpo_acceptor_handle_phase1a) walks one acceptor's column.
    Same cells, same rule, no shared state: identical results (checked on a window just over the switch)."""
    import numpy as np
    from tests import workloads as W

    S, R = 4096, 9
    kw = dict(num_slots=S, num_replicas=R, num_groups=2, f=4, ballot_mode=1)
    a, b = oracle.System(oracle.make_config(**kw)), oracle.System(oracle.make_config(**kw))
    rng = np.random.default_rng(5)
    for step in range(12):
        g, rnd, wm = int(rng.integers(0, 2)), int(rng.integers(0, 6)), int(rng.integers(0, S // 2))
        tgt = W.bits_from_bool(rng.random((1, R)) < 0.6)[0]
        sa, pa, na = a.acceptor_phase1a(g, rnd, wm, tgt)
        pb, nb = [], []
        for r in range(R):          # the per-message handler, acceptor by acceptor
            if (int(tgt[0]) >> r) & 1:
                ok, _ = b.acceptor_handle_phase1a(g, r, rnd, wm)
                (pb if ok else nb).append(r)
        assert oracle.indices_of(pa) == pb and oracle.indices_of(na) == nb
        for x, y in zip(a.read_state(), b.read_state()):
            np.testing.assert_array_equal(x, y)
