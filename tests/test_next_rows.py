"""SURVEY.md section 8(f) "next" rows: f1 replica log ingest + executed watermark, f2 Phase-1
recovery scan.  CPU tests pin the oracle by hand traces; GPU tests compare libfpx with the oracle."""
import numpy as np
import pytest

from tests import workloads as W


# ------------------------------------------------------------------ oracle, by hand (CPU) -------
def test_oracle_replica_log_trace(oracle):
    """multipaxos/Replica.scala:572-590 + 394-404"""
    s = oracle.System(oracle.make_config(num_slots=32, num_replicas=3, f=1))
    st, wm, nc = s.replica_chosen([1, 2, 5], [11, 12, 15])
    assert (st, wm, nc) == (0, 0, 3)                       # hole at slot 0: nothing executes
    st, wm, nc = s.replica_chosen([0, 2, 9], [10, 99, 19], mask=[1, 1, 0])
    assert (st, wm, nc) == (0, 3, 4)                       # 0 fills the hole; 2 is redundant; 9 masked out
    vals, pres = s.replica_read_log(0, 7)
    assert vals.tolist() == [10, 11, 12, -1, -1, 15, -1] and pres.tolist() == [1, 1, 1, 0, 0, 1, 0]
    st, wm, nc = s.replica_chosen([4, 3], [14, 13])
    assert (wm, nc) == (6, 6)
    assert s.replica_chosen([32], [1])[0] == 1             # out of the log window: EINVAL


def test_oracle_phase1b_scan_trace(oracle):
    """multipaxos/Leader.scala:306-329 (safeValue) and :543-566"""
    s = oracle.System(oracle.make_config(num_slots=16, num_replicas=3, f=1))
    # acceptor 0 voted (round 0, 100) in slot 1; acceptor 1 voted (round 2, 200) in slot 1 and
    # (round 2, 300) in slot 4; acceptor 2 voted (round 5, 999) in slot 6
    s.acceptor_handle_phase2a(0, 0, 1, 0, 100)
    s.acceptor_handle_phase2a(0, 1, 1, 2, 200)
    s.acceptor_handle_phase2a(0, 1, 4, 2, 300)
    s.acceptor_handle_phase2a(0, 2, 6, 5, 999)
    q01 = oracle.bits_of([0, 1])
    st, mx, sr, sv = s.leader_phase1b_scan(0, q01, 16)
    assert (st, mx) == (0, 4)                              # acceptor 2 is not in the quorum
    assert sr.tolist() == [-1, 2, -1, -1, 2]
    assert sv.tolist() == [-1, 200, -1, -1, 300]           # holes are filled with Noop (-1)
    st, mx, sr, sv = s.leader_phase1b_scan(2, q01, 16)     # chosenWatermark = 2
    assert (mx, sr.tolist(), sv.tolist()) == (4, [-1, -1, 2], [-1, -1, 300])
    st, mx, sr, sv = s.leader_phase1b_scan(5, q01, 16)     # nothing at or above the watermark
    assert (mx, len(sr)) == (-1, 0)
    st, mx, sr, sv = s.leader_phase1b_scan(0, oracle.bits_of([0, 1, 2]), 4)  # cap
    assert (mx, sr.tolist()) == (6, [-1, 2, -1, -1])


# ------------------------------------------------------------------------------ GPU parity -------
@pytest.fixture(scope="module")
def fa():
    import frankenpaxos_amd

    frankenpaxos_amd.lib()
    return frankenpaxos_amd


def both(fa, oracle, **kw):
    return fa.Context(fa.make_config(**kw)), oracle.System(oracle.make_config(**kw))


@pytest.mark.gpu
def test_replica_log_matches_oracle(fa, oracle):
    S = 1 << 16
    gpu, ref = both(fa, oracle, num_slots=S, num_replicas=3, f=1)
    rng = np.random.default_rng(11)
    order = rng.permutation(S).astype(np.int32)
    vals = W.steady_values(order)
    pos = 0
    for n in (1, 7, 1000, 5000, 20000, S):
        chunk = order[pos:pos + n]
        # duplicates (redundantly chosen, with a different value that must be ignored) and a mask
        extra = order[rng.integers(0, max(1, pos), size=min(50, n))] if pos else chunk[:0]
        slot = np.concatenate([chunk, extra, chunk[:3]])
        val = np.concatenate([vals[pos:pos + n], np.full(len(extra), 7, np.int32), np.full(3, 9, np.int32)])
        mask = (rng.random(len(slot)) < 0.9).astype(np.uint8)
        a = gpu.replica_chosen(slot, val, mask)
        b = ref.replica_chosen(slot, val, mask)
        assert a == b, (n, a, b)
        pos = min(S, pos + n)
    va, pa = gpu.replica_read_log(0, S)
    vb, pb = ref.replica_read_log(0, S)
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(va, vb)
    a = gpu.replica_chosen(order, vals)       # everything: the whole log executes
    assert a == ref.replica_chosen(order, vals) and a[1] == S
    assert gpu.replica_chosen(np.array([S], np.int32), np.array([1], np.int32))[0] == fa.FPX_EINVAL


@pytest.mark.gpu
@pytest.mark.parametrize("ballot_mode", [0, 1])
def test_end_to_end_chosen_feeds_the_replica_log_on_device(fa, oracle, ballot_mode):
    """K3 -> replica log without leaving HBM: `chosen` flags and values of the fused step are the
    mask and the values of the log ingest; the executed watermark covers the whole window."""
    import torch

    S, R = 1 << 18, 256
    gpu = fa.Context(fa.make_config(num_slots=S, num_replicas=R, f=127, ballot_mode=ballot_mode))
    dev = torch.device("cuda:0")
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    slot, rnd, val = W.steady_stream(S)
    perm = np.random.default_rng(3).permutation(S)
    t = [torch.from_numpy(x[perm]).to(dev) for x in (slot, rnd, val)]
    ch = torch.empty(S, dtype=torch.uint8, device=dev)
    cr = torch.empty(S, dtype=torch.int32, device=dev)
    cv = torch.empty(S, dtype=torch.int32, device=dev)
    gpu.phase2_fused_dev(t[0], t[1], t[2], None, ch, cr, cv)
    gpu.replica_chosen_dev(t[0], cv, ch)
    assert gpu.sync() == 0
    assert gpu.replica_state() == (S, S)
    vals, pres = gpu.replica_read_log(0, S)
    assert pres.all() and (vals == val).all()
    gpu.set_stream(None)


@pytest.mark.gpu
@pytest.mark.parametrize("R,kw", [(3, dict(f=1)), (5, dict(quorum_kind=1)), (256, dict(f=127)),
                                  (100, dict(quorum_kind=1)),
                                  (4, dict(num_groups=4, quorum_kind=2, grid_rows=2, grid_cols=2))])
def test_phase1b_scan_matches_oracle(fa, oracle, R, kw):
    S = 2048
    gpu, ref = both(fa, oracle, num_slots=S, num_replicas=R, tally_ways=8, **kw)
    ng = kw.get("num_groups", 1)
    script = W.adversarial_script(S // 2, R, R // 2 + 1, 31 + R, epochs=16, fused=True, ngroups=ng)
    W.assert_same_outputs(W.run_script(gpu, script), W.run_script(ref, script))
    rng = np.random.default_rng(R)
    for wm in (0, 17, S // 4, S // 2 - 1, S // 2 + 5):
        for _ in range(3):
            q = W.bits_from_bool(W.random_subsets(rng, ng, R, 1, R))
            for cap in (S, 100):
                a = gpu.leader_phase1b_scan(wm, q, cap)
                b = ref.leader_phase1b_scan(wm, q, cap)
                assert a[0] == b[0] == 0 and a[1] == b[1]
                np.testing.assert_array_equal(a[2], b[2])
                np.testing.assert_array_equal(a[3], b[3])


@pytest.mark.gpu
def test_proxy_forget_gc(fa, oracle):
    """fpx_proxy_forget: the proxy leader forgets the tallies of a slot window (extension; the
    reference's ProxyLeader.states grows forever) so that more rounds than tally_ways can follow."""
    import ctypes as C

    S = 256
    gpu, ref = both(fa, oracle, num_slots=S, num_replicas=3, f=1, tally_ways=2)
    L = oracle.lib()
    L.fpo_proxy_forget.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    slot = np.arange(S, dtype=np.int32)
    val = W.steady_values(slot)
    for rnd in range(7):
        if rnd and rnd % 2 == 0:  # both ways of every slot are taken: forget slots [64, 192)
            gpu.proxy_forget(64, 128)
            assert L.fpo_proxy_forget(ref._h, 64, 128) == 0
        rr = np.full(S, rnd, np.int32)
        a = gpu.phase2_fused(slot, rr, val)
        b = ref.phase2_fused(slot, rr, val)
        if rnd < 2:
            assert a[0] == b[0] == 0
            for x, y in zip(a[1:], b[1:]):
                np.testing.assert_array_equal(x, y)
        else:
            # slots outside the forgotten window have no free way left: the device reports it,
            # the forgotten window keeps working and matches the oracle
            assert a[0] == fa.FPX_ECAPACITY
            np.testing.assert_array_equal(a[1][64:192], b[1][64:192])
            np.testing.assert_array_equal(a[3][64:192], b[3][64:192])
            assert a[1][64:192].all() and not a[1][:64].any()
    assert gpu.read_tally(100) == ref.read_tally(100)
    with pytest.raises(fa.FpxError):
        gpu.proxy_forget(0, S + 1)


# ---- the acceptor's half of Phase 1: Phase1b.info, and recycling rows of the window ------------------------------
def test_oracle_phase1b_info_trace(oracle):
    """multipaxos/Acceptor.scala:163-181, by hand: states.iteratorFrom(chosenWatermark), ascending slots, only the
    slots the acceptor voted in, overwritten votes report the LAST vote (states(slot) = State(round, value), :205-208)"""
    s = oracle.System(oracle.make_config(num_slots=16, num_replicas=3, f=1))
    s.acceptor_handle_phase2a(0, 1, 4, 2, 300)
    s.acceptor_handle_phase2a(0, 1, 1, 2, 200)
    s.acceptor_handle_phase2a(0, 1, 9, 3, 900)
    s.acceptor_handle_phase2a(0, 1, 4, 3, 301)      # equal-or-higher round: the vote in slot 4 is overwritten
    s.acceptor_handle_phase2a(0, 0, 2, 0, 50)       # another acceptor
    sl, vr, vv = s.acceptor_phase1b_info(0, 1, 0)
    assert (sl.tolist(), vr.tolist(), vv.tolist()) == ([1, 4, 9], [2, 3, 3], [200, 301, 900])
    sl, vr, vv = s.acceptor_phase1b_info(0, 1, 2)
    assert (sl.tolist(), vr.tolist(), vv.tolist()) == ([4, 9], [3, 3], [301, 900])
    assert len(s.acceptor_phase1b_info(0, 1, 10)[0]) == 0 and len(s.acceptor_phase1b_info(0, 2, 0)[0]) == 0
    assert s.acceptor_phase1b_info(0, 0, 0)[0].tolist() == [2]
    # recycled rows: the votes are gone, the round is not (a stale Phase2a still is refused)
    s.recycle_slots(0, 8)
    assert s.acceptor_phase1b_info(0, 1, 0)[0].tolist() == [9]
    assert s.read_acceptor(0, 1)[0] == 3


@pytest.mark.gpu
@pytest.mark.parametrize("R,kw", [(3, dict(f=1)), (5, dict(quorum_kind=1, ballot_mode=1)), (256, dict(f=127)),
                                  (4, dict(num_groups=4, quorum_kind=2, grid_rows=2, grid_cols=2)),
                                  (3, dict(f=1, num_groups=2, num_leader_groups=4))])
def test_phase1b_info_and_recycle_match_oracle(fa, oracle, row_layout, R, kw):
    S = 2048
    gpu, ref = both(fa, oracle, num_slots=S, num_replicas=R, tally_ways=8, **kw)
    ng = kw.get("num_groups", 1) * kw.get("num_leader_groups", 1)
    script = W.adversarial_script(S // 2, R, R // 2 + 1, 77 + R, epochs=16, fused=True, ngroups=kw.get("num_groups", 1)) \
        if "num_leader_groups" not in kw else None
    if script is not None:
        W.assert_same_outputs(W.run_script(gpu, script), W.run_script(ref, script))
    else:  # Mencius geometry: a plain stream in two rounds with thrifty targets
        rng0 = np.random.default_rng(5)
        for rnd in (0, 1):
            slot = rng0.permutation(S)[: S // 2].astype(np.int32)
            tgt = W.bits_from_bool(W.random_subsets(rng0, len(slot), R, 1, R))
            a = gpu.phase2_fused(slot, np.full(len(slot), rnd, np.int32), W.steady_values(slot), tgt)
            b = ref.phase2_fused(slot, np.full(len(slot), rnd, np.int32), W.steady_values(slot), tgt)
            assert a[0] == b[0] == 0
    rng = np.random.default_rng(R)

    def compare():
        for g in range(ng):
            for r in sorted(set([0, R - 1] + rng.integers(0, R, 3).tolist())):
                for wm in (0, 17, S // 4, S // 2 - 1, S - 1):
                    a, b = gpu.acceptor_phase1b_info(g, r, wm), ref.acceptor_phase1b_info(g, r, wm)
                    for x, y in zip(a, b):
                        np.testing.assert_array_equal(x, y)
    compare()
    assert sum(len(ref.acceptor_phase1b_info(g, 0, 0)[0]) for g in range(ng)) > 0
    # cap smaller than the count: the count is the total, the first cap entries are written
    import ctypes as C
    k = C.c_int32()
    sl, vr, vv = (np.full(4, -7, np.int32) for _ in range(3))
    want = ref.acceptor_phase1b_info(0, 0, 0)
    assert gpu.L.fpx_acceptor_phase1b_info(gpu._h, 0, 0, 0, 4, C.byref(k), sl.ctypes.data, vr.ctypes.data, vv.ctypes.data) == 0
    assert k.value == len(want[0]) and sl.tolist()[: min(4, k.value)] == want[0][:4].tolist()
    assert gpu.L.fpx_acceptor_phase1b_info(gpu._h, ng, 0, 0, 0, C.byref(k), None, None, None) == fa.FPX_EINVAL
    # recycle the middle of the window on both, then the same stream continues on it in a higher round
    gpu.recycle_slots(S // 8, S // 4)
    ref.recycle_slots(S // 8, S // 4)
    compare()
    W.assert_same_state(gpu, ref)
    slot = np.arange(S // 8, S // 8 + S // 4, dtype=np.int32)
    hi = np.full(len(slot), 1000, np.int32)   # above every round of the stream
    a, b = gpu.phase2_fused(slot, hi, W.steady_values(slot)), ref.phase2_fused(slot, hi, W.steady_values(slot))
    assert a[0] == b[0] == 0
    for x, y in zip(a[1:], b[1:]):
        np.testing.assert_array_equal(x, y)
    assert a[1].all()            # every recycled row chooses again: the tallies were forgotten with the votes
    compare()
    W.assert_same_state(gpu, ref)
    with pytest.raises(fa.FpxError):
        gpu.recycle_slots(S - 1, 2)
