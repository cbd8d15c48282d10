"""GPU parity: the HIP path (through the C ABI of include/fpx.h) against the CPU oracle on the same
seeded inputs -- bit-exact on every output and on the whole acceptor / proxy-leader state.

Run on the MI355X box: python -m pytest tests -m gpu
"""
import numpy as np
import pytest

from tests import workloads as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import frankenpaxos_amd

    frankenpaxos_amd.lib()  # raises if libfpx.so is missing: no fallback
    return frankenpaxos_amd


def both(fa, oracle, **kw):
    gpu = fa.Context(fa.make_config(**kw))
    ref = oracle.System(oracle.make_config(**kw))
    return gpu, ref


def check(gpu, ref, script, tally_slots=()):
    W.assert_same_outputs(W.run_script(gpu, script), W.run_script(ref, script))
    W.assert_same_state(gpu, ref, tally_slots)


# ---------------------------------------------------------------------------------------------------
# BASELINE.json configs[1]: MultiPaxos f=1, 64k slots x 3 acceptors, bit-exact bring-up
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ballot_mode", [0, 1])
def test_config2_64k_slots_3_acceptors(fa, oracle, ballot_mode):
    S = 65536
    gpu, ref = both(fa, oracle, num_slots=S, num_replicas=3, f=1, ballot_mode=ballot_mode)
    slot, rnd, val = W.steady_stream(S)
    check(gpu, ref, [("fused", slot, rnd, val, None)], tally_slots=[0, 1, S - 1])
    st, ch, cr, cv, nr = gpu.phase2_fused(slot, rnd, val)  # same (slot, round) again: ignored
    assert st == 0 and not ch.any()
    assert (gpu.read_scalars()[1] == S - 1).all()


@pytest.mark.parametrize("ballot_mode", [0, 1])
def test_config2_unfused_pipeline(fa, oracle, ballot_mode):
    S = 4096
    gpu, ref = both(fa, oracle, num_slots=S, num_replicas=3, f=1, ballot_mode=ballot_mode)
    slot, rnd, val = W.steady_stream(S)
    dup = np.zeros(S, bool)
    dup[::7] = True
    check(gpu, ref, [("k1k2", slot, rnd, val, None, dup)], tally_slots=[0, 5, S - 1])


# ---------------------------------------------------------------------------------------------------
# the headline shape: R = 256, threshold f = 127 (q = 128)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ballot_mode", [0, 1])
def test_steady_256_replicas(fa, oracle, ballot_mode):
    S = 8192
    gpu, ref = both(fa, oracle, num_slots=S, num_replicas=256, f=127, ballot_mode=ballot_mode)
    slot, rnd, val = W.steady_stream(S)
    script = [("phase1a", 0, 0, 0, None), ("fused", slot, rnd, val, None)]
    check(gpu, ref, script, tally_slots=[0, 100, S - 1])
    out = W.run_script(gpu, [("fused", slot, rnd + 1, val, None)])
    assert out[0][2].all()  # re-proposal in round 1: every slot chosen again


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("ballot_mode", [0, 1])
@pytest.mark.parametrize("fused", [True, False])
def test_adversarial_256_replicas(fa, oracle, seed, ballot_mode, fused):
    S = 4096
    gpu, ref = both(fa, oracle, num_slots=S, num_replicas=256, f=127, ballot_mode=ballot_mode,
                    tally_ways=8)
    script = W.adversarial_script(S, 256, 128, seed, epochs=64, fused=fused)
    check(gpu, ref, script, tally_slots=range(0, S, 97))


# ---------------------------------------------------------------------------------------------------
# every lanes-per-slot instantiation: R from 1 to 256, odd sizes included
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R", [1, 2, 3, 4, 5, 7, 8, 12, 16, 17, 31, 32, 33, 64, 65, 100, 128, 129, 255])
@pytest.mark.parametrize("ballot_mode", [0, 1])
def test_all_replica_counts(fa, oracle, R, ballot_mode):
    S = 1024
    q = R // 2 + 1
    gpu, ref = both(fa, oracle, num_slots=S, num_replicas=R, quorum_kind=1, ballot_mode=ballot_mode,
                    tally_ways=8)
    check(gpu, ref, W.adversarial_script(S, R, q, 7 + R, epochs=16, fused=True),
          tally_slots=range(0, S, 61))
    gpu.reset()
    ref.reset()
    check(gpu, ref, W.adversarial_script(S, R, q, 11 + R, epochs=16, fused=False),
          tally_slots=range(0, S, 61))


# ---------------------------------------------------------------------------------------------------
# BASELINE.json configs[2]: 16 independent 2x2 grid quorums ; Mencius slot -> group map
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ballot_mode", [0, 1])
def test_config3_grid_2x2_16_groups(fa, oracle, ballot_mode):
    S = 16384
    kw = dict(num_slots=S, num_replicas=4, num_groups=16, quorum_kind=2, grid_rows=2, grid_cols=2,
              ballot_mode=ballot_mode, tally_ways=8)
    gpu, ref = both(fa, oracle, **kw)
    check(gpu, ref, W.adversarial_script(S, 4, 2, 5, epochs=32, fused=True, ngroups=16),
          tally_slots=range(0, S, 331))
    gpu.reset()
    ref.reset()
    check(gpu, ref, W.adversarial_script(S, 4, 2, 6, epochs=32, fused=False, ngroups=16),
          tally_slots=range(0, S, 331))


def test_config5_mencius_slot_map(fa, oracle, row_layout):
    """mencius: leader group = slot % L, acceptor group = (slot / L) % A (mencius/ProxyLeader.scala:231-234)"""
    S = 8192
    kw = dict(num_slots=S, num_replicas=3, num_groups=2, num_leader_groups=8, f=1, tally_ways=8)
    gpu, ref = both(fa, oracle, **kw)
    check(gpu, ref, W.adversarial_script(S, 3, 2, 9, epochs=32, fused=True, ngroups=16),
          tally_slots=range(0, S, 257))
    # leader groups move through rounds independently: one batch, different rounds per group
    gpu.reset()
    ref.reset()
    slot = np.arange(S, dtype=np.int32)
    rnd = (slot % 8).astype(np.int32)  # round = leader group index
    val = W.steady_values(slot)
    check(gpu, ref, [("fused", slot, rnd, val, None)], tally_slots=[0, 1, 9, S - 1])


def test_large_grid_16x16(fa, oracle):
    S = 2048
    kw = dict(num_slots=S, num_replicas=256, quorum_kind=2, grid_rows=16, grid_cols=16, tally_ways=8)
    gpu, ref = both(fa, oracle, **kw)
    # a 16x16 grid needs one acceptor per row: use target subsets around half the grid
    check(gpu, ref, W.adversarial_script(S, 256, 40, 13, epochs=16, fused=True),
          tally_slots=range(0, S, 101))
    gpu.reset()
    ref.reset()
    check(gpu, ref, W.adversarial_script(S, 256, 40, 14, epochs=16, fused=False),
          tally_slots=range(0, S, 101))


def test_unanimous_writes(fa, oracle):
    S = 1024
    gpu, ref = both(fa, oracle, num_slots=S, num_replicas=5, quorum_kind=3, tally_ways=8)
    check(gpu, ref, W.adversarial_script(S, 5, 5, 21, epochs=8, fused=True), tally_slots=range(0, S, 31))


# ---------------------------------------------------------------------------------------------------
# any batch through the host entry points: duplicate slots, rounds going up and down
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ballot_mode", [0, 1])
@pytest.mark.parametrize("R", [3, 256])
def test_arbitrary_batches_are_split_exactly(fa, oracle, ballot_mode, R):
    S = 512
    gpu, ref = both(fa, oracle, num_slots=S, num_replicas=R, quorum_kind=1, ballot_mode=ballot_mode,
                    tally_ways=8)
    rng = np.random.default_rng(123 + R)
    script = []
    for _ in range(6):
        n = 700
        slot = rng.integers(0, S, n).astype(np.int32)         # many duplicates
        rnd = rng.integers(0, 5, n).astype(np.int32)          # non-monotone rounds
        val = rng.integers(0, 1 << 30, n).astype(np.int32)
        tgt = W.bits_from_bool(W.random_subsets(rng, n, R, 1, R))
        script.append(("fused", slot, rnd, val, tgt))
        script.append(("phase2a", slot[:100], rnd[:100], val[:100], tgt[:100]))
    check(gpu, ref, script, tally_slots=range(0, S, 17))


@pytest.mark.parametrize("R,f", [(3, 1), (5, 2)])
def test_random_delivery_orders_message_at_a_time(fa, oracle, R, f):
    """the reference's randomized-schedule test style (tests/paxos_sim.py): the same random schedule
    of single-message deliveries, drops, duplicates and leader changes on GPU and oracle gives the
    same trace and never violates safety"""
    from tests import paxos_sim

    for seed in range(6):
        kw = dict(num_slots=6, num_replicas=R, f=f, tally_ways=8)
        gpu, ref = both(fa, oracle, **kw)
        tr_g, ch_g = paxos_sim.simulate(gpu, seed, R=R, f=f, S=6, steps=300)
        tr_r, ch_r = paxos_sim.simulate(ref, seed, R=R, f=f, S=6, steps=300)
        assert tr_g == tr_r and ch_g == ch_r
        W.assert_same_state(gpu, ref, tally_slots=range(6))


# ---------------------------------------------------------------------------------------------------
# error paths
# ---------------------------------------------------------------------------------------------------
def test_phase2b_unknown_slot_round_is_fatal(fa, oracle):
    gpu, ref = both(fa, oracle, num_slots=64, num_replicas=3, f=1)
    slot = np.array([5], np.int32)
    vb = W.bits_from_bool(np.array([[True, True, False]]))
    for be in (gpu, ref):
        st, ch, cr, cv = be.proxy_phase2b(slot, np.array([0], np.int32), vb)
        assert st == fa.FPX_EFATAL_UNKNOWN_SLOTROUND and not ch.any()
        assert be.error_detail() == (0, 5, 0)
        be.proxy_open(slot, np.array([2], np.int32), np.array([77], np.int32))
        st, *_ = be.proxy_phase2b(slot, np.array([1], np.int32), vb)  # round 1 never opened
        assert st == fa.FPX_EFATAL_UNKNOWN_SLOTROUND
        st, ch, cr, cv = be.proxy_phase2b(slot, np.array([2], np.int32), vb)
        assert st == 0 and ch[0] == 1 and cr[0] == 2 and cv[0] == 77
        st, ch, cr, cv = be.proxy_phase2b(slot, np.array([2], np.int32), vb)  # after Done: ignored
        assert st == 0 and ch[0] == 0 and cr[0] == -1 and cv[0] == -1


def test_invalid_arguments(fa):
    gpu = fa.Context(fa.make_config(num_slots=64, num_replicas=3, f=1))
    one = np.array([0], np.int32)
    assert gpu.phase2_fused(np.array([64], np.int32), one, one)[0] == fa.FPX_EINVAL
    assert gpu.phase2_fused(np.array([-1], np.int32), one, one)[0] == fa.FPX_EINVAL
    assert gpu.phase2_fused(one, np.array([-3], np.int32), one)[0] == fa.FPX_EINVAL
    assert gpu.error_detail() == (0, 0, -3)
    with pytest.raises(fa.FpxError):
        fa.Context(fa.make_config(num_slots=64, num_replicas=257))
    with pytest.raises(fa.FpxError):
        fa.Context(fa.make_config(num_slots=64, num_replicas=6, quorum_kind=2, grid_rows=2, grid_cols=2))


def test_tally_capacity(fa):
    gpu = fa.Context(fa.make_config(num_slots=8, num_replicas=3, f=1, tally_ways=2))
    s = np.array([3], np.int32)
    none = W.bits_from_bool(np.zeros((1, 3), bool))
    for r in range(2):
        assert gpu.phase2_fused(s, np.array([r], np.int32), s, none)[0] == 0
    assert gpu.phase2_fused(s, np.array([2], np.int32), s, none)[0] == fa.FPX_ECAPACITY


def test_dev_batches_violating_the_run_contract_apply_nothing(fa):
    import torch

    S = 256
    gpu = fa.Context(fa.make_config(num_slots=S, num_replicas=3, f=1))
    dev = torch.device("cuda:0")
    before = gpu.read_state()

    def run(slot, rnd):
        t = lambda a: torch.tensor(a, dtype=torch.int32, device=dev)
        ch = torch.zeros(len(slot), dtype=torch.uint8, device=dev)
        gpu.phase2_fused_dev(t(slot), t(rnd), t([1] * len(slot)), chosen=ch)
        return gpu.sync(), ch.cpu().numpy()

    st, ch = run([1, 2, 1], [0, 0, 0])  # duplicate slot
    assert st == fa.FPX_EORDER and not ch.any()
    st, ch = run([1, 2, 3], [0, 1, 0])  # two rounds for one acceptor group
    assert st == fa.FPX_EORDER
    after = gpu.read_state()
    for a, b in zip(before, after):
        np.testing.assert_array_equal(a, b)
    st, ch = run([1, 2, 3], [0, 0, 0])
    assert st == 0 and ch.all()


# ---------------------------------------------------------------------------------------------------
# device-pointer entry points == host entry points
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ballot_mode", [0, 1])
def test_dev_entry_points(fa, oracle, ballot_mode):
    import torch

    S, R = 4096, 256
    gpu, ref = both(fa, oracle, num_slots=S, num_replicas=R, f=127, ballot_mode=ballot_mode)
    dev = torch.device("cuda:0")
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    slot, rnd, val = W.steady_stream(S)
    rng = np.random.default_rng(5)
    tgt = W.bits_from_bool(W.random_subsets(rng, S, R, 100, R))
    t_slot, t_rnd, t_val = (torch.from_numpy(x).to(dev) for x in (slot, rnd, val))
    t_tgt = torch.from_numpy(tgt.view(np.int64)).to(dev)
    ch = torch.empty(S, dtype=torch.uint8, device=dev)
    cr = torch.empty(S, dtype=torch.int32, device=dev)
    cv = torch.empty(S, dtype=torch.int32, device=dev)
    nr = torch.empty(S, dtype=torch.int32, device=dev)
    gpu.phase2_fused_dev(t_slot, t_rnd, t_val, t_tgt, ch, cr, cv, nr)
    assert gpu.sync() == 0
    st, ch_r, cr_r, cv_r, nr_r = ref.phase2_fused(slot, rnd, val, tgt)
    np.testing.assert_array_equal(ch.cpu().numpy(), ch_r)
    np.testing.assert_array_equal(cr.cpu().numpy(), cr_r)
    np.testing.assert_array_equal(cv.cpu().numpy(), cv_r)
    np.testing.assert_array_equal(nr.cpu().numpy(), nr_r)
    W.assert_same_state(gpu, ref, tally_slots=range(0, S, 111))
    # K1 -> K2 on device, vote bitmaps stay in HBM
    gpu.reset()
    ref.reset()
    vb = torch.empty((S, 4), dtype=torch.int64, device=dev)
    nb = torch.empty((S, 4), dtype=torch.int64, device=dev)
    gpu.proxy_open_dev(t_slot, t_rnd, t_val)
    gpu.acceptor_phase2a_dev(t_slot, t_rnd, t_val, t_tgt, vb, nb, nr)
    gpu.proxy_phase2b_dev(t_slot, t_rnd, vb, ch, cr, cv)
    assert gpu.sync() == 0
    ref.proxy_open(slot, rnd, val)
    st, vb_r, nb_r, nr_r = ref.acceptor_phase2a(slot, rnd, val, tgt)
    st, ch_r, cr_r, cv_r = ref.proxy_phase2b(slot, rnd, vb_r)
    np.testing.assert_array_equal(vb.cpu().numpy().view(np.uint64), vb_r)
    np.testing.assert_array_equal(nb.cpu().numpy().view(np.uint64), nb_r)
    np.testing.assert_array_equal(ch.cpu().numpy(), ch_r)
    np.testing.assert_array_equal(cv.cpu().numpy(), cv_r)
    W.assert_same_state(gpu, ref, tally_slots=range(0, S, 111))
    gpu.set_stream(None)


# ---------------------------------------------------------------------------------------------------
# replica-axis sharding: two contexts own 128 acceptors each; the OR (== sum, the
# bit ranges are disjoint) of their partial bitmaps feeds one tally
# ---------------------------------------------------------------------------------------------------
def test_replica_axis_sharding(fa, oracle):
    S, R = 2048, 256
    whole = oracle.System(oracle.make_config(num_slots=S, num_replicas=R, f=127, tally_ways=8))
    shards = [fa.Context(fa.make_config(num_slots=S, num_replicas=128, f=127, replica_base=b,
                                        replicas_total=R, tally_ways=8)) for b in (0, 128)]
    rng = np.random.default_rng(77)
    slot, rnd, val = W.steady_stream(S)
    tgt = W.bits_from_bool(W.random_subsets(rng, S, R, 120, 140))
    whole.proxy_open(slot, rnd, val)
    st, vb_ref, nb_ref, nr_ref = whole.acceptor_phase2a(slot, rnd, val, tgt)
    st, ch_ref, cr_ref, cv_ref = whole.proxy_phase2b(slot, rnd, vb_ref)
    parts = [sh.acceptor_phase2a(slot, rnd, val, tgt)[1] for sh in shards]
    assert not (parts[0] & parts[1]).any()
    summed = parts[0] + parts[1]  # what an RCCL all-reduce(sum) produces
    np.testing.assert_array_equal(summed, vb_ref)
    shards[0].proxy_open(slot, rnd, val)
    st, ch, cr, cv = shards[0].proxy_phase2b(slot, rnd, summed)
    np.testing.assert_array_equal(ch, ch_ref)
    np.testing.assert_array_equal(cv, cv_ref)
    assert 0 < ch.sum() < S


# ---------------------------------------------------------------------------------------------------
# a5 on the device: the reference's known-answer tests through fpx_quorum_eval
# ---------------------------------------------------------------------------------------------------
def _sets(fa, cfg, sets, strict=True, read=False):
    nodes = np.stack([W.bits_from_bool(np.isin(np.arange(256), list(s))[None, :])[0] for s in sets])
    return list(fa.quorum_eval(cfg, nodes, strict=strict, read=read))


def test_device_grid_quorums(fa):
    """quorums/GridTest.scala:11-102 with nodes 1..6 -> bits 0..5, 9001 -> bit 200"""
    from tests.test_oracle_golden import GRID_WRITE_CASES

    cfg = fa.make_config(num_slots=1, num_replicas=6, quorum_kind=2, grid_rows=2, grid_cols=3)
    b = lambda xs: [x - 1 if x != 9001 else 200 for x in xs]
    cases = [c for c in GRID_WRITE_CASES] + [([i], False) for i in range(1, 7)]
    assert _sets(fa, cfg, [b(xs) for xs, _ in cases]) == [w for _, w in cases]
    assert _sets(fa, cfg, [b(xs) for xs, _ in cases], strict=False) == [w for _, w in cases]
    sup = [([9001] + xs, w) for xs, w in GRID_WRITE_CASES if xs]
    assert _sets(fa, cfg, [b(xs) for xs, _ in sup], strict=False) == [w for _, w in sup]
    with pytest.raises(ValueError):
        _sets(fa, cfg, [b([9001, 1, 4])], strict=True)
    reads = [([1, 2, 4], False), ([4, 5, 3], False), ([1, 2, 3], True), ([4, 5, 6], True),
             ([1, 2, 3, 4], True), ([1, 2, 3, 4, 5, 6], True), ([], False), ([1, 2], False)]
    assert _sets(fa, cfg, [b(xs) for xs, _ in reads], read=True) == [w for _, w in reads]
    assert _sets(fa, cfg, [b([9001] + xs) for xs, _ in reads], strict=False, read=True) == [w for _, w in reads]


def test_device_majority_and_unanimous_quorums(fa):
    """quorums/SimpleMajorityTest.scala:11-63, quorums/UnanimousWrites.scala:11-75 (node 5 -> bit 5, foreign)"""
    maj = fa.make_config(num_slots=1, num_replicas=5, quorum_kind=1)
    cases = [([], False), ([0], False), ([0, 1], False), ([0, 1, 2], True), ([0, 1, 2, 3], True),
             ([0, 1, 2, 3, 4], True)]
    for read in (False, True):
        assert _sets(fa, maj, [xs for xs, _ in cases], read=read) == [w for _, w in cases]
        assert _sets(fa, maj, [xs + [5] for xs, _ in cases], strict=False, read=read) == [w for _, w in cases]
    una = fa.make_config(num_slots=1, num_replicas=5, quorum_kind=3)
    wcases = [([], False), ([0], False), ([0, 1], False), ([0, 1, 2], False), ([0, 1, 2, 3], False),
              ([0, 1, 2, 3, 4], True)]
    assert _sets(fa, una, [xs for xs, _ in wcases]) == [w for _, w in wcases]
    assert _sets(fa, una, [xs + [5] for xs, _ in wcases], strict=False) == [w for _, w in wcases]
    rcases = [([], False), ([0], True), ([3], True), ([0, 1], True), ([0, 1, 2, 3, 4], True)]
    assert _sets(fa, una, [xs for xs, _ in rcases], read=True) == [w for _, w in rcases]
    assert _sets(fa, una, [[5]], strict=False, read=True) == [False]
    with pytest.raises(ValueError):
        _sets(fa, maj, [[0, 1, 5]], strict=True)


def test_device_quorum_predicates_match_oracle_exhaustively(fa, oracle):
    """every subset of small systems + random subsets of big ones, device == oracle"""
    rng = np.random.default_rng(3)
    systems = [dict(num_replicas=n, quorum_kind=1) for n in range(1, 10)]
    systems += [dict(num_replicas=n, quorum_kind=3) for n in range(1, 10)]
    systems += [dict(num_replicas=r * c, quorum_kind=2, grid_rows=r, grid_cols=c)
                for r in range(2, 6) for c in range(2, 6)]
    systems += [dict(num_replicas=n, quorum_kind=0, f=f) for n, f in ((3, 1), (5, 2), (255, 127), (256, 127))]
    systems += [dict(num_replicas=256, quorum_kind=2, grid_rows=16, grid_cols=16),
                dict(num_replicas=252, quorum_kind=2, grid_rows=4, grid_cols=63)]
    for kw in systems:
        n = kw["num_replicas"]
        if n <= 10:
            mat = ((np.arange(1 << n)[:, None] >> np.arange(n)[None, :]) & 1).astype(bool)
        else:
            mat = W.random_subsets(rng, 512, n, 0, n)
        nodes = W.bits_from_bool(mat)
        cfg = fa.make_config(num_slots=1, **kw)
        ocfg = oracle.make_config(num_slots=1, **kw)
        ref = oracle.System(ocfg)
        for read in (False, True):
            got = fa.quorum_eval(cfg, nodes, strict=True, read=read)
            want = [ref.is_read_quorum(x) if read else ref.is_write_quorum(x) for x in nodes]
            assert list(got) == want, (kw, read)


def test_round_system_through_the_abi(fa):
    """roundsystem/RoundSystemTest.scala:13-62"""
    from tests.test_oracle_golden import NEXT_CLASSIC_ROUND

    assert [fa.round_leader(3, r) for r in range(9)] == [0, 1, 2, 0, 1, 2, 0, 1, 2]
    for leader, wants in NEXT_CLASSIC_ROUND.items():
        assert [fa.next_classic_round(3, leader, r) for r in range(-1, 7)] == wants


# ---------------------------------------------------------------------------------------------------
# BASELINE.json full size (2^20 slots x 256 acceptors): size-independent properties
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ballot_mode", [0, 1])
def test_full_size_grid_properties(fa, ballot_mode):
    import torch

    S, R = 1 << 20, 256
    gpu = fa.Context(fa.make_config(num_slots=S, num_replicas=R, f=127, ballot_mode=ballot_mode,
                                    flags=fa.FPX_F_TRUSTED))
    dev = torch.device("cuda:0")
    slot, rnd, val = W.steady_stream(S)
    t_slot, t_rnd, t_val = (torch.from_numpy(x).to(dev) for x in (slot, rnd, val))
    ch = torch.empty(S, dtype=torch.uint8, device=dev)
    cr = torch.empty(S, dtype=torch.int32, device=dev)
    cv = torch.empty(S, dtype=torch.int32, device=dev)
    assert gpu.acceptor_phase1a(0, 0)[0] == 0
    gpu.phase2_fused_dev(t_slot, t_rnd, t_val, None, ch, cr, cv)
    assert gpu.sync() == 0
    # every slot chosen in round 0 with its proposed value (SURVEY.md 8d "Expected")
    assert bool(ch.all()) and bool((cr == 0).all()) and bool((cv == t_val).all())
    # idempotence: the same Phase2a's again are ignored by the proxy leader
    gpu.phase2_fused_dev(t_slot, t_rnd, t_val, None, ch, cr, cv)
    assert gpu.sync() == 0 and not bool(ch.any())
    # the acceptors' logs: every cell voted (round 0, value of its slot): checksum of checksums
    vr, vv, bl = gpu.read_state()
    assert (vr == 0).all()
    assert (vv == val[:, None]).all()
    pr, mv = gpu.read_scalars()
    assert (mv == S - 1).all() and (pr == (0 if ballot_mode == 0 else -1)).all()
    # a minority of acceptors ahead of the leader: 129 Nack => nothing is chosen, nack_round = 5
    ahead = W.bits_from_bool((np.arange(R) < 129)[None, :])[0]
    assert gpu.acceptor_phase1a(0, 5, 0, ahead)[0] == 0
    nr = torch.empty(S, dtype=torch.int32, device=dev)
    gpu.phase2_fused_dev(t_slot, t_rnd + 1, t_val, None, ch, cr, cv, nr)
    assert gpu.sync() == 0 and not bool(ch.any()) and bool((nr == 5).all())
    # ... and 128 of 256 promised ahead leaves exactly a quorum of 128 voters => chosen in round 2
    gpu2 = fa.Context(fa.make_config(num_slots=4096, num_replicas=R, f=127, ballot_mode=ballot_mode))
    ahead = W.bits_from_bool((np.arange(R) < 128)[None, :])[0]
    gpu2.acceptor_phase1a(0, 5, 0, ahead)
    st, ch2, cr2, cv2, nr2 = gpu2.phase2_fused(slot[:4096], rnd[:4096] + 2, val[:4096])
    assert st == 0 and ch2.all() and (cr2 == 2).all() and (nr2 == 5).all()


def test_dev_inputs_produced_on_the_torch_stream_are_ordered(fa):
    """Regression (r01): torch.cuda.current_stream().cuda_stream is 0 (the default stream) unless the
    caller switched streams; fpx_set_stream must take it as the default stream, not as "use the private
    stream" -- otherwise the _dev kernels race with the torch kernels that produce their inputs."""
    import torch

    S, R = 1 << 20, 256
    dev = torch.device("cuda:0")
    gpu = fa.Context(fa.make_config(num_slots=S, num_replicas=R, f=127, flags=fa.FPX_F_TRUSTED))
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    for rep in range(3):
        # a long chain of torch kernels whose LAST result is the slot array; no synchronisation
        x = torch.rand(1 << 24, device=dev)
        perm = torch.argsort(x)[:S] % S
        slot = torch.unique(perm.to(torch.int32))            # distinct slots, produced just in time
        rnd = torch.full_like(slot, rep)
        val = slot * 3 + 1
        ch = torch.zeros(slot.numel(), dtype=torch.uint8, device=dev)
        cv = torch.zeros(slot.numel(), dtype=torch.int32, device=dev)
        gpu.phase2_fused_dev(slot, rnd, val, None, ch, None, cv)
        assert gpu.sync() == 0
        assert bool(ch.all()) and bool((cv == val).all())
    gpu.set_stream(None)


@pytest.mark.gpu
def test_page_locked_host_batches(fa, oracle):
    """fpx_host_alloc: the host-pointer entry points take page-locked buffers (the DMA path) with the
    same results as pageable numpy arrays"""
    kw = dict(num_slots=4096, num_replicas=7, f=3, num_groups=2)
    gpu, ref = fa.Context(fa.make_config(**kw)), oracle.System(oracle.make_config(**kw))
    n = 3000
    rng = np.random.default_rng(11)
    pins = [fa.PinnedArray((n,), np.int32) for _ in range(3)]
    slot, rnd, val = (p.array for p in pins)
    slot[:] = rng.permutation(4096)[:n]
    rnd[:] = 0
    val[:] = rng.integers(0, 1 << 30, n)
    tm = fa.PinnedArray((n, 4), np.uint64)
    tm.array[:] = 0
    tm.array[:, 0] = rng.integers(1, 128, n).astype(np.uint64)
    a = gpu.phase2_fused(slot, rnd, val, tm.array)
    b = ref.phase2_fused(slot.copy(), rnd.copy(), val.copy(), tm.array.copy())
    assert a[0] == b[0] == 0
    for x, y in zip(a[1:], b[1:]):
        np.testing.assert_array_equal(x, y)
    W.assert_same_state(gpu, ref)
    for p in pins + [tm]:
        p.free()
    assert fa.lib().fpx_host_free(None) == fa.FPX_EINVAL


@pytest.mark.gpu
@pytest.mark.parametrize("R", [3, 7, 33, 255, 256])
@pytest.mark.parametrize("ballot_mode", [0, 1])
def test_scattered_targets_hint_changes_nothing(fa, oracle, R, ballot_mode):
    """FPX_F_SCATTERED_TARGETS (16-byte read-modify-write of partially voted cells) is a pure
    performance hint: same votes, same tallies, same state as the oracle on the adversarial stream
    whose target masks are random subsets -- fused and unfused"""
    S = 2048
    q = R // 2 + 1
    kw = dict(num_slots=S, num_replicas=R, quorum_kind=1, ballot_mode=ballot_mode, tally_ways=8)
    gpu = fa.Context(fa.make_config(flags=fa.FPX_F_SCATTERED_TARGETS, **kw))
    ref = oracle.System(oracle.make_config(**kw))
    check(gpu, ref, W.adversarial_script(S, R, q, 23 + R, epochs=16, fused=True), tally_slots=range(0, S, 61))
    gpu.reset()
    ref.reset()
    check(gpu, ref, W.adversarial_script(S, R, q, 29 + R, epochs=16, fused=False), tally_slots=range(0, S, 61))


@pytest.mark.parametrize("R,quorum_kind", [(256, 0), (255, 0), (253, 1), (256, 3)])
@pytest.mark.parametrize("ballot_mode", [0, 1])
@pytest.mark.parametrize("fused", [True, False])
def test_thrifty_runs_take_the_packed_walk(fa, oracle, R, quorum_kind, ballot_mode, fused, monkeypatch):
    """Phase2as sent to RUNS of neighbouring acceptors (W.run_subsets: rotating sector-aligned windows, 100 .. 128 wide,
    some short of the quorum) on 256-cell rows: k_runs_check + the packed walk of k_phase2 (two rows per wavefront step,
    the vote set taken from the target set unless somebody Nacks).  The adversarial stream brings pre-promised
    acceptors (Nacks inside packed steps), re-proposals of voted slots (those launches fall back to the row-at-a-time
    walk on the device's own verdict), duplicated votes; every output and the whole state == the oracle, for the
    threshold, majority and unanimous predicates, with and without the padding positions 253 .. 255; and the same
    stream with the packed walk switched off (FPX_NO_PACKED_RUNS) gives the same again."""
    S = 2048
    f = 127 if quorum_kind == 0 else 0
    q = {0: 128, 1: R // 2 + 1, 3: R}[quorum_kind]
    kw = dict(num_slots=S, num_replicas=R, quorum_kind=quorum_kind, f=f, ballot_mode=ballot_mode, tally_ways=8)
    for off in (False, True):
        if off:
            monkeypatch.setenv("FPX_NO_PACKED_RUNS", "1")
        gpu, ref = both(fa, oracle, **kw)
        script = W.adversarial_script(S - 700, R, min(q, 108), 41 + R + quorum_kind, epochs=16, fused=fused, subsets=W.run_subsets)
        check(gpu, ref, script, tally_slots=range(0, S, 61))
        # odd chunk tails, a single message, and a batch one of whose masks is no run
        rng = np.random.default_rng(R)
        base = S - 700
        for n in (1, 2, 31, 33, 257):
            slot = np.arange(base, base + n, dtype=np.int32)
            base += n
            tgt = W.bits_from_bool(W.run_subsets(rng, n, R, 120, 128))
            if n == 33:
                tgt[7] = W.bits_from_bool(W.random_subsets(rng, 1, R, 100, 100))[0]
            ops = [("fused", slot, np.full(n, 900, np.int32), W.steady_values(slot), tgt)] if fused else \
                  [("k1k2", slot, np.full(n, 900, np.int32), W.steady_values(slot), tgt, np.zeros(n, bool))]
            W.assert_same_outputs(W.run_script(gpu, ops), W.run_script(ref, ops))
        W.assert_same_state(gpu, ref, tally_slots=range(S - 700, S, 7))
        gpu.close()


def test_capacity_error_applies_everything_else(fa, oracle):
    """a fused launch that hits FPX_ECAPACITY on some messages must still apply the others in
    full -- votes AND the acceptors' round / maxVotedSlot (a vote in round r with promised < r would let a
    later lower round be accepted).  The oracle has no capacity limit, so the expected state is the oracle's
    after the same batch minus the over-capacity messages."""
    import torch

    S, R = 4096, 256
    kw = dict(num_slots=S, num_replicas=R, f=127, tally_ways=2)
    gpu, ref = both(fa, oracle, **kw)
    slot, rnd, val = W.steady_stream(S)
    none = np.zeros((S, 4), np.uint64)          # delivered to nobody: the tallies stay Pending
    for r in (0, 1):
        assert gpu.phase2_fused(slot, rnd + r, val, none)[0] == 0
        assert ref.phase2_fused(slot, rnd + r, val, none)[0] == 0
    gpu.proxy_forget(0, S // 2)                 # the lower half of the window has free ways again
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    ch = torch.zeros(S, dtype=torch.uint8, device=dev)
    cv = torch.zeros(S, dtype=torch.int32, device=dev)
    gpu.phase2_fused_dev(t(slot), t(rnd + 2), t(val), None, ch, None, cv)
    assert gpu.sync() == fa.FPX_ECAPACITY
    i, s, r = gpu.error_detail()
    assert s >= S // 2 and r == 2
    half = S // 2
    st, ch_r, cr_r, cv_r, nr_r = ref.phase2_fused(slot[:half], rnd[:half] + 2, val[:half])
    assert st == 0 and ch_r.all()
    np.testing.assert_array_equal(ch.cpu().numpy()[:half], ch_r)
    assert not ch.cpu().numpy()[half:].any()
    W.assert_same_state(gpu, ref)               # cells, rounds (= 2 everywhere), maxVotedSlot (= S/2 - 1)
    assert (gpu.read_scalars()[0] == 2).all() and (gpu.read_scalars()[1] == half - 1).all()
    # a later lower round is Nacked, as it must be
    st, chg, crg, cvg, nrg = gpu.phase2_fused(slot[:8] , rnd[:8] + 1, val[:8])
    assert st == 0 and not chg.any() and (nrg == 2).all()


@pytest.mark.parametrize("fused", [True, False])
def test_eight_thousand_acceptors_in_one_context(fa, oracle, fused):
    """num_groups * num_leader_groups * R up to 8192 puts 64+ KiB of maxima tables in LDS, above
    the default dynamic-LDS limit: K1 and K3 opt in per kernel.  Mencius-shaped: 128 leader groups x 8
    acceptor groups x 8 acceptors."""
    S = 16384
    kw = dict(num_slots=S, num_replicas=8, num_groups=8, num_leader_groups=128, f=3, tally_ways=8)
    gpu, ref = both(fa, oracle, **kw)
    check(gpu, ref, W.adversarial_script(S, 8, 4, 41, epochs=4, fused=fused, ngroups=1024,
                                         subsets=W.fast_subsets), tally_slots=range(0, S, 1021))


def test_two_contexts_keep_their_device(fa):
    """entry points run on the context's device whatever the caller's current device is, and
    restore the caller's.  (One GPU on this box: the guard must at least be transparent.)"""
    import torch

    gpu = fa.Context(fa.make_config(num_slots=64, num_replicas=3, f=1))
    before = torch.cuda.current_device()
    st, ch, *_ = gpu.phase2_fused(np.arange(8, dtype=np.int32), np.zeros(8, np.int32), np.arange(8, dtype=np.int32))
    assert st == 0 and ch.all() and torch.cuda.current_device() == before
    if torch.cuda.device_count() > 1:
        other = fa.Context(fa.make_config(num_slots=64, num_replicas=3, f=1, device=1))
        st, ch, *_ = other.phase2_fused(np.arange(8, dtype=np.int32), np.zeros(8, np.int32), np.arange(8, dtype=np.int32))
        assert st == 0 and ch.all() and torch.cuda.current_device() == before


@pytest.mark.parametrize("R,ngroups", [(256, 1), (7, 1), (5, 3), (33, 2)])
def test_lazy_phase1a_promises_equal_the_sweep(fa, oracle, R, ngroups):
    """PER_SLOT Phase1a is O(R) on the device (one lazy record per acceptor) where the oracle rewrites every cell
    from the watermark on.  Random sequences of Phase1a's -- rising and stale rounds, rising and falling
    watermarks, partial target sets, several groups -- interleaved with votes that must see the promises
    (Nacks, nack rounds), checked on every output, and on the whole state before and after an explicit flush."""
    S = 2048
    q = R // 2 + 1
    kw = dict(num_slots=S, num_replicas=R, num_groups=ngroups, quorum_kind=1, ballot_mode=1, tally_ways=8)
    gpu, ref = both(fa, oracle, **kw)
    rng = np.random.default_rng(R * 10 + ngroups)
    rounds = [0] * ngroups
    slots_done = 0
    for step in range(60):
        k = rng.integers(0, 10)
        g = int(rng.integers(0, ngroups))
        if k < 4:
            # Phase1a: mostly ahead of everything (the lazy path), sometimes stale (the checked sweep)
            rnd = rounds[g] + int(rng.integers(1, 4)) if rng.random() < 0.7 else max(0, rounds[g] - int(rng.integers(0, 3)))
            wm = int(rng.integers(0, S)) if rng.random() < 0.7 else 0
            tgt = None if rng.random() < 0.4 else W.bits_from_bool(W.random_subsets(rng, 1, R, 1, R))[0]
            a, b = gpu.acceptor_phase1a(g, rnd, wm, tgt), ref.acceptor_phase1a(g, rnd, wm, tgt)
            assert a[0] == b[0] == 0
            np.testing.assert_array_equal(a[1], b[1], err_msg="promised bits, step %d" % step)
            np.testing.assert_array_equal(a[2], b[2], err_msg="nack bits, step %d" % step)
            if tgt is None and not b[2].any():
                rounds[g] = max(rounds[g], rnd)
        elif k < 9:
            n = int(rng.integers(1, 300))
            slot = np.sort(rng.choice(S, size=n, replace=False)).astype(np.int32)
            grp = slot % ngroups
            rr = np.array([max(0, rounds[int(x)] - (1 if rng.random() < 0.2 else 0)) for x in grp], np.int32)
            if ngroups == 1:
                rr[:] = rr[0]
            val = rng.integers(0, 1 << 30, n).astype(np.int32)
            tgt = None if rng.random() < 0.5 else W.bits_from_bool(W.random_subsets(rng, n, R, 1, R))
            op = [("fused", slot, rr, val, tgt)] if rng.random() < 0.6 else [("phase2a", slot, rr, val, tgt)]
            W.assert_same_outputs(W.run_script(gpu, op), W.run_script(ref, op))
        else:
            if rng.random() < 0.5:
                gpu.flush_promises()
            W.assert_same_state(gpu, ref)
    W.assert_same_state(gpu, ref, tally_slots=range(0, S, 97))
    np.testing.assert_array_equal(gpu.state_digest(), ref.state_digest())


@pytest.mark.parametrize("split", [False, True])
def test_phase1a_dev_is_asynchronous_and_equal(fa, oracle, monkeypatch, split):
    """Phase1a's on device-resident arguments between fused steps, nothing synchronised until the end: a fresh round for
    everybody, for 30 % of the acceptors, a STALE one (every cell of the group is ahead: the sweep answers, cell by cell),
    the round again, two that move the watermark past an older lazy promise (which the sweep first writes into the cells
    it alone covers), and a stale one behind a step of PARTIAL votes (target masks) in a higher round.  With a ballot per
    cell a Phase1a is ONE launch that does not wait for the fold of the step before it (k_p1a_fast: it bounds what that
    fold will raise by what the step left in part_all; split: FPX_P1A_SPLIT=1, the launches of rounds 2 - 5); the reply
    bits land where the caller wants them."""
    import torch

    if split:
        monkeypatch.setenv("FPX_P1A_SPLIT", "1")
    else:
        monkeypatch.delenv("FPX_P1A_SPLIT", raising=False)
    S, R = 4096, 256
    dev = torch.device("cuda:0")
    for ballot_mode in (0, 1):
        gpu, ref = both(fa, oracle, num_slots=S, num_replicas=R, f=127, ballot_mode=ballot_mode, tally_ways=8)
        gpu.set_stream(torch.cuda.current_stream().cuda_stream)
        rng = np.random.default_rng(3)
        slot, rnd, val = W.steady_stream(S)
        t = lambda a: torch.from_numpy(a).to(dev)
        outs = []
        # (Phase1a round, fraction of the acceptors it goes to, watermark, round of the step behind it, that step's targets)
        plan = ((0, None, 0, 0, None), (3, 0.3, 0, 3, None), (2, None, 0, 2, None), (3, None, 0, 3, None),
                (5, None, S // 2, 5, None), (4, 0.5, S // 4, 5, None), (6, 0.5, 3 * S // 4, 9, (100, 200)),
                (8, None, 0, 8, (1, 256)), (10, 0.7, 0, 10, None), (9, None, 0, 11, (120, 136)), (10, None, 0, 10, None))
        for k, (rr, frac, wm, vr, tsub) in enumerate(plan):
            tgt = None if frac is None else W.bits_from_bool(rng.random((1, R)) < frac)[0]
            pb = torch.full((4,), -1, dtype=torch.int64, device=dev)
            nb = torch.full((4,), -1, dtype=torch.int64, device=dev)
            only = k == 4  # one call that wants the promises only
            gpu.acceptor_phase1a_dev(0, rr, wm, None if tgt is None else t(tgt.view(np.int64)), pb, None if only else nb)
            ch = torch.zeros(S, dtype=torch.uint8, device=dev)
            nr = torch.zeros(S, dtype=torch.int32, device=dev)
            vt = None if tsub is None else W.bits_from_bool(W.random_subsets(rng, S, R, tsub[0], tsub[1]))
            gpu.phase2_fused_dev(t(slot), t(np.full(S, vr, np.int32)), t(val), None if vt is None else t(vt.view(np.int64)), ch, None, None, nr)
            outs.append((pb, None if only else nb, ch, nr, rr, tgt, wm, vr, vt))          # nothing synchronised so far
        assert gpu.sync() == 0
        for pb, nb, ch, nr, rr, tgt, wm, vr, vt in outs:
            st, pb_r, nb_r = ref.acceptor_phase1a(0, rr, wm, tgt)
            st, ch_r, cr_r, cv_r, nr_r = ref.phase2_fused(slot, np.full(S, vr, np.int32), val, vt)
            np.testing.assert_array_equal(pb.cpu().numpy().view(np.uint64), pb_r)
            if nb is not None:
                np.testing.assert_array_equal(nb.cpu().numpy().view(np.uint64), nb_r)
            np.testing.assert_array_equal(ch.cpu().numpy(), ch_r)
            np.testing.assert_array_equal(nr.cpu().numpy(), nr_r)
        if ballot_mode == 1 and not split:
            assert gpu.deferred_folds() >= 5, gpu.deferred_folds()   # folds kept riding in the vote launches, past the Phase1a's
        W.assert_same_state(gpu, ref, tally_slots=range(0, S, 211))
        gpu.set_stream(None)


@pytest.mark.parametrize("ballot_mode", [0, 1])
def test_pipelined_host_batches(fa, oracle, ballot_mode, monkeypatch):
    """big host batches run as a 3-stream pipeline over pieces (upload / K3 / download overlap).  Forced here at a
    small piece size: a well-formed batch, a batch with range errors (FPX_EINVAL: nothing applied), and batches
    that break the run contract in a LATER piece (duplicate slots, a round change) -- the pieces before it stay
    applied, the host-split replay takes over from the offending piece, results equal message-at-a-time delivery"""
    monkeypatch.setenv("FPX_HOST_PIECE", "1024")
    S, R = 1 << 15, 256
    gpu, ref = both(fa, oracle, num_slots=S, num_replicas=R, f=127, ballot_mode=ballot_mode, tally_ways=8)
    rng = np.random.default_rng(17 + ballot_mode)
    n = 9000
    slot = rng.permutation(S)[:n].astype(np.int32)
    rnd = np.zeros(n, np.int32)
    val = rng.integers(0, 1 << 30, n).astype(np.int32)
    tgt = W.bits_from_bool(W.random_subsets(rng, n, R, 100, R))
    check(gpu, ref, [("phase1a", 0, 0, 0, None), ("fused", slot, rnd, val, tgt)], tally_slots=range(0, S, 997))
    bad = slot.copy()
    bad[7000] = S + 5
    before = gpu.state_digest()
    a, b = gpu.phase2_fused(bad, rnd + 1, val), ref.phase2_fused(bad, rnd + 1, val)
    assert a[0] == b[0] == fa.FPX_EINVAL and gpu.error_detail() == ref.error_detail() == (7000, S + 5, 1)
    np.testing.assert_array_equal(gpu.state_digest(), before)
    dup = slot.copy()
    dup[5000:5050] = dup[4000:4050]                      # duplicates inside piece 4, and of piece-3 slots
    r2 = rnd + 1
    r2[8000:] = 2                                         # and a round change further on
    check(gpu, ref, [("fused", dup, r2, val, None), ("fused", slot[::-1].copy(), rnd + 3, val, tgt)],
          tally_slots=range(0, S, 997))
    monkeypatch.delenv("FPX_HOST_PIECE")


@pytest.mark.gpu
@pytest.mark.parametrize("n,shape", [(5000, "steady"), (300000, "steady"), (1 << 20, "steady"), (300000, "repeats"),
                                      (700001, "rounds")])
def test_page_locked_batches_are_staged_by_kernels(fa, oracle, n, shape):
    """fpx_phase2_fused with EVERY array in page-locked memory (fpx_host_alloc): the batch is staged by k_stage on its own
    streams in pieces, pipelined with the fused step, the outputs are written back to host memory by kernels too.  Same
    results as the oracle for batches that are one device run (steady), that repeat slots (split into runs by the
    replay), and that change rounds in the middle; a slot out of range is FPX_EINVAL with its index, nothing applied."""
    import ctypes as C

    S, R = 1 << 21, 5
    kw = dict(num_slots=S, num_replicas=R, f=2, tally_ways=8)
    gpu, ref = fa.Context(fa.make_config(**kw)), oracle.System(oracle.make_config(**kw))
    rng = np.random.default_rng(n)
    pins = {k: fa.PinnedArray((n,), dt) for k, dt in (("slot", np.int32), ("rnd", np.int32), ("val", np.int32), ("ch", np.uint8),
                                                      ("cr", np.int32), ("cv", np.int32), ("nr", np.int32))}
    tm = fa.PinnedArray((n, 4), np.uint64)
    slot, rnd, val = pins["slot"].array, pins["rnd"].array, pins["val"].array
    slot[:] = rng.permutation(S)[:n] if shape != "steady" else np.arange(n)
    rnd[:] = 0
    if shape == "repeats":
        slot[n // 2:] = slot[:n - n // 2]                 # every slot of the first half once more: Done -> ignored
    if shape == "rounds":
        rnd[n // 3:] = 2                                  # a leader change in the middle of the batch
        slot[:] = np.sort(slot)
    val[:] = W.steady_values(slot)
    tm.array[:] = 0
    tm.array[:, 0] = rng.integers(1, 32, n).astype(np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L = fa.lib()
    for k in ("ch", "cr", "cv", "nr"):
        pins[k].array[:] = 77
    st = L.fpx_phase2_fused(gpu._h, n, p(slot), p(rnd), p(val), p(tm.array), p(pins["ch"].array), p(pins["cr"].array),
                            p(pins["cv"].array), p(pins["nr"].array))
    b = ref.phase2_fused(slot.copy(), rnd.copy(), val.copy(), tm.array.copy())
    assert st == b[0] == 0
    for got, want in zip((pins["ch"].array, pins["cr"].array, pins["cv"].array, pins["nr"].array), b[1:]):
        np.testing.assert_array_equal(got, want)
    assert 0 < int(pins["ch"].array.sum()) < n
    np.testing.assert_array_equal(gpu.state_digest(), ref.state_digest())
    # a slot outside the window: refused whole, the first offender reported
    bad = n * 2 // 3
    slot[:] = (np.arange(n) + 7) % S
    slot[bad] = S
    before = gpu.state_digest()
    st = L.fpx_phase2_fused(gpu._h, n, p(slot), p(rnd), p(val), None, p(pins["ch"].array), None, p(pins["cv"].array), None)
    assert st == fa.FPX_EINVAL and gpu.error_detail()[0] == bad
    np.testing.assert_array_equal(gpu.state_digest(), before)
    for x in list(pins.values()) + [tm]:
        x.free()


@pytest.mark.gpu
def test_page_locked_calls_in_flight(fa, oracle):
    """fpx_phase2_fused_submit / _wait: three calls in flight, waited for in order, equal the oracle fed the same batches
    in the same order; a fourth submit is FPX_ECAPACITY; pageable arrays are FPX_EINVAL; a batch that is not one device
    run comes back FPX_EORDER with nothing applied (and the calls queued behind it with it), after which the
    synchronous call takes the same batch"""
    import ctypes as C

    S, R, n = 1 << 18, 5, 40000
    kw = dict(num_slots=S, num_replicas=R, f=2, tally_ways=8)
    gpu, ref = fa.Context(fa.make_config(**kw)), oracle.System(oracle.make_config(**kw))
    L = fa.lib()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rng = np.random.default_rng(8)

    def batch(k, shape="run"):
        a = {name: fa.PinnedArray((n,), dt) for name, dt in (("slot", np.int32), ("rnd", np.int32), ("val", np.int32),
                                                             ("ch", np.uint8), ("cr", np.int32), ("cv", np.int32), ("nr", np.int32))}
        a["slot"].array[:] = np.arange(k * n, (k + 1) * n)
        if shape == "twice":
            a["slot"].array[n // 2:] = a["slot"].array[:n - n // 2]
        a["rnd"].array[:] = k % 2
        a["val"].array[:] = W.steady_values(a["slot"].array) + k
        return a

    def submit(a):
        t = C.c_int32(-1)
        st = L.fpx_phase2_fused_submit(gpu._h, n, p(a["slot"].array), p(a["rnd"].array), p(a["val"].array), None,
                                       p(a["ch"].array), p(a["cr"].array), p(a["cv"].array), p(a["nr"].array), C.byref(t))
        return st, t.value

    def same(a):
        b = ref.phase2_fused(a["slot"].array.copy(), a["rnd"].array.copy(), a["val"].array.copy())
        assert b[0] == 0
        for k, want in zip(("ch", "cr", "cv", "nr"), b[1:]):
            np.testing.assert_array_equal(a[k].array, want)

    bs = [batch(k) for k in range(5)]
    tickets = [submit(b) for b in bs[:3]]
    assert [st for st, _ in tickets] == [0, 0, 0] and sorted(t for _, t in tickets) == [0, 1, 2]
    assert submit(bs[3])[0] == fa.FPX_ECAPACITY
    for (st, t), b in zip(tickets, bs[:3]):
        assert L.fpx_phase2_fused_wait(gpu._h, t) == 0
        same(b)
    assert L.fpx_phase2_fused_wait(gpu._h, tickets[0][1]) == fa.FPX_EINVAL           # not in flight any more
    pageable = np.zeros(n, np.int32)
    t = C.c_int32()
    assert L.fpx_phase2_fused_submit(gpu._h, n, p(pageable), p(pageable), p(pageable), None, None, None, None, None,
                                     C.byref(t)) == fa.FPX_EINVAL
    # a batch that repeats its slots is not one device run
    bad, after = batch(3, "twice"), bs[4]
    (s1, t1), (s2, t2) = submit(bad), submit(after)
    assert s1 == s2 == 0
    assert L.fpx_phase2_fused_wait(gpu._h, t1) == fa.FPX_EORDER
    assert L.fpx_phase2_fused_wait(gpu._h, t2) in (fa.FPX_EORDER, 0)
    np.testing.assert_array_equal(gpu.state_digest(), ref.state_digest())            # nothing of either was applied ...
    st = L.fpx_phase2_fused(gpu._h, n, p(bad["slot"].array), p(bad["rnd"].array), p(bad["val"].array), None, p(bad["ch"].array),
                            p(bad["cr"].array), p(bad["cv"].array), p(bad["nr"].array))
    assert st == 0
    same(bad)                                                                        # ... the synchronous call splits it
    np.testing.assert_array_equal(gpu.state_digest(), ref.state_digest())


@pytest.mark.gpu
@pytest.mark.parametrize("ballot_mode", [0, 1])
def test_page_locked_calls_on_the_headline_shape(fa, oracle, ballot_mode):
    """Round 6: a submitted call's inputs go up by the copy engine, and the vote kernel's output arrays ARE the caller's
    page-locked arrays (no copy down).  On the headline's shape (256 acceptors, dense): calls pumped three deep, a burst
    that ends with calls in flight waited for newest first, other entry points called between submit and wait, a stale
    leader's batch (Nack rounds through the same arrays) -- all equal the oracle fed the same batches in the same order."""
    import ctypes as C

    S, R, n = 1 << 19, 256, 1 << 15
    kw = dict(num_slots=S, num_replicas=R, f=127, tally_ways=8, ballot_mode=ballot_mode)
    gpu, ref = fa.Context(fa.make_config(**kw)), oracle.System(oracle.make_config(**kw))
    assert gpu.acceptor_phase1a(0, 0)[0] == 0 and ref.acceptor_phase1a(0, 0)[0] == 0
    L = fa.lib()
    p = lambda a: a.ctypes.data_as(C.c_void_p)

    def batch(k, rnd=0):
        a = {name: fa.PinnedArray((n,), dt) for name, dt in (("slot", np.int32), ("rnd", np.int32), ("val", np.int32),
                                                             ("ch", np.uint8), ("cr", np.int32), ("cv", np.int32), ("nr", np.int32))}
        a["slot"].array[:] = np.arange(k * n, (k + 1) * n)
        a["rnd"].array[:] = rnd
        a["val"].array[:] = W.steady_values(a["slot"].array) + k
        for o in ("cr", "cv", "nr"):
            a[o].array[:] = -7
        return a

    def submit(a):
        t = C.c_int32(-1)
        assert L.fpx_phase2_fused_submit(gpu._h, n, p(a["slot"].array), p(a["rnd"].array), p(a["val"].array), None,
                                         p(a["ch"].array), p(a["cr"].array), p(a["cv"].array), p(a["nr"].array), C.byref(t)) == 0
        return t.value

    def same(a):
        b = ref.phase2_fused(a["slot"].array.copy(), a["rnd"].array.copy(), a["val"].array.copy())
        assert b[0] == 0
        for k, want in zip(("ch", "cr", "cv", "nr"), b[1:]):
            np.testing.assert_array_equal(a[k].array, want, err_msg=k)

    bs = [batch(k) for k in range(9)]
    # (1) the pump of bench.py --config host_path: three in flight, the oldest waited for before the next submit
    inflight = []
    for b in bs[:6]:
        if len(inflight) == 3:
            t, done = inflight.pop(0)
            assert L.fpx_phase2_fused_wait(gpu._h, t) == 0
            same(done)
        inflight.append((submit(b), b))
    # (2) the burst ends: two of the three calls in flight were never launched or never sent; newest first
    for t, done in reversed(inflight):
        assert L.fpx_phase2_fused_wait(gpu._h, t) == 0
    for _, done in inflight:
        same(done)
    np.testing.assert_array_equal(gpu.state_digest(), ref.state_digest())
    # (3) another entry point between submit and wait sees the submitted calls applied
    t6, t7 = submit(bs[6]), submit(bs[7])
    got = gpu.read_acceptor(0, 200)
    ref.phase2_fused(bs[6]["slot"].array.copy(), bs[6]["rnd"].array.copy(), bs[6]["val"].array.copy())
    ref.phase2_fused(bs[7]["slot"].array.copy(), bs[7]["rnd"].array.copy(), bs[7]["val"].array.copy())
    want = ref.read_acceptor(0, 200)
    assert got[:2] == want[:2]
    for u, v in zip(got[2:], want[2:]):
        np.testing.assert_array_equal(u, v)
    assert L.fpx_phase2_fused_wait(gpu._h, t7) == 0 and L.fpx_phase2_fused_wait(gpu._h, t6) == 0
    for k in (6, 7):
        b = bs[k]
        assert b["ch"].array.all() and (b["cv"].array == b["val"].array).all() and (b["cr"].array == 0).all()
    # (4) a stale leader's batch among calls in flight: Nacks come back through the same records
    assert gpu.acceptor_phase1a(0, 1)[0] == 0 and ref.acceptor_phase1a(0, 1)[0] == 0
    stale = batch(8, rnd=0)
    t8 = submit(stale)
    assert L.fpx_phase2_fused_wait(gpu._h, t8) == 0
    same(stale)
    assert (stale["nr"].array == 1).all() and not stale["ch"].array.any()
    np.testing.assert_array_equal(gpu.state_digest(), ref.state_digest())
    gpu.close()


@pytest.mark.gpu
@pytest.mark.parametrize("R,q", [(256, 128), (4, 3)])
def test_the_fold_that_rides_in_the_next_launch_is_not_observable(fa, oracle, monkeypatch, R, q):
    """Round 6: with a ballot per cell the fold of a fused step's maxima (Acceptor.round / maxVotedSlot as per-acceptor
    scalars) waits for the NEXT fused step and rides in its launch (k_phase2_fin; include/fpx.h, fpx_deferred_folds).
    Device-resident steps back to back -- the adversarial stream with its Phase1a's (every one of them an entry point
    that folds what is pending first), read-backs between steps, a reset in the middle -- leave exactly the state of a
    context that folds at once (FPX_NO_DEFER_FINALIZE=1) and of the oracle; and folds did ride."""
    import torch

    S = 1 << 15
    kw = dict(num_slots=S, num_replicas=R, f=q - 1, ballot_mode=1, tally_ways=8)
    script = W.adversarial_script(S, R, q, 5, epochs=32, fused=True, subsets=W.fast_subsets)
    dev = torch.device("cuda:0")
    d = lambda a, view=None: torch.from_numpy(np.ascontiguousarray(a) if view is None else np.ascontiguousarray(a).view(view)).to(dev)

    def run(ctx):
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        seen = []
        for k, op in enumerate(script):
            if op[0] == "phase1a":
                _, g_, rnd, wm, tgt = op
                ctx.acceptor_phase1a_dev(g_, rnd, wm, None if tgt is None else d(tgt, np.int64))
            else:
                _, slot, rr, val, tgt = op
                n = len(slot)
                outs = [torch.zeros(n, dtype=torch.uint8, device=dev)] + [torch.full((n,), -7, dtype=torch.int32, device=dev) for _ in range(3)]
                ctx.phase2_fused_dev(d(slot), d(rr), d(val), d(tgt, np.int64), *outs)
                if k % 7 == 3:
                    seen.append(ctx.read_acceptor(0, R - 1)[:2])          # (promised, maxVotedSlot) between two steps
        assert ctx.sync() == 0
        return seen

    monkeypatch.delenv("FPX_NO_DEFER_FINALIZE", raising=False)
    a = fa.Context(fa.make_config(flags=fa.FPX_F_SCATTERED_TARGETS | fa.FPX_F_TRUSTED, **kw))
    seen_a = run(a)
    assert a.deferred_folds() > 8, a.deferred_folds()
    monkeypatch.setenv("FPX_NO_DEFER_FINALIZE", "1")
    b = fa.Context(fa.make_config(flags=fa.FPX_F_SCATTERED_TARGETS | fa.FPX_F_TRUSTED, **kw))
    seen_b = run(b)
    assert b.deferred_folds() == 0 and seen_a == seen_b
    ref = oracle.System(oracle.make_config(**kw))
    W.run_script(ref, script)
    np.testing.assert_array_equal(a.state_digest(), b.state_digest())
    np.testing.assert_array_equal(a.state_digest(), ref.state_digest())
    for r in (0, R // 2, R - 1):
        x, y, z = a.read_acceptor(0, r), b.read_acceptor(0, r), ref.read_acceptor(0, r)
        assert x[:2] == y[:2] == z[:2]
    # a reset with a fold pending, then the steady stream: the scalars are the new life's
    monkeypatch.delenv("FPX_NO_DEFER_FINALIZE")
    slot, rnd, val = W.steady_stream(4096)
    outs = [torch.zeros(4096, dtype=torch.uint8, device=dev)] + [torch.full((4096,), -7, dtype=torch.int32, device=dev) for _ in range(3)]
    a.phase2_fused_dev(d(slot), d(rnd + 40), d(val), None, *outs)
    a.reset()
    ref.reset()
    assert a.acceptor_phase1a(0, 0)[0] == 0 and ref.acceptor_phase1a(0, 0)[0] == 0
    a.phase2_fused_dev(d(slot), d(rnd), d(val), None, *outs)
    a.phase2_fused_dev(d(slot + 4096), d(rnd), d(val), None, *outs)
    ref.phase2_fused(slot, rnd, val)
    ref.phase2_fused(slot + 4096, rnd, val)
    assert a.sync() == 0 and bool(outs[0].all())
    np.testing.assert_array_equal(a.state_digest(), ref.state_digest())
    assert a.read_acceptor(0, 0)[:2] == ref.read_acceptor(0, 0)[:2]
    a.close(), b.close()
