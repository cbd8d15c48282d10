"""Replica-axis shards of every geometry bench.py --shard replica can produce (2, 4, 8 ranks): each
shard context owns acceptors [base, base + R/N) of one 256-acceptor group; K1 on every shard, the sum
of the partial bitmaps (what the RCCL all-reduce computes), K2 on the full bitmaps == the unsharded
oracle.  Also the Phase-1 scan and Nacks on a shard with a non-zero base."""
import numpy as np
import pytest

from tests import workloads as W

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("ballot_mode", [0, 1])
def test_n_way_replica_shards(oracle, world, ballot_mode):
    import frankenpaxos_amd as fa
    from frankenpaxos_amd import sharding

    S, R = 1024, 256
    whole = oracle.System(oracle.make_config(num_slots=S, num_replicas=R, f=127, ballot_mode=ballot_mode,
                                             tally_ways=8))
    shards = []
    for rank in range(world):
        base, n = sharding.replica_shard(R, world, rank)
        shards.append(fa.Context(fa.make_config(num_slots=S, num_replicas=n, f=127, ballot_mode=ballot_mode,
                                                replica_base=base, replicas_total=R, tally_ways=8)))
    rng = np.random.default_rng(world * 10 + ballot_mode)
    slot, rnd, val = W.steady_stream(S)
    ahead = W.bits_from_bool(W.random_subsets(rng, 1, R, 60, 60))[0]   # 60 acceptors promised round 3
    whole.acceptor_phase1a(0, 3, 0, ahead)
    for sh in shards:
        assert sh.acceptor_phase1a(0, 3, 0, ahead)[0] == 0
    for step, r in enumerate((1, 3)):                                   # round 1: Nacks; round 3: all vote
        tgt = W.bits_from_bool(W.random_subsets(rng, S, R, 150, 256))
        rr = np.full(S, r, np.int32)
        whole.proxy_open(slot, rr, val)
        st, vb_ref, nb_ref, nr_ref = whole.acceptor_phase2a(slot, rr, val, tgt)
        st, ch_ref, cr_ref, cv_ref = whole.proxy_phase2b(slot, rr, vb_ref)
        vsum = np.zeros((S, 4), np.uint64)
        nsum = np.zeros((S, 4), np.uint64)
        nmax = np.full(S, -1, np.int32)
        for sh in shards:
            st, vb, nb, nr = sh.acceptor_phase2a(slot, rr, val, tgt)
            assert st == 0 and not (vsum & vb).any()
            vsum += vb
            nsum += nb
            nmax = np.maximum(nmax, nr)
        np.testing.assert_array_equal(vsum, vb_ref)
        np.testing.assert_array_equal(nsum, nb_ref)
        np.testing.assert_array_equal(nmax, nr_ref)
        for sh in (shards[0], shards[-1]):                              # any rank can tally the full bitmaps
            sh.proxy_open(slot, rr, val)
            st, ch, cr, cv = sh.proxy_phase2b(slot, rr, vsum)
            np.testing.assert_array_equal(ch, ch_ref)
            np.testing.assert_array_equal(cv, cv_ref)
        assert 0 < ch_ref.sum() <= S
    # state of every shard == the matching columns of the unsharded oracle
    vr_ref, vv_ref, bl_ref = whole.read_state()
    pr_ref, mv_ref = whole.read_scalars()
    for rank, sh in enumerate(shards):
        base, n = sharding.replica_shard(R, world, rank)
        vr, vv, bl = sh.read_state()
        np.testing.assert_array_equal(vr, vr_ref[:, base:base + n])
        np.testing.assert_array_equal(vv, vv_ref[:, base:base + n])
        np.testing.assert_array_equal(bl, bl_ref[:, base:base + n])
        pr, mv = sh.read_scalars()
        np.testing.assert_array_equal(pr, pr_ref[:, base:base + n])
        np.testing.assert_array_equal(mv, mv_ref[:, base:base + n])
        # Phase-1 recovery scan restricted to the shard's acceptors
        q = W.bits_from_bool((np.arange(R) >= base)[None, :] & (np.arange(R) < base + n)[None, :])
        a, b = sh.leader_phase1b_scan(0, q, S), whole.leader_phase1b_scan(0, q, S)
        assert a[1] == b[1]
        np.testing.assert_array_equal(a[2], b[2])
        np.testing.assert_array_equal(a[3], b[3])


@pytest.mark.parametrize("with_comm", [False, True])
@pytest.mark.parametrize("ballot_mode", [0, 1])
def test_replica_sharded_entry_point_world_one(oracle, with_comm, ballot_mode):
    """fpx_phase2_replica_sharded_dev on the degenerate world of one rank -- with a real RCCL communicator of
    one rank created through the C ABI (fpx_comm_unique_id / fpx_comm_create) and without any -- is K1 ->
    (reduce-scatter over one rank) -> open + K2, and must equal the oracle's unfused pipeline bit for bit."""
    import torch
    import frankenpaxos_amd as fa

    S, R = 8192, 256
    kw = dict(num_slots=S, num_replicas=R, f=127, ballot_mode=ballot_mode, tally_ways=8)
    gpu, ref = fa.Context(fa.make_config(**kw)), oracle.System(oracle.make_config(**kw))
    dev = torch.device("cuda:0")
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    if with_comm:
        uid = fa.comm_unique_id()
        assert len(uid) == fa.FPX_COMM_ID_BYTES
        gpu.comm_create(uid, 0, 1)
        assert gpu.comm_info() == (0, 1)
    gpu.profile_enable(True)
    rng = np.random.default_rng(9 + ballot_mode)
    slot, rnd, val = W.steady_stream(S)
    ahead = W.bits_from_bool(W.random_subsets(rng, 1, R, 100, 100))[0]
    for be in (gpu, ref):
        assert be.acceptor_phase1a(0, 0)[0] == 0
        assert be.acceptor_phase1a(0, 4, 0, ahead)[0] == 0        # 100 acceptors are ahead: Nacks in round 2
    for r in (2, 4):
        rr = np.full(S, r, np.int32)
        tgt = W.bits_from_bool(W.random_subsets(rng, S, R, 120, 256))
        t = lambda a: torch.from_numpy(a).to(dev)
        ch = torch.zeros(S, dtype=torch.uint8, device=dev)
        cr = torch.zeros(S, dtype=torch.int32, device=dev)
        cv = torch.zeros(S, dtype=torch.int32, device=dev)
        nr = torch.zeros(S, dtype=torch.int32, device=dev)
        gpu.phase2_replica_sharded_dev(t(slot), t(rr), t(val), t(tgt.view(np.int64)), ch, cr, cv, nr)
        assert gpu.sync() == 0
        ref.proxy_open(slot, rr, val)
        st, vb_r, nb_r, nr_r = ref.acceptor_phase2a(slot, rr, val, tgt)
        st, ch_r, cr_r, cv_r = ref.proxy_phase2b(slot, rr, vb_r)
        np.testing.assert_array_equal(ch.cpu().numpy(), ch_r)
        np.testing.assert_array_equal(cr.cpu().numpy(), cr_r)
        np.testing.assert_array_equal(cv.cpu().numpy(), cv_r)
        np.testing.assert_array_equal(nr.cpu().numpy(), nr_r)
    assert 0 < int(ch_r.sum())
    W.assert_same_state(gpu, ref, tally_slots=range(0, S, 257))
    np.testing.assert_array_equal(gpu.state_digest(), ref.state_digest())
    n_coll, ms = gpu.profile_read_collective()
    assert n_coll == (2 if with_comm else 0)  # a communicator, even of one rank, goes through ncclReduceScatter
    # the all-gather of Chosen records (group sharding) degenerates to a copy
    allch = torch.zeros(S, dtype=torch.uint8, device=dev)
    allcv = torch.zeros(S, dtype=torch.int32, device=dev)
    gpu.comm_allgather_chosen_dev(ch, None, cv, allch, None, allcv)
    assert gpu.sync() == 0 and bool((allch == ch).all()) and bool((allcv == cv).all())
    if with_comm:
        gpu.comm_destroy()
        assert gpu.comm_info() == (0, 1)
    gpu.set_stream(None)
    gpu.close()


def test_replica_sharded_entry_point_rejects_ragged_batches():
    import torch
    import frankenpaxos_amd as fa

    gpu = fa.Context(fa.make_config(num_slots=64, num_replicas=4, f=1))
    with pytest.raises(fa.FpxError):
        gpu.comm_create(b"\0" * 128, 3, 2)        # rank outside the world: FPX_EINVAL before RCCL is touched
    gpu.close()
