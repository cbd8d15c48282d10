"""fpx_phase2_replica_sharded_dev -- the SHIPPED multi-GPU exchange path: K1 on my acceptor columns ->
ncclReduceScatter(sum, u64) of the partial vote bitmaps -> ncclAllReduce(max) of the Nack rounds -> open + K2 on my
slots, fpx_api.hip -- with a world of TWO ranks.  The boxes this build runs on have one GPU and RCCL refuses two ranks
on one device, so the two ranks are two processes with a libfpx context each on the one GPU, and the collectives are
tests/rccl_double/fpx_fake_rccl.c (FPX_RCCL_LIB: the hook libfpx binds RCCL through).  Everything above the five ncclXxx
symbols is the product's own code and the oracle is the unsharded group.  ("so that the first real
8-GPU run is not also the first run of that code with N > 1".)  Round 5: the same with worlds of
FOUR and EIGHT ranks -- the geometry the driver's 8-GPU run uses: 32 acceptors per rank (the G = 8 kernels), an 8-slice
reduce-scatter, the all-reduce(max) of Nack rounds over eight ranks, the all-gather of Chosen records inside one
ncclGroupStart / ncclGroupEnd, and a batch whose size the world does not divide refused with FPX_EINVAL."""
import multiprocessing as mp
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOUBLE = os.path.join(ROOT, "tests", "build", "libfpx_fake_rccl.so")


def build_double():
    src = os.path.join(ROOT, "tests", "rccl_double", "fpx_fake_rccl.c")
    if os.path.exists(DOUBLE) and os.path.getmtime(DOUBLE) >= os.path.getmtime(src):
        return DOUBLE
    os.makedirs(os.path.dirname(DOUBLE), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-I/opt/rocm/include", "-o", DOUBLE, src,
                           "-L/opt/rocm/lib", "-lamdhip64", "-lrt"])
    return DOUBLE


def _rank_main(rank, world, so, conn, ballot_mode):
    try:
        os.environ["FPX_RCCL_LIB"] = so          # before libfpx binds its collectives (once per process)
        import sys
        sys.path.insert(0, ROOT)
        import torch
        import frankenpaxos_amd as fa
        from frankenpaxos_amd import sharding
        from oracle import pyoracle
        from tests import workloads as W

        pyoracle.build()
        S, R = 4096, 256
        whole = pyoracle.System(pyoracle.make_config(num_slots=S, num_replicas=R, f=127, ballot_mode=ballot_mode, tally_ways=8))
        base, nloc = sharding.replica_shard(R, world, rank)
        gpu = fa.Context(fa.make_config(num_slots=S, num_replicas=nloc, f=127, ballot_mode=ballot_mode,
                                        replica_base=base, replicas_total=R, tally_ways=8))
        dev = torch.device("cuda:0")
        gpu.set_stream(torch.cuda.current_stream().cuda_stream)
        if rank == 0:
            uid = fa.comm_unique_id()
            conn.send(uid)
        else:
            uid = conn.recv()
        gpu.comm_create(uid, rank, world)
        assert gpu.comm_info() == (rank, world)
        gpu.profile_enable(True)
        rng = np.random.default_rng(31 + ballot_mode)      # the same stream on every rank
        slot, rnd, val = W.steady_stream(S)
        ahead = W.bits_from_bool(W.random_subsets(rng, 1, R, 100, 100))[0]   # 100 acceptors, on both ranks, are ahead
        for be in (gpu, whole):
            assert be.acceptor_phase1a(0, 0)[0] == 0
            assert be.acceptor_phase1a(0, 4, 0, ahead)[0] == 0
        per = S // world
        lo, hi = rank * per, (rank + 1) * per
        t = lambda a: torch.from_numpy(a).to(dev)
        for r in (2, 4):                                     # round 2: Nacks from acceptors of BOTH ranks; round 4: all vote
            rr = np.full(S, r, np.int32)
            tgt = W.bits_from_bool(W.random_subsets(rng, S, R, 120, 256))
            ch = torch.zeros(per, dtype=torch.uint8, device=dev)
            cr = torch.zeros(per, dtype=torch.int32, device=dev)
            cv = torch.zeros(per, dtype=torch.int32, device=dev)
            nr = torch.zeros(S, dtype=torch.int32, device=dev)
            gpu.phase2_replica_sharded_dev(t(slot), t(rr), t(val), t(tgt.view(np.int64)), ch, cr, cv, nr)
            assert gpu.sync() == 0
            whole.proxy_open(slot, rr, val)
            st, vb_r, nb_r, nr_r = whole.acceptor_phase2a(slot, rr, val, tgt)
            st, ch_r, cr_r, cv_r = whole.proxy_phase2b(slot, rr, vb_r)
            np.testing.assert_array_equal(ch.cpu().numpy(), ch_r[lo:hi])        # my slots, tallied over ALL acceptors' votes
            np.testing.assert_array_equal(cr.cpu().numpy(), cr_r[lo:hi])
            np.testing.assert_array_equal(cv.cpu().numpy(), cv_r[lo:hi])
            np.testing.assert_array_equal(nr.cpu().numpy(), nr_r)               # the largest Nack round over BOTH ranks
            if r == 2:
                assert (nr_r >= 0).any() and 0 < int(ch_r.sum()) < S
            # group sharding's exchange: every rank learns all Chosen records
            allch = torch.zeros(S, dtype=torch.uint8, device=dev)
            allcr = torch.zeros(S, dtype=torch.int32, device=dev)
            allcv = torch.zeros(S, dtype=torch.int32, device=dev)
            gpu.comm_allgather_chosen_dev(ch, cr, cv, allch, allcr, allcv)
            assert gpu.sync() == 0
            np.testing.assert_array_equal(allch.cpu().numpy(), ch_r)
            np.testing.assert_array_equal(allcr.cpu().numpy(), cr_r)
            np.testing.assert_array_equal(allcv.cpu().numpy(), cv_r)
        # my acceptors' state == the matching columns of the unsharded group; my tallies == the group's, for my slots
        vr_ref, vv_ref, bl_ref = whole.read_state()
        vr, vv, bl = gpu.read_state()
        np.testing.assert_array_equal(vr, vr_ref[:, base:base + nloc])
        np.testing.assert_array_equal(vv, vv_ref[:, base:base + nloc])
        np.testing.assert_array_equal(bl, bl_ref[:, base:base + nloc])
        pr_ref, mv_ref = whole.read_scalars()
        pr, mv = gpu.read_scalars()
        np.testing.assert_array_equal(pr, pr_ref[:, base:base + nloc])
        np.testing.assert_array_equal(mv, mv_ref[:, base:base + nloc])
        for s in range(lo, hi, 97):
            assert gpu.read_tally(int(s)) == whole.read_tally(int(s)), "tally of slot %d" % s
        n_coll, _ = gpu.profile_read_collective()
        assert n_coll == 2 + 2                               # a reduce-scatter per step, an all-gather group per step
        import ctypes
        assert ctypes.CDLL(so).fpx_fake_rccl_groups_closed() == 2      # the three all-gathers of a step travel as ONE group
        # a batch the world does not divide has no slice for every rank: refused before anything is launched
        k = S - 1 if (S - 1) % world else S - 3
        with pytest.raises(fa.FpxError) as err:
            gpu.phase2_replica_sharded_dev(t(slot[:k]), t(np.full(k, 6, np.int32)), t(val[:k]), None, ch, cr, cv, nr)
        assert err.value.status == fa.FPX_EINVAL
        gpu.comm_destroy()
        gpu.set_stream(None)
        gpu.close()
        conn.send("ok")
    except BaseException as e:  # noqa: BLE001 -- the parent reports
        import traceback
        conn.send("rank %d: %s\n%s" % (rank, e, traceback.format_exc()))


@pytest.mark.parametrize("world,ballot_mode", [(2, 0), (2, 1), (4, 1), (8, 0), (8, 1)])
def test_replica_sharded_entry_point_world_two(world, ballot_mode):
    so = build_double()
    ctx = mp.get_context("spawn")
    pipes = [ctx.Pipe() for _ in range(world)]
    procs = [ctx.Process(target=_rank_main, args=(r, world, so, pipes[r][1], ballot_mode)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        assert pipes[0][0].poll(400), "rank 0 never produced the communicator id"
        uid = pipes[0][0].recv()
        assert isinstance(uid, bytes), uid
        for r in range(1, world):
            pipes[r][0].send(uid)
        for r in range(world):
            assert pipes[r][0].poll(600), "rank %d did not finish" % r
            msg = pipes[r][0].recv()
            assert msg == "ok", msg
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.terminate()
