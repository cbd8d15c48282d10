"""The N > 1 paths on CPU: world_size 2, gloo backend, one process per (pretend) GPU.  The per-rank
compute is done by the oracle here (test infrastructure); what is under test is the sharding logic
of frankenpaxos_amd/sharding.py and the exchange step (all-reduce(sum) of disjoint vote bitmaps)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import workloads as W

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        ret[rank] = fn(rank)
    finally:
        dist.destroy_process_group()


def _spawn(fn):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(_free_port(), fn, ret), nprocs=WORLD, join=True)
    return [ret[r] for r in range(WORLD)]


S, R = 1024, 256


def _replica_axis(rank):
    from frankenpaxos_amd import sharding
    from oracle import pyoracle as O

    base, n = sharding.replica_shard(R, WORLD, rank)
    shard = O.System(O.make_config(num_slots=S, num_replicas=n, f=127, replica_base=base,
                                   replicas_total=R, tally_ways=8))
    rng = np.random.default_rng(42)  # same stream on every rank
    slot, rnd, val = W.steady_stream(S)
    tgt = W.bits_from_bool(W.random_subsets(rng, S, R, 135, 165))
    shard.acceptor_phase1a(0, 3, 0, W.bits_from_bool((np.arange(R) % 7 == 0)[None, :])[0])
    st, vb, nb, nr = shard.acceptor_phase2a(slot, rnd + 2, val, tgt)
    t = torch.from_numpy(vb.view(np.int64).copy())
    sharding.allreduce_vote_bitmaps(t)                      # the exchange step (RCCL on the GPU box)
    full = t.numpy().view(np.uint64)
    shard.proxy_open(slot, rnd + 2, val)
    st, ch, cr, cv = shard.proxy_phase2b(slot, rnd + 2, full)
    return full.copy(), ch, cv


def test_replica_axis_sharding_allreduce_sum_is_or():
    from oracle import pyoracle as O

    outs = _spawn(_replica_axis)
    whole = O.System(O.make_config(num_slots=S, num_replicas=R, f=127, tally_ways=8))
    rng = np.random.default_rng(42)
    slot, rnd, val = W.steady_stream(S)
    tgt = W.bits_from_bool(W.random_subsets(rng, S, R, 135, 165))
    whole.acceptor_phase1a(0, 3, 0, W.bits_from_bool((np.arange(R) % 7 == 0)[None, :])[0])
    whole.proxy_open(slot, rnd + 2, val)
    st, vb, nb, nr = whole.acceptor_phase2a(slot, rnd + 2, val, tgt)
    st, ch, cr, cv = whole.proxy_phase2b(slot, rnd + 2, vb)
    assert 0 < ch.sum() < S
    for full, ch_r, cv_r in outs:
        np.testing.assert_array_equal(full, vb)
        np.testing.assert_array_equal(ch_r, ch)
        np.testing.assert_array_equal(cv_r, cv)


GROUPS = 6


def _group_axis(rank):
    from frankenpaxos_amd import sharding
    from oracle import pyoracle as O

    sysm = O.System(O.make_config(num_slots=S, num_replicas=3, num_groups=GROUPS, f=1, tally_ways=8))
    slot, rnd, val = W.steady_stream(S)
    mine = sharding.slots_of_rank(slot, GROUPS, 1, WORLD, rank)
    assert set(sharding.group_of_slot(slot[mine], GROUPS) % WORLD) == {rank}
    st, ch, cr, cv, nr = sysm.phase2_fused(slot[mine], rnd[mine], val[mine])
    committed = torch.tensor([int(ch.sum())])
    dist.all_reduce(committed)                               # only the count / chosen records leave a rank
    out = np.full(S, -1, np.int32)
    out[mine] = cv
    t = torch.from_numpy(out)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                 # gather of the chosen values
    return int(committed.item()), t.numpy().copy(), sharding.groups_of_rank(GROUPS, WORLD, rank)


def test_group_sharding_needs_no_exchange():
    outs = _spawn(_group_axis)
    slot, rnd, val = W.steady_stream(S)
    assert sorted(outs[0][2] + outs[1][2]) == list(range(GROUPS))
    for committed, cv, _ in outs:
        assert committed == S
        np.testing.assert_array_equal(cv, val)


def test_replica_shard_geometry():
    from frankenpaxos_amd import sharding

    assert [sharding.replica_shard(256, 8, r) for r in (0, 3, 7)] == [(0, 32), (96, 32), (224, 32)]
    assert sharding.replica_shard(256, 1, 0) == (0, 256)
    with pytest.raises(ValueError):
        sharding.replica_shard(255, 2, 0)
    with pytest.raises(ValueError):
        sharding.replica_shard(24, 4, 0)  # 6 per rank: not a multiple of 4
    s = np.arange(100)
    g = sharding.group_of_slot(s, 2, 5)   # mencius map
    assert (g == (s % 5) * 2 + (s // 5) % 2).all()


def _replica_axis_reduce_scatter(rank):
    from frankenpaxos_amd import sharding
    from oracle import pyoracle as O

    base, n = sharding.replica_shard(R, WORLD, rank)
    shard = O.System(O.make_config(num_slots=S, num_replicas=n, f=127, replica_base=base,
                                   replicas_total=R, tally_ways=8))
    rng = np.random.default_rng(42)
    slot, rnd, val = W.steady_stream(S)
    tgt = W.bits_from_bool(W.random_subsets(rng, S, R, 135, 165))
    shard.acceptor_phase1a(0, 3, 0, W.bits_from_bool((np.arange(R) >= 200)[None, :])[0])   # all on rank 1's shard
    st, vb, nb, nr = shard.acceptor_phase2a(slot, rnd + 2, val, tgt)       # K1: all slots, my acceptors
    mine = sharding.reduce_scatter_vote_bitmaps(torch.from_numpy(vb.view(np.int64).copy()))
    lo, hi = sharding.slot_slice(S, WORLD, rank)
    full = mine.numpy().view(np.uint64)
    shard.proxy_open(slot[lo:hi], rnd[lo:hi] + 2, val[lo:hi])              # K2: my slice of the slots only
    st, ch, cr, cv = shard.proxy_phase2b(slot[lo:hi], rnd[lo:hi] + 2, full)
    own = nr.copy()
    nack = sharding.allreduce_nack_rounds(torch.from_numpy(nr.copy())).numpy()  # the Nacks of every rank's acceptors
    return lo, hi, full.copy(), ch, cv, nack.copy(), own


def test_replica_axis_sharding_reduce_scatter_slices_the_tally():
    from oracle import pyoracle as O

    outs = _spawn(_replica_axis_reduce_scatter)
    whole = O.System(O.make_config(num_slots=S, num_replicas=R, f=127, tally_ways=8))
    rng = np.random.default_rng(42)
    slot, rnd, val = W.steady_stream(S)
    tgt = W.bits_from_bool(W.random_subsets(rng, S, R, 135, 165))
    whole.acceptor_phase1a(0, 3, 0, W.bits_from_bool((np.arange(R) >= 200)[None, :])[0])   # all on rank 1's shard
    whole.proxy_open(slot, rnd + 2, val)
    st, vb, nb, nr = whole.acceptor_phase2a(slot, rnd + 2, val, tgt)
    st, ch, cr, cv = whole.proxy_phase2b(slot, rnd + 2, vb)
    covered = 0
    for lo, hi, full, ch_r, cv_r, nack, own in outs:
        np.testing.assert_array_equal(full, vb[lo:hi])
        np.testing.assert_array_equal(ch_r, ch[lo:hi])
        np.testing.assert_array_equal(cv_r, cv[lo:hi])
        np.testing.assert_array_equal(nack, nr)             # == the whole group's largest Nacked round per message
        covered += hi - lo
    assert covered == S
    # ... which no single rank saw on its own: each shard's acceptors Nack only some of the messages
    assert (nr >= 0).any() and any((own != nr).any() for *_, own in outs)
