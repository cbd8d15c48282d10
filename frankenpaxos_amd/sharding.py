"""Multi-GPU host logic: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The Phase-2 path shards two ways:

(1) by acceptor group -- no exchange step.  Non-flexible MultiPaxos sends slot s to acceptor group
    s % numAcceptorGroups (multipaxos/ProxyLeader.scala:190; Mencius: leader group s % L, then
    (s / L) % A, mencius/ProxyLeader.scala:231-234) and groups never interact in Phase 2.  Rank r owns
    the groups g with g % world == r, holds all R acceptors of those groups, tallies locally; only the
    chosen records leave the GPU.

(2) by the acceptor (replica) axis -- one exchange step.  For one big group (flexible mode: "the log
    is not partitioned", multipaxos/Config.scala:16-21) rank r owns acceptors
    [r * R/world, (r+1) * R/world); K1 on every rank yields partial per-slot vote bitmaps whose set
    bits lie in the rank's own range, one all-reduce(sum) of the uint64 words yields the full bitmaps
    (the bit ranges are disjoint, so sum == OR and never carries; RCCL has no bitwise-OR reduction),
    then K2 tallies.

Nothing here touches protocol state; the kernels do.
"""
import numpy as np


def groups_of_rank(num_groups_total, world, rank):
    """acceptor groups owned by `rank` under slot-partition sharding"""
    return [g for g in range(num_groups_total) if g % world == rank]


def group_of_slot(slot, num_groups, num_leader_groups=1):
    """slot -> group id, vectorised (the same map the kernels use: fpx_kernels.hpp group_of_slot)"""
    slot = np.asarray(slot)
    return (slot % num_leader_groups) * num_groups + (slot // num_leader_groups) % num_groups


def slots_of_rank(slot, num_groups, num_leader_groups, world, rank):
    """boolean mask of the messages whose acceptor group lives on `rank`"""
    return group_of_slot(slot, num_groups, num_leader_groups) % world == rank


def replica_shard(replicas_total, world, rank):
    """(replica_base, num_replicas) of `rank` under replica-axis sharding; bases are multiples of 4"""
    if replicas_total % world or (replicas_total // world) % 4:
        raise ValueError("replica-axis sharding needs replicas_total / world to be a multiple of 4")
    n = replicas_total // world
    return rank * n, n


def allreduce_vote_bitmaps(bitmaps, group=None):
    """in-place all-reduce(sum) of partial per-slot vote bitmaps ([n, 4] int64 view of the uint64
    words); with disjoint bit ownership this is the bitwise OR.  32 B per slot: 32 MiB for 2^20 slots,
    a ring moves 2 (G-1)/G of that per GPU over one xGMI link."""
    import torch.distributed as dist

    assert bitmaps.dtype.is_floating_point is False and bitmaps.element_size() == 8
    dist.all_reduce(bitmaps, op=dist.ReduceOp.SUM, group=group)
    return bitmaps


def allreduce_nack_rounds(nack_round, group=None):
    """in-place all-reduce(max) of the per-message Nack rounds: a Nack from any rank's acceptors reaches the
    leader (fpx_phase2_replica_sharded_dev does the same with ncclAllReduce(ncclMax) behind the C ABI)"""
    import torch.distributed as dist

    dist.all_reduce(nack_round, op=dist.ReduceOp.MAX, group=group)
    return nack_round


def slot_slice(n, world, rank):
    """[lo, hi) of the messages whose tally `rank` runs after a reduce-scatter of the bitmaps"""
    if n % world:
        raise ValueError("the batch must divide evenly over the ranks")
    per = n // world
    return rank * per, (rank + 1) * per


def reduce_scatter_vote_bitmaps(bitmaps, out=None, group=None):
    """reduce-scatter(sum) of the partial per-slot vote bitmaps: rank r receives the FULL bitmaps of its
    slice of the batch (slot_slice) and tallies only those -- half the bytes of the all-reduce on a ring
    ((G-1)/G x 32 B per slot per GPU) and 1/G of the K2 work per GPU.  Backends without a reduce-scatter
    (gloo, the CPU tests) fall back to all-reduce + slice."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = slot_slice(bitmaps.shape[0], world, rank)
    if out is None:
        out = torch.empty((hi - lo,) + tuple(bitmaps.shape[1:]), dtype=bitmaps.dtype, device=bitmaps.device)
    if dist.get_backend(group) == "nccl":
        dist.reduce_scatter_tensor(out, bitmaps, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.all_reduce(bitmaps, op=dist.ReduceOp.SUM, group=group)
        out.copy_(bitmaps[lo:hi])
    return out
