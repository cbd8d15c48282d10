"""Seeded synthetic command streams of SURVEY.md section 8(d), as backend-agnostic op scripts.

A script is a list of ops; `run_script(backend, script)` applies it to either the oracle
(`oracle.pyoracle.System`) or the GPU (`frankenpaxos_amd.Context`) -- both expose the same methods --
and returns every output, so parity is `assert_same(run_script(gpu), run_script(oracle))`.
"""
import numpy as np

MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64_at(x):
    """one splitmix64 output for each state in x (vectorised, wrap-around uint64 arithmetic)"""
    with np.errstate(over="ignore"):
        z = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def steady_values(slots):
    """value_id[i] = splitmix64(seed = 0xF9A405 + i) & 0x7fffffff   (steady stream)"""
    return (splitmix64_at(np.asarray(slots, dtype=np.uint64) + np.uint64(0xF9A405))
            & np.uint64(0x7FFFFFFF)).astype(np.int32)


def steady_stream(S, start=0):
    slot = np.arange(start, start + S, dtype=np.int32)
    return slot, np.zeros(S, np.int32), steady_values(slot)


class Rng:
    def __init__(self, seed):
        self.s = np.uint64(seed)

    def next(self):
        with np.errstate(over="ignore"):
            self.s = self.s + np.uint64(0x9E3779B97F4A7C15)
        z = self.s
        with np.errstate(over="ignore"):
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return int(z ^ (z >> np.uint64(31)))

    def below(self, n):
        return self.next() % n

    def chance(self, num, den):
        return self.below(den) < num

    def np_rng(self):
        return np.random.default_rng(self.next() & 0xFFFFFFFF)


def bits_from_bool(mat):
    """bool [n, total] -> uint64 [n, 4] little-endian bitmaps"""
    n, total = mat.shape
    padded = np.zeros((n, 256), np.uint8)
    padded[:, :total] = mat
    by = np.packbits(padded, axis=1, bitorder="little")  # [n, 32] bytes
    return by.view(np.uint64).reshape(n, 4).copy()


def bool_from_bits(bits, total):
    by = np.ascontiguousarray(bits, dtype=np.uint64).view(np.uint8).reshape(-1, 32)
    return np.unpackbits(by, axis=1, bitorder="little")[:, :total].astype(bool)


def random_subsets(nprng, n, total, lo, hi):
    """n random subsets of [0, total), sizes uniform in [lo, hi]"""
    sizes = nprng.integers(lo, hi + 1, size=n)
    keys = nprng.random((n, total))
    order = np.argsort(keys, axis=1)
    rank = np.empty_like(order)
    np.put_along_axis(rank, order, np.arange(total)[None, :].repeat(n, 0), axis=1)
    return rank < sizes[:, None]


def fast_subsets(nprng, n, total, lo, hi):
    """n pseudo-random subsets of [0, total) with sizes uniform in [lo, hi], in O(n * total) byte operations
    (random_subsets sorts n x total keys: minutes at 2^20 x 256).  total must be a power of two <= 256: member j
    of subset i is (a_i * j + b_i) mod total for j < size_i with a_i odd -- j -> a j + b is a permutation of
    Z/total, so the size is exact.  Used by the full-size parity streams; which acceptors a Phase2a reaches is
    the caller's choice in the reference too (an unseeded shuffle)."""
    assert total & (total - 1) == 0 and total <= 256
    sizes = nprng.integers(lo, hi + 1, size=n)
    a = (nprng.integers(0, total // 2 if total > 1 else 1, size=n) * 2 + 1).astype(np.int64)
    b = nprng.integers(0, total, size=n).astype(np.int64)
    # position of element e in subset i's order: j = a^-1 (e - b) mod total ; member iff j < size
    inv = np.array([pow(int(x), -1, total) if total > 1 else 0 for x in range(1, 2 * total, 2)], np.int64)
    ainv = inv[(a - 1) // 2 % len(inv)]
    e = np.arange(total, dtype=np.int64)[None, :]
    j = (ainv[:, None] * (e - b[:, None])) & (total - 1)
    return j < sizes[:, None]


def run_subsets(nprng, n, total, lo, hi):
    """n RUNS of neighbouring acceptors: positions [start, start + len) of the 256 a row has room for, cyclically, start a
    multiple of 16 (a 64-byte sector of the row), len uniform in [lo, min(hi, 128)]; positions >= total have no
    acceptor.  What a thrifty proxy leader sends when it rotates a window of f + 1 acceptors over the group instead of
    shuffling (any f + 1 will do: multipaxos/ProxyLeader.scala:190-191) -- the delivery k_phase2's packed walk takes."""
    start = 16 * nprng.integers(0, 16, size=n)
    length = nprng.integers(lo, min(hi, 128) + 1, size=n)
    j = np.arange(256)[None, :]
    return ((((j - start[:, None]) % 256) < length[:, None]) & (j < total))[:, :total]


def next_classic_round(n, leader, rnd):
    if rnd < 0:
        return leader
    m = n * (rnd // n)
    off = leader % n
    return m + off if m + off > rnd else m + n + off


def adversarial_script(S, R, q, seed, epochs=64, num_leaders=2, fused=True, ngroups=1,
                       group_of=None, subsets=None):
    """The parity / adversarial stream of SURVEY.md 8(d): delivery in `epochs` epochs of S/epochs
    slots; before an epoch, with probability 1/4, a leader change bumps the proposing round to
    nextClassicRound(leader = e % 2, round) and a random 25 % of the acceptors is pre-promised to the
    new round (Phase1a), so stale Phase2a's get Nacked; after a leader change 5 % of the already
    proposed slots are re-proposed in the new round with the same value; target_mask is a random
    subset of size U[q-8, R] (clamped to [1, R]); in the unfused pipeline 10 % of the vote messages
    are delivered twice.  Every epoch is one batch, so a batch carries one round per group."""
    subsets = subsets or random_subsets
    rng = Rng(seed)
    ops = []
    rnd = 0
    per = max(1, S // epochs)
    values = steady_values(np.arange(S))
    proposed = 0
    # the leader of round 0 runs Phase 1 first (all acceptors promise round 0)
    for g in range(ngroups):
        ops.append(("phase1a", g, 0, 0, None))
    e = 0
    pending_round = None
    while proposed < S:
        lo, hi = proposed, min(S, proposed + per)
        reprop = np.zeros(0, np.int64)
        if pending_round is not None:
            # the new leader finishes Phase 1 with everybody and takes over; it re-proposes 5 % of
            # the old slots in its round with the same value (Paxos-safe)
            rnd = pending_round
            pending_round = None
            nprng = rng.np_rng()
            for g in range(ngroups):
                ops.append(("phase1a", g, rnd, 0, None))
            if lo > 0:
                k = max(1, lo // 20)
                reprop = np.sort(nprng.choice(lo, size=k, replace=False))
        elif e > 0 and rng.chance(1, 4):
            # a competing leader pre-promises 25 % of the acceptors to its next round; the current
            # leader keeps proposing in the old round during this epoch => those acceptors Nack
            pending_round = next_classic_round(num_leaders, e % 2, rnd)
            if pending_round == rnd:
                pending_round = next_classic_round(num_leaders, e % 2, rnd + 1)
            nprng = rng.np_rng()
            pre = random_subsets(nprng, 1, R, max(1, R // 4), max(1, R // 4))
            for g in range(ngroups):
                ops.append(("phase1a", g, pending_round, 0, bits_from_bool(pre)[0]))
        nprng = rng.np_rng()
        slot = np.concatenate([reprop, np.arange(lo, hi)]).astype(np.int32)
        rr = np.full(len(slot), rnd, np.int32)
        val = values[slot]
        tgt = bits_from_bool(subsets(nprng, len(slot), R, max(1, q - 8), R))
        if fused:
            ops.append(("fused", slot, rr, val, tgt))
        else:
            dup = nprng.random(len(slot)) < 0.10
            ops.append(("k1k2", slot, rr, val, tgt, dup))
        proposed = hi
        e += 1
    return ops


def run_script(be, script):
    """apply a script to a backend; returns a list of (tag, arrays...) outputs"""
    out = []
    for op in script:
        kind = op[0]
        if kind == "phase1a":
            _, g, rnd, wm, tgt = op
            st, pb, nb = be.acceptor_phase1a(g, rnd, wm, tgt)
            out.append(("phase1a", st, pb, nb))
        elif kind == "fused":
            _, slot, rr, val, tgt = op
            out.append(("fused",) + tuple(be.phase2_fused(slot, rr, val, tgt)))
        elif kind == "k1k2":
            _, slot, rr, val, tgt, dup = op
            st0, new = be.proxy_open(slot, rr, val)
            st1, vb, nb, nr = be.acceptor_phase2a(slot, rr, val, tgt)
            st2, ch, cr, cv = be.proxy_phase2b(slot, rr, vb)
            # duplicated vote messages (10 %): delivered again, must change nothing
            idx = np.nonzero(dup)[0]
            st3, ch2, cr2, cv2 = be.proxy_phase2b(slot[idx], rr[idx], vb[idx])
            out.append(("k1k2", st0, new, st1, vb, nb, nr, st2, ch, cr, cv, st3, ch2, cr2, cv2))
        elif kind == "open":
            _, slot, rr, val = op
            out.append(("open",) + tuple(be.proxy_open(slot, rr, val)))
        elif kind == "phase2a":
            _, slot, rr, val, tgt = op
            out.append(("phase2a",) + tuple(be.acceptor_phase2a(slot, rr, val, tgt)))
        elif kind == "phase2b":
            _, slot, rr, vb = op
            out.append(("phase2b",) + tuple(be.proxy_phase2b(slot, rr, vb)))
        else:
            raise ValueError(kind)
    return out


def snapshot(be):
    vr, vv, bl = be.read_state()
    pr, mv = be.read_scalars()
    return {"vote_round": vr, "vote_value": vv, "ballot": bl, "promised": pr, "max_voted": mv}


def assert_same_outputs(a, b):
    assert len(a) == len(b)
    for k, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0]
        for j, (u, v) in enumerate(zip(x[1:], y[1:])):
            if isinstance(u, np.ndarray) or isinstance(v, np.ndarray):
                np.testing.assert_array_equal(np.asarray(u), np.asarray(v),
                                              err_msg="op %d (%s) output %d" % (k, x[0], j))
            else:
                assert u == v, "op %d (%s) output %d: %r != %r" % (k, x[0], j, u, v)


def assert_same_state(be_a, be_b, tally_slots=()):
    sa, sb = snapshot(be_a), snapshot(be_b)
    for key in sa:
        np.testing.assert_array_equal(sa[key], sb[key], err_msg=key)
    for s in tally_slots:
        assert be_a.read_tally(int(s)) == be_b.read_tally(int(s)), "tally of slot %d" % s


# ---- EPaxos: one tick of fresh instances (used by tests/test_epaxos.py and bench_configs.py) ----------------------
def random_tick(rng, n, num_keys, m, next_number, skew, fifo=True):
    """One tick of fresh instances.  Every replica sees the tick in roughly the global order, perturbed by a
    replica-specific skew.  A leader numbers its instances in the order IT processes them
    (Replica.scala nextAvailableInstance), and with fifo=True every other replica receives one leader's
    PreAccepts in sending order (the reference's transports are TCP channels: per-pair FIFO).  fifo=False
    lets the channels reorder, which is what makes a replica meet (L, 5) before (L, 4) -- the case
    dependencies.subtractOne(instance) (Replica.scala:582) exists for."""
    leader = rng.integers(0, n, m).astype(np.int32)
    key = rng.integers(0, num_keys, m).astype(np.int32)
    is_set = (rng.random(m) < 0.5).astype(np.uint8)  # Bernoulli get/set, J/Workload.scala:75-103
    mask = np.zeros(m, np.uint8)
    others = np.array([[r for r in range(n) if r != L] for L in range(n)])   # [n][n-1]
    drop = rng.integers(0, n - 1, m)
    for j in range(n - 1):
        mask |= np.where(drop != j, 1 << others[leader, j], 0).astype(np.uint8)
    rank = np.zeros((n, m), np.int32)
    for r in range(n):
        noisy = np.arange(m) + rng.normal(0, skew, m)
        rank[r, np.argsort(noisy, kind="stable")] = np.arange(m)
    number = np.zeros(m, np.int32)
    for L in range(n):
        idx = np.nonzero(leader == L)[0]
        by_own = idx[np.argsort(rank[L, idx], kind="stable")]     # the leader's own processing order
        number[by_own] = next_number[L] + np.arange(len(idx))
        next_number[L] += len(idx)
        if fifo:
            for r in range(n):
                if r != L:   # the positions leader L's messages occupy at r, refilled in sending order
                    rank[r, by_own] = np.sort(rank[r, idx])
    return leader, number, key, is_set, mask, rank
