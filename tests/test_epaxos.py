"""K5 (BASELINE.json configs[3]): EPaxos pre-accept fast path.
CPU: pins the oracle's TopOne and popularItems on the reference's known-answer tests
(shared/src/test/scala/util/TopOneTest.scala, shared/src/test/scala/UtilTest.scala:20-26) and checks a
hand-computed tick.  GPU: libfpx (fpx_epx_*) == oracle on random ticks."""
import numpy as np
import pytest


# ------------------------------------------------------------- golden vectors (CPU) ------------
def test_top_one_golden(oracle):
    """util/TopOneTest.scala:15-86"""
    t = oracle.top_one
    assert t(3) == [0, 0, 0]
    assert t(3, [(0, 0)]) == [1, 0, 0]
    assert t(3, [(0, 0), (1, 1), (2, 2)]) == [1, 2, 3]
    assert t(5, [(0, 0), (0, 1), (0, 2), (1, 1), (1, 10), (2, 2), (4, 3), (4, 1), (4, 7), (4, 1)]) == [3, 11, 3, 0, 8]
    assert t(3, [], merge_with=t(3)) == [0, 0, 0]
    assert t(3, [], merge_with=t(3, [(0, 0), (1, 1), (2, 2)])) == [1, 2, 3]
    assert t(3, [(0, 0), (1, 1), (2, 2)], merge_with=t(3)) == [1, 2, 3]
    assert t(3, [(0, 0), (1, 1), (2, 2)], merge_with=t(3, [(0, 0), (1, 10)])) == [1, 11, 3]


def test_popular_items_golden(oracle):
    """UtilTest.scala:20-26"""
    p = oracle.popular_items
    assert p([], 42) == set()
    assert p([1, 2, 3, 4], 2) == set()
    assert p([1, 2, 4, 3, 4], 2) == {4}
    assert p([4, 1, 2, 1, 3, 4], 2) == {1, 4}
    assert p([4, 1, 2, 1, 4, 3, 4], 2) == {1, 4}
    assert p([(0, 1), (0, 1), (1, 0)], 2) == {(0, 1)}


def test_key_value_store_top_one_conflict_index_golden(oracle):
    """statemachine/TopKConflictIndexTest.scala:281-329 ("get complicated conflicts correctly", k = 1).  The
    reference puts multi-key commands; a put of set(x, y) is the put of set x and of set y under the same
    instance (KeyValueStore.scala:232-253 loops over the keys), and a multi-key query is the merge of the
    single-key ones (:259-302), so the vector runs on the single-key index unchanged."""
    X, Y, Z = 0, 1, 2
    GET, SET = 0, 1
    e = oracle.EPaxos(3, 3)
    puts = [((0, 0), GET, [X]), ((1, 3), SET, [X, Y]), ((2, 20), GET, [Y, Z]), ((2, 10), GET, [Y, Z]),
            ((0, 1), GET, [X]), ((0, 3), GET, [X]), ((0, 2), SET, [X]), ((1, 1), SET, [X, Y]),
            ((2, 20), GET, [Y, Z])]
    for (leader, number), kind, keys in puts:
        for k in keys:
            e.index_put(0, k, kind, leader, number)
    assert e.index_conflicts(0, X, GET) == [3, 4, 0]
    assert e.index_conflicts(0, Y, GET) == [0, 4, 0]
    assert e.index_conflicts(0, Z, GET) == [0, 0, 0]
    assert e.index_conflicts(0, X, SET) == [4, 4, 0]
    assert e.index_conflicts(0, Y, SET) == [0, 4, 21]
    assert e.index_conflicts(0, Z, SET) == [0, 0, 21]
    merge = lambda *vs: [max(c) for c in zip(*vs)]
    assert merge(*(e.index_conflicts(0, k, GET) for k in (X, Y, Z))) == [3, 4, 0]    # get("x", "y", "z")
    assert merge(*(e.index_conflicts(0, k, SET) for k in (X, Y, Z))) == [4, 4, 21]   # set("x", "y", "z")
    assert e.index_conflicts(1, X, SET) == [0, 0, 0]                                 # another replica's index


# ------------------------------------------------------------------- a tick by hand (CPU) -------
def test_oracle_tick_by_hand(oracle):
    """n = 3 (f = 1): fast quorum n-1 = 2, the leader asks ONE other replica (n-2 = 1), so every
    fresh command takes the fast path and commits the union of its and the other's conflicts."""
    e = oracle.EPaxos(3, 4)
    #            leader number key is_set  asks
    msgs = [(0, 0, 1, 1, 0b010),   # A = set k1 by replica 0, asks replica 1
            (1, 0, 1, 0, 0b100),   # B = get k1 by replica 1, asks replica 2
            (2, 0, 1, 1, 0b001),   # C = set k1 by replica 2, asks replica 0
            (0, 1, 2, 0, 0b100)]   # D = get k2 by replica 0, asks replica 2 (no conflicts at all)
    leader, number, key, is_set, mask = (np.array(x) for x in zip(*msgs))
    # processing orders: replica 0: A, C, D ; replica 1: B, A ; replica 2: C, B, D
    rank = np.array([[0, 3, 1, 2],     # positions of A, B, C, D at replica 0 (B does not reach it)
                     [1, 0, 2, 3],     # replica 1: B first, then A
                     [3, 1, 0, 2]])    # replica 2: C, B, D
    st, fast, deps, ldeps, own = e.preaccept(leader, number, key, is_set, mask, rank)
    assert st == 0
    # A at its leader (replica 0, first): no conflicts -> [0,0,0]; at replica 1 after B (a get by 1,
    # instance 0): a set conflicts with gets -> [0,1,0]; one answer => fast, deps = [0,1,0]
    assert ldeps[0].tolist() == [0, 0, 0] and deps[0].tolist() == [0, 1, 0] and fast[0] == 1
    # B at replica 1 (first): nothing; at replica 2 after C (set by 2, instance 0): get vs set -> [0,0,1]
    assert ldeps[1].tolist() == [0, 0, 0] and deps[1].tolist() == [0, 0, 1]
    # C at replica 2 (first): nothing; at replica 0 after A (set by 0): [1,0,0]
    assert ldeps[2].tolist() == [0, 0, 0] and deps[2].tolist() == [1, 0, 0]
    assert ldeps[3].tolist() == [0, 0, 0] and deps[3].tolist() == [0, 0, 0]
    assert fast.tolist() == [1, 1, 1, 1]
    # after the tick every replica's index knows every instance (commit -> updateConflictIndex)
    for r in range(3):
        g, s = e.read_index(r, 1)
        assert g.tolist() == [0, 1, 0] and s.tolist() == [1, 0, 1]
        g, s = e.read_index(r, 2)
        assert g.tolist() == [2, 0, 0] and s.tolist() == [0, 0, 0]
    # next tick: a get of k1 by replica 1 conflicts with both sets everywhere: identical answers
    st, fast, deps, ldeps, own = e.preaccept([1], [1], [1], [0], [0b001], np.zeros((3, 1), np.int32))
    assert fast[0] == 1 and deps[0].tolist() == [1, 0, 1] and ldeps[0].tolist() == [1, 0, 1]


def test_oracle_slow_path_by_hand(oracle):
    """n = 5: the leader asks 3 others; two of them saw a conflicting set first, one did not =>
    the answers differ => slow path with the union (preAcceptingSlowPath)."""
    e = oracle.EPaxos(5, 2)
    # X = set k0 by replica 4 (instance 0), asks 1, 2, 3.   Y = set k0 by replica 0 (instance 7), asks 1, 2, 3
    leader, number, key, is_set = [4, 0], [0, 7], [0, 0], [1, 1]
    mask = [0b01110, 0b01110]
    # replica 1 and 2 process X then Y; replica 3 processes Y then X
    rank = np.array([[0, 1], [0, 1], [0, 1], [1, 0], [0, 1]])
    st, fast, deps, ldeps, own = e.preaccept(leader, number, key, is_set, mask, rank)
    assert st == 0
    # Y's answers: replicas 1, 2 saw X -> [0,0,0,0,1]; replica 3 did not -> [0,0,0,0,0]  => slow, union
    assert fast[1] == 0 and deps[1].tolist() == [0, 0, 0, 0, 1]
    # X's answers: replicas 1, 2 saw nothing -> [0..]; replica 3 saw Y (instance 7 of leader 0) -> [8,0,0,0,0]
    assert fast[0] == 0 and deps[0].tolist() == [8, 0, 0, 0, 0]
    assert e.preaccept([0], [0], [0], [1], [0b00110], np.zeros((5, 1), np.int32))[0] == 1  # 2 != n-2 others
    assert e.preaccept([0], [0], [0], [1], [0b00111], np.zeros((5, 1), np.int32))[0] == 1  # asks itself


def test_oracle_not_thrifty_by_hand(oracle):
    """ThriftySystem.NotThrifty (the reference's default, Replica.scala:83, 556-562): the PreAccept goes to
    EVERY other replica, each of them records the command in its conflict index, and the leader decides on
    the first n-2 answers.  n = 3: A = set k1 by replica 0, B = get k1 by replica 2; both are answered by
    replica 1; every replica processes A before B."""
    leader, number, key, is_set = [0, 2], [0, 0], [1, 1], [1, 0]
    resp, seen = [0b010, 0b010], [0b110, 0b011]
    rank = np.array([[0, 1], [0, 1], [0, 1]])
    e = oracle.EPaxos(3, 4)
    st, fast, deps, ldeps, own = e.preaccept(leader, number, key, is_set, resp, rank, seen_mask=seen)
    assert st == 0 and fast.tolist() == [1, 1]
    # A reached replica 2 although its answer is not waited for: B's own leader already conflicts with it
    assert ldeps[1].tolist() == [1, 0, 0] and deps[1].tolist() == [1, 0, 0]
    assert ldeps[0].tolist() == [0, 0, 0] and deps[0].tolist() == [0, 0, 0]
    # thrifty: replica 2 never sees A's PreAccept, B's PreAccept carries no dependency; replica 1 adds it
    e = oracle.EPaxos(3, 4)
    st, fast, deps, ldeps, own = e.preaccept(leader, number, key, is_set, resp, rank)
    assert ldeps[1].tolist() == [0, 0, 0] and deps[1].tolist() == [1, 0, 0]
    # the answers counted must come from replicas the PreAccept was sent to; never from the leader itself
    assert e.preaccept([0], [5], [0], [1], [0b010], np.zeros((3, 1), np.int32), seen_mask=[0b100])[0] == 1
    assert e.preaccept([0], [5], [0], [1], [0b010], np.zeros((3, 1), np.int32), seen_mask=[0b011])[0] == 1


# ----------------------------------------------------------------------------- GPU parity -------
from tests.workloads import random_tick  # noqa: E402,F401 -- shared with bench_configs.py


# ------------------------------------------------ subtractOne: the instance is never its own dependency ----
def _ips(oracle, fn, *args):
    import ctypes as C
    L = oracle.lib()
    out = np.zeros(4096, np.int32)
    wm = C.c_int(0)
    arrs = []
    cargs = []
    for a in args:
        if isinstance(a, (list, tuple, set)):
            v = np.array(sorted(a), dtype=np.int32)
            arrs.append(v)
            cargs += [v.ctypes.data_as(C.POINTER(C.c_int)), len(v)]
        else:
            cargs.append(int(a))
    f = getattr(L, fn)
    f.restype = C.c_int
    f.argtypes = None
    n = f(*cargs, out.ctypes.data_as(C.POINTER(C.c_int)), C.byref(wm))
    return wm.value, out[:n].tolist()


def _materialize(wm, values):
    return set(range(wm)) | set(values)


def test_int_prefix_set_restatement_is_pinned_like_the_reference(oracle):
    """compact/IntPrefixSetTest.scala:231-239 ("addAll random sets correctly") and :251-263 ("subtractOne
    random sets correctly"): the reference pins these operations by materialize() == the plain set operation
    on random sets of 0..100; the oracle's IntPrefixSet restatement is held to the same property, and to the
    canonical (watermark, values) form equality depends on (IntPrefixSet.scala:214-221, compact() :426-431)."""
    rng = np.random.default_rng(2024)
    for trial in range(600):
        a = set(rng.integers(0, 100, rng.integers(0, 60)).tolist())
        b = set(rng.integers(0, 100, rng.integers(0, 60)).tolist())
        x = int(rng.integers(0, 101))
        wm, vals = _ips(oracle, "fpo_ips_subtract_one", 0, a, x)
        assert _materialize(wm, vals) == a - {x}
        assert all(v > wm for v in vals) and wm not in vals          # compacted
        wm, vals = _ips(oracle, "fpo_ips_add_all", 0, a, 0, b)
        assert _materialize(wm, vals) == a | b
        assert all(v > wm for v in vals)
    # the shapes this path produces: a bare watermark minus one id
    assert _ips(oracle, "fpo_ips_subtract_one", 6, [], 4) == (4, [5])       # {0..5} - 4 = {0..3, 5}
    assert _ips(oracle, "fpo_ips_subtract_one", 6, [], 6) == (6, [])        # x >= watermark: values -= x
    assert _ips(oracle, "fpo_ips_subtract_one", 6, [], 5) == (5, [])
    assert _ips(oracle, "fpo_ips_subtract_one", 9, [], 2) == (2, [3, 4, 5, 6, 7, 8])
    assert _ips(oracle, "fpo_ips_add_all", 4, [5], 4, [5, 6]) == (4, [5, 6])
    assert _ips(oracle, "fpo_ips_add_all", 4, [5], 5, []) == (6, [])        # union {0..3,5} + {0..4} compacts


def test_oracle_out_of_order_instances_of_one_leader_by_hand(oracle):
    """VERDICT r01 weak #2.  n = 3; replica 0 leads instances (0, 4) and (0, 5), both sets of key 1, and asks
    replica 1 (n - 2 = 1 other).  Replica 1 processes (0, 5) BEFORE (0, 4).
      at the leader (in order): (0,4) sees nothing -> {<0} ; (0,5) sees (0,4) -> column 0 = {<5}, minus itself
        (5 >= watermark 5: values -= 5, IntPrefixSet.scala:389-390) -> {<5}
      at replica 1: (0,5) first: nothing -> {<0}; then (0,4): the index holds (0,5) => TopOne column 0 = 6,
        fromWatermarks -> {<6}; subtractOne(4): 4 < 6 => values += 5, watermark = 4 (:391-396) -> {<4, 5},
        i.e. {0..3, 5}: NOT a dependency on itself.
      PreAcceptOk((0,4)) = {<4,5} U leader's {<0} = {<4, 5}; one answer, n = 3 => fast path with it."""
    e = oracle.EPaxos(3, 4)
    leader, number, key, is_set = [0, 0], [4, 5], [1, 1], [1, 1]
    mask = [0b010, 0b010]
    rank = np.array([[0, 1], [1, 0], [0, 1]])       # replica 1: message 1 = (0,5) first
    st, fast, deps, ldeps, own = e.preaccept(leader, number, key, is_set, mask, rank)
    assert st == 0 and fast.tolist() == [1, 1]
    assert ldeps.tolist() == [[0, 0, 0], [5, 0, 0]] and own[:, 1].tolist() == [0, 0]
    assert deps[0].tolist() == [4, 0, 0] and own[0, 0] == 6      # (0,4): {<4, 5}
    assert deps[1].tolist() == [5, 0, 0] and own[1, 0] == 0      # (0,5): {<5}
    # n = 5: two responders disagree about the hole => slow path, and the union keeps the hole
    e = oracle.EPaxos(5, 4)
    mask = [0b01110, 0b01110]
    rank = np.array([[0, 1], [1, 0], [0, 1], [0, 1], [0, 1]])   # only replica 1 sees (0,5) first
    st, fast, deps, ldeps, own = e.preaccept(leader, number, key, is_set, mask, rank)
    assert st == 0 and fast.tolist() == [0, 1]
    assert deps[0].tolist() == [4, 0, 0, 0, 0] and own[0, 0] == 6   # {<0} U {<0} U {<4,5} = {<4, 5}
    assert deps[1].tolist() == [5, 0, 0, 0, 0] and own[1, 0] == 0
    # the leader's own conflicts can carry the hole too (across ticks: its index already holds (0, 9))
    st, fast, deps, ldeps, own = e.preaccept([0], [9], [1], [1], [0b01110], np.zeros((5, 1), np.int32))
    st, fast, deps, ldeps, own = e.preaccept([0], [7], [1], [1], [0b01110], np.zeros((5, 1), np.int32))
    assert st == 0 and fast[0] == 1 and ldeps[0].tolist() == [7, 0, 0, 0, 0] and own[0].tolist() == [10, 10]


def test_random_tick_respects_channel_fifo():
    rng = np.random.default_rng(1)
    nxt = [0] * 5
    for fifo in (True, False):
        leader, number, key, is_set, mask, rank = random_tick(rng, 5, 16, 3000, nxt, 20.0, fifo=fifo)
        for r in range(5):
            assert (np.sort(rank[r]) == np.arange(3000)).all()
        inversions = 0
        for L in range(5):
            idx = np.nonzero(leader == L)[0]
            idx = idx[np.argsort(number[idx])]
            assert (np.diff(rank[L, idx]) > 0).all()                # a leader processes its own in number order
            for r in range(5):
                inversions += int((np.diff(rank[r, idx]) < 0).sum())
        assert (inversions == 0) == fifo
        assert (np.array([bin(x).count("1") for x in mask]) == 3).all() and not ((mask >> leader) & 1).any()


@pytest.mark.gpu
@pytest.mark.parametrize("n,num_keys,m,skew", [(3, 8, 500, 3.0), (5, 16, 4000, 5.0), (5, 1024, 20000, 50.0),
                                               (7, 64, 3000, 10.0), (5, 1, 300, 2.0),
                                               # sort shapes: whole tiles / 2 digit passes, 3 digit passes,
                                               # 255 keys + the non-participant bucket = exactly one digit
                                               (3, 256, 2048, 4.0), (5, 70000, 5000, 20.0), (3, 255, 1025, 7.0)])
@pytest.mark.parametrize("fifo", [True, False])
def test_epaxos_ticks_match_oracle(oracle, n, num_keys, m, skew, fifo):
    """fifo=False: channels reorder one leader's PreAccepts, so replicas meet higher-numbered instances of a
    leader first and dependencies.subtractOne(instance) leaves explicit values (own_values_end != 0)"""
    from frankenpaxos_amd.epaxos import EPaxos

    gpu, ref = EPaxos(n, num_keys), oracle.EPaxos(n, num_keys)
    rng = np.random.default_rng(n * 1000 + num_keys)
    nxt = [0] * n
    seen_fast = seen_slow = holes = 0
    for tick in range(4):
        args = random_tick(rng, n, num_keys, m, nxt, skew, fifo=fifo)
        a, b = gpu.preaccept(*args), ref.preaccept(*args)
        assert a[0] == b[0] == 0
        for x, y in zip(a[1:], b[1:]):
            np.testing.assert_array_equal(x, y)
        seen_fast += int(a[1].sum())
        seen_slow += int((a[1] == 0).sum())
        holes += int((a[4][:, 0] != 0).sum())
    assert (holes == 0) == (fifo or num_keys > m)
    for r in range(n):
        for k in range(0, num_keys, max(1, num_keys // 16)):
            ga, sa = gpu.read_index(r, k)
            gb, sb = ref.read_index(r, k)
            assert ga.tolist() == gb.tolist() and sa.tolist() == sb.tolist()
    assert seen_fast > 0 and (seen_slow > 0 or n == 3)


@pytest.mark.gpu
def test_config4_epaxos_1m_commands_5_replicas(oracle):
    """BASELINE.json configs[3] at its stated size: n = 5, 2^20 commands per tick, keys
    splitmix64(i) % 1024, Bernoulli(1/2) get/set (J/Workload.scala:75-103) -- GPU == oracle bit for bit on
    every output of two consecutive ticks and on every replica's whole conflict index; the second tick's
    channels reorder, so the own-column holes of subtractOne are exercised at size too."""
    from frankenpaxos_amd.epaxos import EPaxos
    from tests import workloads as W

    n, num_keys, m = 5, 1024, 1 << 20
    gpu, ref = EPaxos(n, num_keys), oracle.EPaxos(n, num_keys)
    rng = np.random.default_rng(4)
    nxt = [0] * n
    for tick, fifo in enumerate((True, False)):
        leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 64.0, fifo=fifo)
        key = (W.splitmix64_at(np.arange(tick * m, (tick + 1) * m, dtype=np.uint64)) % np.uint64(num_keys)).astype(np.int32)
        a = gpu.preaccept(leader, number, key, is_set, mask, rank)
        b = ref.preaccept(leader, number, key, is_set, mask, rank)
        assert a[0] == b[0] == 0
        for x, y in zip(a[1:], b[1:]):
            np.testing.assert_array_equal(x, y)
        assert 0 < int(a[1].sum()) < m                      # both paths taken
        assert (int((a[4] != 0).sum()) > 0) == (not fifo)
    for r in range(n):
        for k in range(num_keys):
            ga, sa = gpu.read_index(r, k)
            gb, sb = ref.read_index(r, k)
            assert ga.tolist() == gb.tolist() and sa.tolist() == sb.tolist()


@pytest.mark.gpu
def test_epaxos_invalid_ticks(oracle):
    from frankenpaxos_amd.epaxos import EPaxos
    import frankenpaxos_amd as fa

    gpu = EPaxos(5, 4)
    z = np.zeros((5, 1), np.int32)
    assert gpu.preaccept([0], [0], [0], [1], [0b00110], z)[0] == fa.FPX_EINVAL   # not n-2 others
    assert gpu.preaccept([0], [0], [0], [1], [0b00111], z)[0] == fa.FPX_EINVAL   # asks itself
    assert gpu.preaccept([0], [0], [9], [1], [0b01110], z)[0] == fa.FPX_EINVAL   # key out of range
    st, fast, deps, ldeps, own = gpu.preaccept([0], [0], [0], [1], [0b01110], z)
    assert st == 0 and fast[0] == 1 and deps[0].tolist() == [0] * 5 and own.tolist() == [[0, 0]]
    with pytest.raises(fa.FpxError):
        EPaxos(4, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("n,num_keys,m", [(3, 8, 700), (5, 64, 5000), (7, 16, 3000), (5, 64, 4999), (3, 8, 2049)])  # (m % 4 != 0: the partition's element-wise loads)
def test_epaxos_not_thrifty_ticks_match_oracle(oracle, n, num_keys, m):
    """seen_mask: the PreAccept reaches every other replica (ThriftySystem.NotThrifty, the reference's
    default) or a random superset of the n-2 that are waited for"""
    from frankenpaxos_amd.epaxos import EPaxos
    import frankenpaxos_amd as fa

    gpu, ref = EPaxos(n, num_keys), oracle.EPaxos(n, num_keys)
    rng = np.random.default_rng(n * 77 + num_keys)
    nxt = [0] * n
    for tick in range(4):
        leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 8.0)
        everyone = ((1 << n) - 1) & ~(1 << leader.astype(np.int64))
        seen = np.where(rng.random(m) < 0.7, everyone, mask).astype(np.uint8)   # mixed deployments in one tick
        a = gpu.preaccept(leader, number, key, is_set, mask, rank, seen_mask=seen)
        b = ref.preaccept(leader, number, key, is_set, mask, rank, seen_mask=seen)
        assert a[0] == b[0] == 0
        for x, y in zip(a[1:], b[1:]):
            np.testing.assert_array_equal(x, y)
    for r in range(n):
        for k in range(num_keys):
            ga, sa = gpu.read_index(r, k)
            gb, sb = ref.read_index(r, k)
            assert ga.tolist() == gb.tolist() and sa.tolist() == sb.tolist()
    z = np.zeros((n, 1), np.int32)
    others = [r for r in range(1, n)]
    resp = sum(1 << r for r in others[: n - 2])
    assert gpu.preaccept([0], [0], [0], [1], [resp], z, seen_mask=[resp | 1])[0] == fa.FPX_EINVAL       # the leader
    assert gpu.preaccept([0], [0], [0], [1], [resp], z, seen_mask=[resp & (resp - 1)])[0] == fa.FPX_EINVAL  # resp not in seen


def test_oracle_rejects_a_rank_that_is_not_a_permutation(oracle):
    e = oracle.EPaxos(3, 4)
    rank = np.array([[0, 1], [0, 0], [0, 1]])          # replica 1: two messages in position 0
    assert e.preaccept([0, 1], [0, 0], [1, 1], [1, 1], [0b010, 0b100], rank)[0] == 1
    assert e.read_index(0, 1)[1].tolist() == [0, 0, 0]  # nothing applied


@pytest.mark.gpu
def test_epaxos_rank_must_be_a_permutation_and_tick_sizes_may_vary(oracle):
    """a malformed delivery order is FPX_EINVAL with nothing applied (positions nobody was scattered to would
    otherwise be read with an older tick's contents); ticks of different sizes reuse the same buffers"""
    from frankenpaxos_amd.epaxos import EPaxos
    import frankenpaxos_amd as fa

    n, K = 5, 32
    gpu, ref = EPaxos(n, K), oracle.EPaxos(n, K)
    rng = np.random.default_rng(3)
    nxt = [0] * n
    for m in (5000, 300, 2600, 64, 1, 5000, 1025):
        args = random_tick(rng, n, K, m, nxt, 6.0)
        a, b = gpu.preaccept(*args), ref.preaccept(*args)
        assert a[0] == b[0] == 0
        for x, y in zip(a[1:], b[1:]):
            np.testing.assert_array_equal(x, y)
        if m >= 64:
            # break the permutation of one replica: two messages share a position, one position is empty
            leader, number, key, is_set, mask, rank = random_tick(rng, n, K, m, list(nxt), 6.0)
            bad = rank.copy()
            bad[2, int(rng.integers(0, m))] = bad[2, int(rng.integers(0, m - 1)) + 1 if m > 1 else 0]
            if not (np.sort(bad[2]) == np.arange(m)).all():
                assert gpu.preaccept(leader, number, key, is_set, mask, bad)[0] == fa.FPX_EINVAL
                assert ref.preaccept(leader, number, key, is_set, mask, bad)[0] == 1
    for r in range(n):
        for k in range(K):
            ga, sa = gpu.read_index(r, k)
            gb, sb = ref.read_index(r, k)
            assert ga.tolist() == gb.tolist() and sa.tolist() == sb.tolist()


@pytest.mark.parametrize("n,not_thrifty", [(3, False), (5, False), (5, True), (7, False), (7, True)])
def test_oracle_epaxos_safety_invariant(oracle, n, not_thrifty):
    """EPaxos' dependency invariant, the property its execution order rests on: of two conflicting commands
    (same key, at least one of them a set) at least one has the other in its committed dependencies -- fast
    quorums intersect, and the replica in the intersection saw one of the two first.  Checked on the oracle
    (the reference's own EPaxos tests are randomized simulations of the same property,
    shared/src/test/scala/epaxos/EPaxosTest.scala), within a tick under skewed delivery orders and across
    ticks; dependencies are TopOne watermarks: deps[leader] > number means "depends on (leader, number)"."""
    num_keys, m = 6, 260
    e = oracle.EPaxos(n, num_keys)
    rng = np.random.default_rng(100 * n + int(not_thrifty))
    nxt = [0] * n
    history = []  # (leader, number, key, is_set, deps)
    pairs = 0
    for tick in range(4):
        leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 25.0)
        seen = None
        if not_thrifty:
            seen = (((1 << n) - 1) & ~(1 << leader.astype(np.int64))).astype(np.uint8)
        st, fast, deps, ldeps, own = e.preaccept(leader, number, key, is_set, mask, rank, seen_mask=seen)
        assert st == 0
        cur = [(int(leader[i]), int(number[i]), int(key[i]), int(is_set[i]), deps[i]) for i in range(m)]
        for i, (li, ni, ki, si, di) in enumerate(cur):
            # every conflicting command of an EARLIER tick is committed everywhere: i must depend on it
            for (lj, nj, kj, sj, dj) in history:
                if ki == kj and (si or sj):
                    assert di[lj] > nj, (tick, i, (lj, nj))
            # within the tick: one of the two depends on the other
            for (lj, nj, kj, sj, dj) in cur[:i]:
                if ki == kj and (si or sj):
                    pairs += 1
                    assert di[lj] > nj or dj[li] > ni, (tick, (li, ni), (lj, nj))
        history.extend(cur)
    assert pairs > 1000


def test_oracle_epaxos_safety_invariant_negative_control(oracle):
    """the invariant test is not vacuous: the PreAccept's own dependencies (the leader's view alone, before
    the fast quorum answered) do violate it under skewed delivery"""
    n, num_keys, m = 5, 6, 260
    e = oracle.EPaxos(n, num_keys)
    rng = np.random.default_rng(9)
    leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, [0] * n, 25.0)
    st, fast, deps, ldeps, own = e.preaccept(leader, number, key, is_set, mask, rank)
    violations = 0
    for i in range(m):
        for j in range(i):
            if key[i] == key[j] and (is_set[i] or is_set[j]):
                if not (ldeps[i][leader[j]] > number[j] or ldeps[j][leader[i]] > number[i]):
                    violations += 1
                assert deps[i][leader[j]] > number[j] or deps[j][leader[i]] > number[i]
    assert violations > 0


# ------------------------------------------------ beyond fresh instances: command log, Prepare, Accept ------------
def enc(ordering, replica):
    return ordering * 8 + replica


def test_oracle_accept_phase_by_hand(oracle):
    """n = 5 (f = 2: slow quorum 3, the proposer included).  X = set k0 by replica 4 (instance (4,0)), Y = set k0 by
    replica 0 (instance (0,7)); both pre-accept on the slow path (test_oracle_slow_path_by_hand), then:
      the command log after the pre-accept: PreAcceptedEntry(Ballot(0, leader), Ballot(0, leader), triple) at the
        leader and its three responders (Replica.scala:688-696, 1259-1271), nothing at the fifth replica;
      Accept X by 4 in Ballot(0,4) to {1, 2}: 0 < entry ballot? no -> AcceptedEntry, AcceptOk; with 4's own that is
        3 = f + 1 responses -> commit: CommittedEntry at ALL five replicas (handleAcceptOk :1557-1563, commit);
      replica 2 starts recovering Y: Prepare(Y, Ballot(1,2)) to {1, 3}: PreAcceptedEntry -> PrepareOk(PreAccepted,
        voteBallot (0,0), triple), the entries' ballot moves to (1,2), largestBallot of 1 and 3 too (:1637, 1711-1724);
      the old leader 0 still sends Accept(Y, Ballot(0,0)) to {1, 3}: (0,0) < (1,2) -> Nack(largestBallot = (1,2))
        from both (:1432-1439); 0's own AcceptOk alone is no quorum;
      replica 2 finishes: Accept(Y, Ballot(1,2)) to {1, 3}: equal ballot is not smaller -> accepted; 3 responses ->
        committed everywhere, replica 0 included;
      0 tries once more: its own entry is Committed -> logger.fatal in transitionToAcceptPhase (:740-744)."""
    e = oracle.EPaxos(5, 2, num_instances=16)
    leader, number, key, is_set = [4, 0], [0, 7], [0, 0], [1, 1]
    rank = np.array([[0, 1], [0, 1], [0, 1], [1, 0], [0, 1]])
    st, fast, deps, ldeps, own = e.preaccept(leader, number, key, is_set, [0b01110, 0b01110], rank, triple_id=[100, 107])
    assert st == 0 and fast.tolist() == [0, 0]
    for r, want in ((4, (2, enc(0, 4), enc(0, 4), 100)), (1, (2, enc(0, 4), enc(0, 4), 100)), (0, (0, -1, -1, -1))):
        assert e.read_cmdlog(r, 4, 0)[:4] == want
    assert e.read_cmdlog(0, 0, 7)[:4] == (2, enc(0, 0), enc(0, 0), 107) and e.read_cmdlog(4, 0, 7)[0] == 0
    st, ok, nack, com, nb, done = e.accept([4], [0], [0], [4], [100], [0b00110])
    assert st == 0 and ok[0] == 0b10110 and nack[0] == 0 and done[0] == 1
    assert all(e.read_cmdlog(r, 4, 0)[:4] == (4, -1, -1, 100) for r in range(5))
    st, ok, nack, com, nb, rs, rv, rt = e.prepare([0], [7], [1], [2], [0b01010])
    assert st == 0 and ok[0] == 0b01010 and nack[0] == 0
    assert rs[0].tolist() == [-1, 2, -1, 2, -1] and rv[0].tolist() == [-1, 0, -1, 0, -1] and rt[0].tolist() == [-1, 107, -1, 107, -1]
    assert e.read_cmdlog(1, 0, 7) == (2, enc(1, 2), enc(0, 0), 107, enc(1, 2))
    st, ok, nack, com, nb, done = e.accept([0], [7], [0], [0], [107], [0b01010])
    assert st == 0 and ok[0] == 0b00001 and nack[0] == 0b01010 and nb[0] == enc(1, 2) and done[0] == 0
    assert e.read_cmdlog(0, 0, 7)[:4] == (3, enc(0, 0), enc(0, 0), 107)
    st, ok, nack, com, nb, done = e.accept([0], [7], [1], [2], [107], [0b01010])
    assert st == 0 and ok[0] == 0b01110 and nack[0] == 0 and done[0] == 1
    assert all(e.read_cmdlog(r, 0, 7)[:4] == (4, -1, -1, 107) for r in range(5))
    st, ok, nack, com, nb, done = e.accept([0], [7], [0], [0], [107], [0b01010])
    assert st == 9 and done[0] == 0                      # FPX_EFATAL_PROTOCOL: the proposer holds a CommittedEntry
    # a Prepare and an Accept that reach committed replicas are answered with the Commit (:1744-1755, :1463-1474)
    st, ok, nack, com, nb, rs, rv, rt = e.prepare([0], [7], [2], [3], [0b10111])
    assert com[0] == 0b10111 and ok[0] == 0 and e.read_cmdlog(4, 0, 7)[4] == enc(2, 3)   # largestBallot moved all the same
    # an instance nobody has heard of: Prepare leaves NoCommandEntry(ballot), a smaller Accept is then Nacked,
    # an equal one accepted; re-sending the Accept is answered again without changing anything (:1451-1461)
    st, ok, nack, com, nb, rs, rv, rt = e.prepare([3], [5], [4], [1], [0b00101])
    assert ok[0] == 0b00101 and rs[0].tolist() == [0, -1, 0, -1, -1] and e.read_cmdlog(2, 3, 5)[:4] == (1, enc(4, 1), -1, -1)
    st, ok, nack, com, nb, done = e.accept([3], [5], [3], [3], [55], [0b00101])
    assert ok[0] == 0b01000 and nack[0] == 0b00101 and nb[0] == enc(4, 1)
    st, ok, nack, com, nb, done = e.accept([3], [5], [4], [1], [56], [0b00101])
    assert ok[0] == 0b00111 and done[0] == 1
    # the instances of one call must be distinct; the proposer is never among its own targets
    assert e.accept([1, 1], [2, 2], [0, 0], [1, 1], [1, 1], [0b00101, 0b00101])[0] == 1
    assert e.accept([1], [3], [0], [1], [1], [0b00011])[0] == 1
    assert oracle.EPaxos(5, 2).accept([1], [3], [0], [1], [1], [0b00101])[0] == 1   # no command log configured


def _accept_teaches_the_conflict_index(e):
    """updateConflictIndex wherever a triple is stored (Replica.scala:602-614).  n = 5, key 1.
      X = (0, 0) = set k1 reaches replicas 1 and 2 as a PreAccept (they index it, :1279); replicas 3 and 4 have never
        heard of X;
      Accept(X, Ballot(0,0), set k1) by 0 to {3}: the proposer indexes X (:763), replica 3 takes the Accept in and
        indexes X (:1503); two AcceptOks are no quorum.  Replica 4 still knows nothing;
      PreAccept(Y = (4, 0), set k1) at 3 and at 4: replica 3 -- which only ever saw X in an Accept -- answers with the
        dependency X; replica 4 with none;
      the Accept again to {1, 3}: 3 answers again, 1 accepts -> f + 1 = 3 AcceptOks -> commit at EVERY replica (:828):
        replica 4 learns X from the Commit;
      PreAccept(Z = (4, 1), get k1) at 4: depends on X now.  A Noop triple (key -1) teaches nothing."""
    zeros = np.zeros((1, 5), np.int32)
    X = [1, 0, 0, 0, 0]
    st, ok, resend, nack, com, nb, rd, re, rt = e.handle_preaccept([0], [0], [0], [0], [1], [1], [7], zeros, None, [0b00110])
    assert st == 0 and ok[0] == 0b00110
    for r, want in ((0, [0] * 5), (1, X), (2, X), (3, [0] * 5), (4, [0] * 5)):
        assert e.read_index(r, 1)[1].tolist() == want
    st, ok, nack, com, nb, done = e.accept([0], [0], [0], [0], [7], [0b01000], key=[1], is_set=[1])
    assert st == 0 and ok[0] == 0b01001 and done[0] == 0
    for r, want in ((0, X), (3, X), (4, [0] * 5)):
        assert e.read_index(r, 1)[1].tolist() == want and e.read_index(r, 1)[0].tolist() == [0] * 5
    st, ok, resend, nack, com, nb, rd, re, rt = e.handle_preaccept([4], [0], [0], [4], [1], [1], [8], zeros, None, [0b11000])
    assert st == 0 and ok[0] == 0b11000
    assert rd[0][3].tolist() == X and rd[0][4].tolist() == [0] * 5
    st, ok, nack, com, nb, done = e.accept([0], [0], [0], [0], [7], [0b01010], key=[1], is_set=[1])
    assert st == 0 and ok[0] == 0b01011 and done[0] == 1
    for r in range(5):
        assert e.read_index(r, 1)[1].tolist() == [1, 0, 0, 0, 1 if r >= 3 else 0]   # X everywhere; Y where it was pre-accepted
    st, ok, resend, nack, com, nb, rd, re, rt = e.handle_preaccept([4], [1], [0], [4], [1], [0], [9], zeros, None, [0b10000])
    assert st == 0 and ok[0] == 0b10000 and rd[0][4].tolist() == [1, 0, 0, 0, 1]
    # a Noop triple: the entry is stored, the index is left alone
    before = [e.read_index(r, k) for r in range(5) for k in range(4)]
    st, ok, nack, com, nb, done = e.accept([2], [3], [0], [2], [10], [0b01010], key=[-1], is_set=[0])
    assert st == 0 and done[0] == 1 and e.read_cmdlog(4, 2, 3)[0] == 4
    after = [e.read_index(r, k) for r in range(5) for k in range(4)]
    assert all(a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist() for a, b in zip(before, after))
    # a key outside the store is a require failure, nothing applied
    assert e.accept([2], [4], [0], [2], [11], [0b01010], key=[4], is_set=[0])[0] == 1
    assert e.read_cmdlog(2, 2, 4)[0] == 0


def test_oracle_accept_teaches_the_conflict_index_by_hand(oracle):
    _accept_teaches_the_conflict_index(oracle.EPaxos(5, 4, num_instances=16))


@pytest.mark.gpu
def test_epaxos_accept_teaches_the_conflict_index(oracle):
    """the same by-hand trace on the GPU, and random Accept batches with commands: every replica's whole index"""
    from frankenpaxos_amd.epaxos import EPaxos

    _accept_teaches_the_conflict_index(EPaxos(5, 4, num_instances=16))
    n, NI, K = 5, 256, 6
    gpu, ref = EPaxos(n, K, num_instances=NI), oracle.EPaxos(n, K, num_instances=NI)
    rng = np.random.default_rng(77)
    nxt = [0] * n
    for step in range(8):
        leader, number, b_ord, b_rep, tgt = _cl_batch(rng, n, NI, 300, nxt)
        tgt = (tgt & ~(1 << b_rep)).astype(np.uint8)
        tr = rng.integers(0, 1 << 20, len(leader)).astype(np.int32)
        key = rng.integers(-1, K, len(leader)).astype(np.int32)
        is_set = rng.integers(0, 2, len(leader)).astype(np.uint8)
        _same(gpu.accept(leader, number, b_ord, b_rep, tr, tgt, key, is_set),
              ref.accept(leader, number, b_ord, b_rep, tr, tgt, key, is_set))
        for r in range(n):
            for k in range(K):
                ga, sa = gpu.read_index(r, k)
                gb, sb = ref.read_index(r, k)
                assert ga.tolist() == gb.tolist() and sa.tolist() == sb.tolist()
    assert any(ref.read_index(r, k)[1].max() > 0 for r in range(n) for k in range(K))


def _cl_batch(rng, n, NI, m, nxt):
    """m distinct instances, random ballots and target sets.  Instances come from the upper half of every leader's
    numbers and from those the pre-accept ticks have already used (nxt): the numbers the NEXT pre-accept tick will
    take stay untouched, so that its instances are fresh"""
    pool = np.array([L * NI + x for L in range(n) for x in list(range(nxt[L])) + list(range(NI // 2, NI))])
    m = min(m, len(pool))
    inst = rng.choice(pool, size=m, replace=False)
    leader, number = (inst // NI).astype(np.int32), (inst % NI).astype(np.int32)
    b_ord = rng.integers(0, 4, m).astype(np.int32)
    b_rep = rng.integers(0, n, m).astype(np.int32)
    tgt = rng.integers(0, 1 << n, m).astype(np.uint8)
    return leader, number, b_ord, b_rep, tgt


@pytest.mark.gpu
@pytest.mark.parametrize("n,NI,m", [(3, 64, 40), (5, 512, 700), (7, 300, 1500), (5, 4096, 5000)])
def test_epaxos_prepare_accept_match_oracle(oracle, n, NI, m):
    """random Prepare / Accept batches on the command log (ballots going up and down, arbitrary target sets, instances
    in every state incl. committed, re-sent Accepts, proposers that must refuse), interleaved with pre-accept ticks
    that create PreAccepted / Committed entries: every reply and the whole command log, GPU == oracle"""
    from frankenpaxos_amd.epaxos import EPaxos
    import frankenpaxos_amd as fa

    gpu, ref = EPaxos(n, 8, num_instances=NI), oracle.EPaxos(n, 8, num_instances=NI)
    rng = np.random.default_rng(n * 31 + NI)
    nxt = [0] * n
    fatal = nacks = commits = 0
    for step in range(14):
        kind = step % 4
        if kind == 0 and max(nxt) + min(m, 200) // 2 + 8 < NI // 2:
            mm = min(m, 200)
            leader, number, key, is_set, mask, rank = random_tick(rng, n, 8, mm, nxt, 3.0)
            tr = rng.integers(0, 1 << 20, mm).astype(np.int32)
            a = gpu.preaccept(leader, number, key, is_set, mask, rank, triple_id=tr)
            b = ref.preaccept(leader, number, key, is_set, mask, rank, triple_id=tr)
            assert a[0] == b[0] == 0
            for x, y in zip(a[1:], b[1:]):
                np.testing.assert_array_equal(x, y)
            # the same tick again: the instances are no longer fresh -> FPX_EINVAL on both, nothing applied
            assert gpu.preaccept(leader, number, key, is_set, mask, rank)[0] == fa.FPX_EINVAL
            assert ref.preaccept(leader, number, key, is_set, mask, rank)[0] == 1
        elif kind in (1, 3):
            leader, number, b_ord, b_rep, tgt = _cl_batch(rng, n, NI, m, nxt)
            tgt = (tgt & ~(1 << b_rep)).astype(np.uint8)          # an Accept never targets its proposer
            tr = rng.integers(0, 1 << 20, len(leader)).astype(np.int32)
            a, b = gpu.accept(leader, number, b_ord, b_rep, tr, tgt), ref.accept(leader, number, b_ord, b_rep, tr, tgt)
            assert a[0] == b[0] and a[0] in (0, fa.FPX_EFATAL_PROTOCOL)
            fatal += a[0] != 0
            for x, y in zip(a[1:], b[1:]):
                np.testing.assert_array_equal(x, y)
            nacks += int((a[2] != 0).sum())
            commits += int(a[5].sum())
            if kind == 3:     # re-send the very same Accepts: answered again (or refused by now-committed proposers)
                a, b = gpu.accept(leader, number, b_ord, b_rep, tr, tgt), ref.accept(leader, number, b_ord, b_rep, tr, tgt)
                assert a[0] == b[0]
                for x, y in zip(a[1:], b[1:]):
                    np.testing.assert_array_equal(x, y)
        else:
            leader, number, b_ord, b_rep, tgt = _cl_batch(rng, n, NI, m, nxt)
            a, b = gpu.prepare(leader, number, b_ord, b_rep, tgt), ref.prepare(leader, number, b_ord, b_rep, tgt)
            assert a[0] == b[0] == 0
            for x, y in zip(a[1:], b[1:]):
                np.testing.assert_array_equal(x, y)
            nacks += int((a[2] != 0).sum())
    assert fatal > 0 and nacks > 0 and commits > 0
    for r in range(n):
        for inst in rng.choice(n * NI, size=min(200, n * NI), replace=False):
            assert gpu.read_cmdlog(r, int(inst) // NI, int(inst) % NI) == ref.read_cmdlog(r, int(inst) // NI, int(inst) % NI)
    # malformed batches: duplicate instance, proposer among its targets, no command log configured
    one = np.zeros(2, np.int32)
    assert gpu.accept(one, one, one, one, one, np.zeros(2, np.uint8))[0] == fa.FPX_EINVAL
    assert gpu.accept([1], [1], [0], [1], [0], [0b010])[0] == fa.FPX_EINVAL
    assert EPaxos(n, 8).accept([1], [1], [0], [1], [0], [0b001])[0] == fa.FPX_EINVAL


def test_oracle_handle_commit_by_hand(oracle):
    """Replica.handleCommit (epaxos/Replica.scala:1567-1575 -> commit :815-830), n = 5, by hand: replica 2 holds an
    AcceptedEntry for (0, 5) in Ballot(3, 1) (one AcceptOk + the proposer's own = 2 < f + 1 = 3: not committed); a Commit
    for (0, 5) with another triple arrives at replicas {2, 3}: the entry is replaced WITHOUT a ballot comparison
    (:826-827 -- a Commit is final), the ballots become the null ballot, the triple's dependencies are what the Commit
    carried, the conflict index of key 1 learns (0, 5) as a set at replicas 2 and 3 (:828); replica 4 knows nothing.  A
    later Prepare in Ballot(9, 0) finds the CommittedEntry and is answered with the Commit (:1746-1756)."""
    e = oracle.EPaxos(5, 4, num_instances=16)
    assert e.accept([0], [5], [3], [1], [77], [0b00100], key=[2], is_set=[0])[0] == 0     # proposer 1 -> replica 2
    assert e.read_cmdlog(2, 0, 5)[:4] == (3, enc(3, 1), enc(3, 1), 77)
    assert e.handle_commit([0], [5], [88], [0b01100], key=[1], is_set=[1], deps=[[0, 3, 2, 0, 1]], deps_values_end=[0]) == 0
    for r in (2, 3):
        assert e.read_cmdlog(r, 0, 5)[:4] == (4, -1, -1, 88)
        d, end = e.read_cmdlog_deps(r, 0, 5)
        assert d.tolist() == [0, 3, 2, 0, 1] and end == 0
        assert e.read_index(r, 1)[1].tolist() == [6, 0, 0, 0, 0]               # sets: TopOne of leader 0 = 5 + 1
    assert e.read_cmdlog(4, 0, 5)[0] == 0 and e.read_index(4, 1)[1].tolist() == [0] * 5
    assert e.read_cmdlog(1, 0, 5)[0] == 3                                      # the proposer was not among the recipients
    st, ok, nack, com, nb, rs, rv, rt = e.prepare([0], [5], [9], [0], [0b01100])
    assert st == 0 and ok[0] == 0 and nack[0] == 0 and com[0] == 0b01100
    # explicit ids of the own column must lie above the instance; a key outside the index: nothing applied
    assert e.handle_commit([0], [6], [1], [0b00001], key=[0], is_set=[0], deps=[[7, 0, 0, 0, 0]], deps_values_end=[9]) == 1
    assert e.handle_commit([0], [6], [1], [0b00001], key=[4], is_set=[0]) == 1
    assert e.read_cmdlog(0, 0, 6)[0] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,NI,m", [(3, 64, 40), (5, 512, 700), (7, 300, 1500)])
def test_epaxos_handle_commit_matches_oracle(oracle, n, NI, m):
    """Commits from outside (Replica.handleCommit) between Accept / Prepare batches and pre-accept ticks: instances in every
    state, with dependencies or by triple id alone, Noops among them, arbitrary recipients -- the command log, the stored
    dependencies and the conflict indices, GPU == oracle; and the replies of the Prepares that follow (a CommittedEntry is
    answered with the Commit)"""
    from frankenpaxos_amd.epaxos import EPaxos
    import frankenpaxos_amd as fa

    K = 8
    gpu, ref = EPaxos(n, K, num_instances=NI), oracle.EPaxos(n, K, num_instances=NI)
    rng = np.random.default_rng(n * 7 + NI)
    nxt = [0] * n
    commits = 0
    for step in range(9):
        kind = step % 3
        if kind == 0 and max(nxt) + 60 < NI // 2:
            leader, number, key, is_set, mask, rank = random_tick(rng, n, K, 100, nxt, 3.0)
            tr = rng.integers(0, 1 << 20, 100).astype(np.int32)
            a, b = gpu.preaccept(leader, number, key, is_set, mask, rank, triple_id=tr), ref.preaccept(leader, number, key, is_set, mask, rank, triple_id=tr)
            assert a[0] == b[0] == 0
        elif kind == 1:
            leader, number, _, _, tgt = _cl_batch(rng, n, NI, m, nxt)
            k = len(leader)
            tr = rng.integers(0, 1 << 20, k).astype(np.int32)
            key = rng.integers(-1, K, k).astype(np.int32)
            is_set = rng.integers(0, 2, k).astype(np.uint8)
            deps = rng.integers(0, NI, (k, n)).astype(np.int32)
            ends = np.zeros(k, np.int32)
            own = deps[np.arange(k), leader]
            holes = rng.random(k) < 0.3                           # the own column as watermark <= number + explicit ids above
            deps[np.arange(k), leader] = np.where(holes, np.minimum(own, number), own)
            ends[holes] = number[holes] + 2 + rng.integers(0, 5, int(holes.sum()))
            by_id = step == 4
            args = dict(key=key, is_set=is_set) if by_id else dict(key=key, is_set=is_set, deps=deps, deps_values_end=ends)
            assert gpu.handle_commit(leader, number, tr, tgt, **args) == ref.handle_commit(leader, number, tr, tgt, **args) == 0
            commits += k
        else:
            leader, number, b_ord, b_rep, tgt = _cl_batch(rng, n, NI, m, nxt)
            a, b = gpu.prepare(leader, number, b_ord, b_rep, tgt), ref.prepare(leader, number, b_ord, b_rep, tgt)
            assert a[0] == b[0] == 0
            for x, y in zip(a[1:], b[1:]):
                np.testing.assert_array_equal(x, y)
    assert commits > 0
    for r in range(n):
        for inst in rng.choice(n * NI, size=min(300, n * NI), replace=False):
            L, x = int(inst) // NI, int(inst) % NI
            assert gpu.read_cmdlog(r, L, x) == ref.read_cmdlog(r, L, x)
            (da, ea), (db, eb) = gpu.read_cmdlog_deps(r, L, x), ref.read_cmdlog_deps(r, L, x)
            assert ea == eb and np.array_equal(da, db)
        for k in range(K):
            for x, y in zip(gpu.read_index(r, k), ref.read_index(r, k)):
                np.testing.assert_array_equal(x, y)
    assert gpu.handle_commit([0], [NI], [1], [1]) == fa.FPX_EINVAL and gpu.handle_commit([0], [1], [1], [1 << n]) == fa.FPX_EINVAL
    assert EPaxos(n, K).handle_commit([0], [1], [1], [1]) == fa.FPX_EINVAL


@pytest.mark.gpu
@pytest.mark.parametrize("n", [3, 5, 7])
def test_epaxos_commits_of_one_instance_in_one_batch_apply_in_order(oracle, n):
    """several Commits for ONE instance in one call (a re-sent Commit with another triple id, other dependencies,
    other recipients): the reference applies them one after the other, so per replica the LAST message that reaches it wins
    whole -- never one message's triple with another's dependencies; the conflict index sees every message's put"""
    from frankenpaxos_amd.epaxos import EPaxos

    NI, K = 64, 6
    gpu, ref = EPaxos(n, K, num_instances=NI), oracle.EPaxos(n, K, num_instances=NI)
    rng = np.random.default_rng(900 + n)
    for rep in range(6):
        base = 40
        inst = rng.integers(0, n * 12, base)                       # 12 numbers per leader: many repeats among 40 messages
        leader, number = (inst // 12).astype(np.int32), (inst % 12).astype(np.int32)
        tr = (1000 * rep + np.arange(base)).astype(np.int32)      # every message its own triple id
        key = rng.integers(0, K, base).astype(np.int32)
        is_set = rng.integers(0, 2, base).astype(np.uint8)
        deps = rng.integers(0, 12, (base, n)).astype(np.int32)
        deps[np.arange(base), leader] = np.minimum(deps[np.arange(base), leader], number)
        ends = np.where(rng.random(base) < 0.4, number + 2 + rng.integers(0, 3, base), 0).astype(np.int32)
        tgt = rng.integers(1, 1 << n, base).astype(np.uint8)
        args = dict(key=key, is_set=is_set, deps=deps, deps_values_end=ends)
        assert gpu.handle_commit(leader, number, tr, tgt, **args) == ref.handle_commit(leader, number, tr, tgt, **args) == 0
        for r in range(n):
            for L in range(n):
                for x in range(12):
                    assert gpu.read_cmdlog(r, L, x) == ref.read_cmdlog(r, L, x), (rep, r, L, x)
                    (da, ea), (db, eb) = gpu.read_cmdlog_deps(r, L, x), ref.read_cmdlog_deps(r, L, x)
                    assert ea == eb and np.array_equal(da, db), (rep, r, L, x)
            for k in range(K):
                for x, y in zip(gpu.read_index(r, k), ref.read_index(r, k)):
                    np.testing.assert_array_equal(x, y)


# ------------------------------------------------ handlePreAccept in full: ballots, Nacks, re-sent replies ----
def test_oracle_handle_preaccept_every_branch_by_hand(oracle):
    """n = 5, instance X = (0, 0): set k1 led by replica 0.  Replica 1 already knows the conflicting instance (2, 6).
      tick: X pre-accepts at {1, 2, 3}: replica 1 answers [0,0,7,0,0], 2 and 3 answer zeros -> slow path; the command log
        holds PreAcceptedEntry(Ballot(0,0), Ballot(0,0), triple = ITS OWN answer) at 0, 1, 2, 3 (Replica.scala:688-696,
        1259-1271); after the tick every index knows X (commit, :815-828);
      a) replica 0 re-sends PreAccept(X, Ballot(0,0)) to {1, 2, 4}: 1 and 2 have voted in that ballot -> the
        PreAcceptOk again with the stored dependencies (:1196-1210); 4 has no entry -> processes it (:1170-1172): its
        index holds X itself, subtractOne(X) takes it out again -> no dependencies;
      b) replica 2 starts a recovery: Prepare(X, Ballot(1,2)) at 1; the old PreAccept(X, Ballot(0,0)) at 1 is now
        Nacked with largestBallot (1,2) (:1188-1192);
      c) PreAccept(X, Ballot(2,3), deps [0,0,0,2,0]) at {1, 2}: larger than every ballot, not the vote ballot ->
        processed afresh: local conflicts (with X itself removed) U the message's dependencies;
      d) Accept(X, Ballot(3,4)) by 4 at 1, then PreAccept(X, Ballot(3,4)) at {1, 4}: accepted in that very ballot ->
        ignored (:1219-1224); PreAccept(X, Ballot(4,0)) at 1: a larger ballot pre-accepts again;
      e) Accept(X, Ballot(5,2)) at {0, 1, 3} commits X; any PreAccept(X) is answered with the Commit (:1227-1238)."""
    e = oracle.EPaxos(5, 4, num_instances=16)
    e.index_put(1, 1, 1, 2, 6)
    st, fast, deps, ldeps, own = e.preaccept([0], [0], [1], [1], [0b01110], np.zeros((5, 1), np.int32), triple_id=[42])
    assert st == 0 and fast[0] == 0 and deps[0].tolist() == [0, 0, 7, 0, 0] and ldeps[0].tolist() == [0] * 5
    assert e.read_cmdlog(1, 0, 0)[:4] == (2, enc(0, 0), enc(0, 0), 42) and e.read_cmdlog(4, 0, 0)[0] == 0
    assert e.read_cmdlog_deps(1, 0, 0)[0].tolist() == [0, 0, 7, 0, 0] and e.read_cmdlog_deps(0, 0, 0)[0].tolist() == [0] * 5
    zeros = np.zeros((1, 5), np.int32)
    # a)
    st, ok, resend, nack, com, nb, rd, re, rt = e.handle_preaccept([0], [0], [0], [0], [1], [1], [42], zeros, None, [0b10110])
    assert (st, ok[0], resend[0], nack[0], com[0], nb[0]) == (0, 0b10000, 0b00110, 0, 0, -1)
    assert rd[0].tolist() == [[0] * 5, [0, 0, 7, 0, 0], [0] * 5, [0] * 5, [0] * 5] and re[0].tolist() == [0] * 5
    assert rt[0].tolist() == [-1, 42, 42, -1, 42]
    assert e.read_cmdlog(4, 0, 0) == (2, enc(0, 0), enc(0, 0), 42, enc(0, 4))       # largestBallot stays (0,4) > (0,0)
    # b)
    assert e.prepare([0], [0], [1], [2], [0b00010])[1][0] == 0b00010
    st, ok, resend, nack, com, nb, rd, re, rt = e.handle_preaccept([0], [0], [0], [0], [1], [1], [42], zeros, None, [0b00010])
    assert (st, ok[0], resend[0], nack[0], com[0], nb[0]) == (0, 0, 0, 0b00010, 0, enc(1, 2))
    assert rd[0].tolist() == [[0] * 5] * 5 and rt[0].tolist() == [-1] * 5
    # c)
    st, ok, resend, nack, com, nb, rd, re, rt = e.handle_preaccept([0], [0], [2], [3], [1], [1], [43], [[0, 0, 0, 2, 0]], None,
                                                                    [0b00110])
    assert (st, ok[0], resend[0], nack[0], com[0]) == (0, 0b00110, 0, 0, 0)
    assert rd[0][1].tolist() == [0, 0, 7, 2, 0] and rd[0][2].tolist() == [0, 0, 0, 2, 0] and re[0].tolist() == [0] * 5
    assert e.read_cmdlog(1, 0, 0) == (2, enc(2, 3), enc(2, 3), 43, enc(2, 3))
    assert e.read_cmdlog_deps(1, 0, 0)[0].tolist() == [0, 0, 7, 2, 0]
    # d)
    st, ok, nack, com, nb, done = e.accept([0], [0], [3], [4], [44], [0b00010])
    assert st == 0 and ok[0] == 0b10010 and done[0] == 0
    st, ok, resend, nack, com, nb, rd, re, rt = e.handle_preaccept([0], [0], [3], [4], [1], [1], [44], zeros, None, [0b10010])
    assert (st, ok[0], resend[0], nack[0], com[0]) == (0, 0, 0, 0, 0)
    st, ok, resend, nack, com, nb, rd, re, rt = e.handle_preaccept([0], [0], [4], [0], [1], [1], [45], zeros, None, [0b00010])
    assert ok[0] == 0b00010 and e.read_cmdlog(1, 0, 0)[:4] == (2, enc(4, 0), enc(4, 0), 45)
    # e)
    st, ok, nack, com, nb, done = e.accept([0], [0], [5], [2], [46], [0b01011])
    assert st == 0 and done[0] == 1
    st, ok, resend, nack, com, nb, rd, re, rt = e.handle_preaccept([0], [0], [0], [1], [1], [1], [47], zeros, None, [0b11111])
    assert (st, ok[0], resend[0], nack[0], com[0]) == (0, 0, 0, 0, 0b11111)
    assert rt[0].tolist() == [46] * 5 and rd[0][:, 0].tolist() == [-1] * 5        # an Accept names its triple by id only
    # a Noop (key -1) has no conflicts and leaves the index alone (:592-593, 602-614); a fresh instance of leader 3
    g0, s0 = e.read_index(2, 1)
    st, ok, resend, nack, com, nb, rd, re, rt = e.handle_preaccept([3], [2], [0], [3], [-1], [0], [50], [[4, 0, 0, 0, 1]], None,
                                                                    [0b00100])
    assert ok[0] == 0b00100 and rd[0][2].tolist() == [4, 0, 0, 0, 1]
    g1, s1 = e.read_index(2, 1)
    assert g0.tolist() == g1.tolist() and s0.tolist() == s1.tolist()
    # the hole: replica 3 met (1, 5) on k2 before PreAccept((1, 3)) arrives: dependencies {0,1,2,4,5} of leader 1 =
    # watermark 3 + explicit values 4 .. 5 (values_end 6); a message that already carries a hole up to 8 widens it
    e.index_put(3, 2, 1, 1, 5)
    st, ok, resend, nack, com, nb, rd, re, rt = e.handle_preaccept([1, 1], [3, 2], [0, 0], [1, 1], [2, 2], [0, 0], [60, 61],
                                                                    [[0, 1, 0, 0, 0], [0, 2, 0, 0, 0]], [0, 9], [0b01000, 0b01000])
    assert st == 0 and ok.tolist() == [0b01000, 0b01000]
    assert rd[0][3].tolist() == [0, 3, 0, 0, 0] and re[0][3] == 6
    assert rd[1][3].tolist() == [0, 2, 0, 0, 0] and re[1][3] == 9
    # malformed: a PreAccept that depends on its own instance; duplicate instances in one batch; no command log
    assert e.handle_preaccept([1], [7], [0], [1], [2], [0], [1], [[0, 8, 0, 0, 0]], None, [0b00001])[0] == 1
    assert e.handle_preaccept([1, 1], [7, 7], [0, 0], [1, 1], [2, 2], [0, 0], [1, 1], np.zeros((2, 5)), None, [1, 1])[0] == 1
    assert oracle.EPaxos(5, 4).handle_preaccept([1], [7], [0], [1], [2], [0], [1], zeros, None, [1])[0] == 1


def _same(a, b):
    assert a[0] == b[0], (a[0], b[0])
    for x, y in zip(a[1:], b[1:]):
        np.testing.assert_array_equal(x, y)


@pytest.mark.gpu
def test_epaxos_handle_preaccept_hand_trace_matches_oracle(oracle):
    """the calls of test_oracle_handle_preaccept_every_branch_by_hand, GPU beside oracle: every reply, the command log
    with its stored dependencies and the conflict indexes after each step"""
    from frankenpaxos_amd.epaxos import EPaxos
    import frankenpaxos_amd as fa

    gpu, ref = EPaxos(5, 4, num_instances=16), oracle.EPaxos(5, 4, num_instances=16)
    zeros = np.zeros((1, 5), np.int32)

    def state():
        for r in range(5):
            for (L, x) in ((0, 0), (3, 2), (1, 3), (1, 2)):
                assert gpu.read_cmdlog(r, L, x) == ref.read_cmdlog(r, L, x)
                a, b = gpu.read_cmdlog_deps(r, L, x), ref.read_cmdlog_deps(r, L, x)
                assert a[0].tolist() == b[0].tolist() and a[1] == b[1]
            for k in range(4):
                for x, y in zip(gpu.read_index(r, k), ref.read_index(r, k)):
                    np.testing.assert_array_equal(x, y)

    # replica 1 alone knows (2, 6) on k1 before anything else
    for e in (gpu, ref):
        assert e.handle_preaccept([2], [6], [0], [2], [1], [1], [7], zeros, None, [0b00010])[1][0] == 0b00010
    tick = ([0], [0], [1], [1], [0b01110], np.zeros((5, 1), np.int32))
    _same(gpu.preaccept(*tick, triple_id=[42]), ref.preaccept(*tick, triple_id=[42]))
    state()
    calls = [
        ("hp", ([0], [0], [0], [0], [1], [1], [42], zeros, None, [0b10110])),
        ("prepare", ([0], [0], [1], [2], [0b00010])),
        ("hp", ([0], [0], [0], [0], [1], [1], [42], zeros, None, [0b00010])),
        ("hp", ([0], [0], [2], [3], [1], [1], [43], [[0, 0, 0, 2, 0]], None, [0b00110])),
        ("accept", ([0], [0], [3], [4], [44], [0b00010])),
        ("hp", ([0], [0], [3], [4], [1], [1], [44], zeros, None, [0b10010])),
        ("hp", ([0], [0], [4], [0], [1], [1], [45], zeros, None, [0b00010])),
        ("accept", ([0], [0], [5], [2], [46], [0b01011])),
        ("hp", ([0], [0], [0], [1], [1], [1], [47], zeros, None, [0b11111])),
        ("hp", ([3], [2], [0], [3], [-1], [0], [50], [[4, 0, 0, 0, 1]], None, [0b00100])),
        ("hp", ([1, 1], [3, 2], [0, 0], [1, 1], [2, 2], [0, 0], [60, 61], [[0, 1, 0, 0, 0], [0, 2, 0, 0, 0]], [0, 9],
                [0b01000, 0b01000])),
    ]
    for kind, args in calls:
        fn = {"hp": "handle_preaccept", "prepare": "prepare", "accept": "accept"}[kind]
        _same(getattr(gpu, fn)(*args), getattr(ref, fn)(*args))
        state()
    assert gpu.handle_preaccept([1], [7], [0], [1], [2], [0], [1], [[0, 8, 0, 0, 0]], None, [0b00001])[0] == fa.FPX_EINVAL
    assert gpu.handle_preaccept([1, 1], [7, 7], [0, 0], [1, 1], [2, 2], [0, 0], [1, 1], np.zeros((2, 5)), None, [1, 1])[0] == fa.FPX_EINVAL
    assert EPaxos(5, 4).handle_preaccept([1], [7], [0], [1], [2], [0], [1], zeros, None, [1])[0] == fa.FPX_EINVAL
    state()


@pytest.mark.gpu
@pytest.mark.parametrize("n,NI,m,num_keys", [(3, 64, 40, 4), (5, 512, 700, 8), (7, 300, 1500, 3), (5, 4096, 5000, 64)])
def test_epaxos_handle_preaccept_matches_oracle(oracle, n, NI, m, num_keys):
    """random PreAccept batches (ballots up and down, arbitrary targets, instances in every command-log state, Noops,
    dependencies with and without holes) between pre-accept ticks, Prepares and Accepts: every reply, then the whole
    command log with the stored dependencies and every conflict index, GPU == oracle"""
    from frankenpaxos_amd.epaxos import EPaxos

    gpu, ref = EPaxos(n, num_keys, num_instances=NI), oracle.EPaxos(n, num_keys, num_instances=NI)
    rng = np.random.default_rng(n * 131 + NI)
    nxt = [0] * n
    seen = {k: 0 for k in ("ok", "resend", "nack", "commit", "ignored", "holes")}
    for step in range(16):
        kind = step % 4
        if kind == 0 and max(nxt) + min(m, 200) // 2 + 8 < NI // 2:
            mm = min(m, 200)
            leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, mm, nxt, 3.0, fifo=bool(step & 4))
            tr = rng.integers(0, 1 << 20, mm).astype(np.int32)
            _same(gpu.preaccept(leader, number, key, is_set, mask, rank, triple_id=tr),
                  ref.preaccept(leader, number, key, is_set, mask, rank, triple_id=tr))
            continue
        leader, number, b_ord, b_rep, tgt = _cl_batch(rng, n, NI, m, nxt)
        mm = len(leader)
        if kind == 2 and step % 8 == 2:
            _same(gpu.prepare(leader, number, b_ord, b_rep, tgt), ref.prepare(leader, number, b_ord, b_rep, tgt))
            continue
        if kind == 2:
            tgt = (tgt & ~(1 << b_rep)).astype(np.uint8)
            tr = rng.integers(0, 1 << 20, mm).astype(np.int32)
            a, b = gpu.accept(leader, number, b_ord, b_rep, tr, tgt), ref.accept(leader, number, b_ord, b_rep, tr, tgt)
            _same(a, b)
            continue
        key = rng.integers(-1, num_keys, mm).astype(np.int32)           # -1: Noop
        is_set = (rng.random(mm) < 0.5).astype(np.uint8)
        tr = rng.integers(0, 1 << 20, mm).astype(np.int32)
        din = rng.integers(0, NI, (mm, n)).astype(np.int32)
        dend = np.zeros(mm, np.int32)
        own = din[np.arange(mm), leader]
        hole = rng.random(mm) < 0.3
        # the own-leader column: a plain watermark <= the instance number, or watermark == number with values above
        din[np.arange(mm), leader] = np.where(hole, number, np.minimum(own, number))
        dend[:] = np.where(hole, number + 2 + rng.integers(0, 5, mm), 0)
        if step % 8 == 5:    # the old leaders' original PreAccepts once more: Ballot(0, leader)
            b_ord[:], b_rep[:] = 0, leader
        a = gpu.handle_preaccept(leader, number, b_ord, b_rep, key, is_set, tr, din, dend, tgt)
        b = ref.handle_preaccept(leader, number, b_ord, b_rep, key, is_set, tr, din, dend, tgt)
        _same(a, b)
        st, ok, resend, nack, com, nb, rd, re, rt = a
        assert st == 0
        for name, bits in (("ok", ok), ("resend", resend), ("nack", nack), ("commit", com)):
            seen[name] += int(np.unpackbits(bits).sum())
        seen["ignored"] += int(np.unpackbits(tgt & ~(ok | resend | nack | com)).sum())
        seen["holes"] += int((re != 0).sum())
    assert seen["ok"] > 0 and seen["nack"] > 0 and (m < 500 or all(v > 0 for v in seen.values())), seen
    for r in range(n):
        for inst in rng.choice(n * NI, size=min(300, n * NI), replace=False):
            L, x = int(inst) // NI, int(inst) % NI
            assert gpu.read_cmdlog(r, L, x) == ref.read_cmdlog(r, L, x)
            a, b = gpu.read_cmdlog_deps(r, L, x), ref.read_cmdlog_deps(r, L, x)
            assert a[0].tolist() == b[0].tolist() and a[1] == b[1]
        for k in range(num_keys):
            for x, y in zip(gpu.read_index(r, k), ref.read_index(r, k)):
                np.testing.assert_array_equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,hot,NI", [(5, 6000, 0.5, 0), (5, 6000, 0.5, 4096), (3, 9000, 0.4, 0), (7, 4000, 0.3, 2048),
                                         (5, 3000, 0.0, 2048)])
def test_epaxos_hot_keys_take_the_long_way(oracle, n, m, hot, NI):
    """a key with more commands per tick than the on-chip tables of k_epx_key hold (1152 at n = 5) is left to
    k_epx_scan / k_epx_decide while the other keys of the same tick are scanned and decided on chip: a hot key that
    draws `hot` of the tick beside 63 cold ones, with and without the command log -- every output, the stored
    dependencies and the conflict indexes, GPU == oracle"""
    from frankenpaxos_amd.epaxos import EPaxos

    num_keys = 64
    gpu, ref = EPaxos(n, num_keys, num_instances=NI), oracle.EPaxos(n, num_keys, num_instances=NI)
    rng = np.random.default_rng(n * 77 + m)
    nxt = [0] * n
    for tick in range(3):
        leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 8.0, fifo=bool(tick & 1))
        key = np.where(rng.random(m) < hot, 5, key).astype(np.int32)
        if NI and max(nxt) >= NI:
            break
        tr = rng.integers(0, 1 << 20, m).astype(np.int32) if NI else None
        a = gpu.preaccept(leader, number, key, is_set, mask, rank, triple_id=tr)
        b = ref.preaccept(leader, number, key, is_set, mask, rank, triple_id=tr)
        _same(a, b)
        assert a[0] == 0
    for r in range(n):
        for k in range(num_keys):
            for x, y in zip(gpu.read_index(r, k), ref.read_index(r, k)):
                np.testing.assert_array_equal(x, y)
        if NI:
            for inst in rng.choice(n * NI, size=400, replace=False):
                L, x = int(inst) // NI, int(inst) % NI
                assert gpu.read_cmdlog(r, L, x) == ref.read_cmdlog(r, L, x)
                c, d = gpu.read_cmdlog_deps(r, L, x), ref.read_cmdlog_deps(r, L, x)
                assert c[0].tolist() == d[0].tolist() and c[1] == d[1]


@pytest.mark.gpu
def test_epaxos_ticks_without_the_per_key_workgroups(oracle, monkeypatch):
    """k_epx_scan / k_epx_decide alone (what a tick of 2^21 or more commands, or more than 65536 keys, uses):
    FPX_EPX_NO_KEY_TILES switches k_epx_key off; same answers"""
    from frankenpaxos_amd.epaxos import EPaxos

    monkeypatch.setenv("FPX_EPX_NO_KEY_TILES", "1")
    n, num_keys, m = 5, 16, 3000
    gpu, ref = EPaxos(n, num_keys, num_instances=2048), oracle.EPaxos(n, num_keys, num_instances=2048)
    rng = np.random.default_rng(99)
    nxt = [0] * n
    for tick in range(2):
        args = random_tick(rng, n, num_keys, m, nxt, 6.0, fifo=False)
        tr = rng.integers(0, 1 << 20, m).astype(np.int32)
        _same(gpu.preaccept(*args, triple_id=tr), ref.preaccept(*args, triple_id=tr))
    for r in range(n):
        for inst in rng.choice(n * 2048, size=300, replace=False):
            L, x = int(inst) // 2048, int(inst) % 2048
            assert gpu.read_cmdlog(r, L, x) == ref.read_cmdlog(r, L, x)
            a, b = gpu.read_cmdlog_deps(r, L, x), ref.read_cmdlog_deps(r, L, x)
            assert a[0].tolist() == b[0].tolist() and a[1] == b[1]


# ------------------------------------------------ commit -> dependency graph -> execution ----
def _slow_path_accepts(e, leader, number, key, is_set, mask, fast, triple, f):
    """the slow-path instances of a tick go through the Accept phase (preAcceptingSlowPath ->
    transitionToAcceptPhase, Replica.scala:796-813, 732-792): Ballot(0, leader), sent to the first f replicas of the
    fast quorum (thriftyOtherReplicas(slowQuorumSize - 1), :774); with the proposer's own AcceptOk that is f + 1"""
    slow = np.nonzero(fast == 0)[0]
    tgt = np.zeros(len(slow), np.uint8)
    for j, i in enumerate(slow):
        picked = [r for r in range(8) if (mask[i] >> r) & 1][:f]
        tgt[j] = sum(1 << r for r in picked)
    out = e.accept(leader[slow], number[slow], np.zeros(len(slow), np.int32), leader[slow], triple[slow], tgt,
                   key[slow], is_set[slow])
    return slow, out


def check_execution_order(n, leader, number, deps, own_end, el, ei, cs):
    """size-independent properties of an execution order when EVERYTHING was committed: every instance exactly once;
    a dependency's component never comes after the instance's (reverse topological order of the condensation)"""
    m = len(leader)
    assert len(el) == m and int(cs.sum()) == m
    comp_of_pos = np.repeat(np.arange(len(cs)), cs)
    per = [int(number[leader == L].max()) + 1 if (leader == L).any() else 0 for L in range(n)]
    comp = [np.full(per[L], -1, np.int64) for L in range(n)]
    for L in range(n):
        sel = el == L
        comp[L][ei[sel]] = comp_of_pos[sel]
        assert (comp[L] >= 0).all()                       # every instance of the column executed (ids are dense)
    seen = sum(len(np.unique(ei[el == L])) for L in range(n))
    assert seen == m                                      # ... and none twice
    prefmax = [np.concatenate([[-1], np.maximum.accumulate(comp[L])]) for L in range(n)]
    mine = np.empty(m, np.int64)
    for L in range(n):
        sel = leader == L
        mine[sel] = comp[L][number[sel]]
    for L in range(n):
        w = np.minimum(deps[:, L], per[L])
        assert (prefmax[L][w] <= mine).all()
    for i in np.nonzero(own_end)[0]:                      # the explicit ids of the own column
        L = leader[i]
        assert comp[L][number[i] + 1: own_end[i]].max(initial=-1) <= mine[i]


@pytest.mark.gpu
@pytest.mark.parametrize("n,num_keys,m,fifo,kind", [(3, 8, 3000, True, "zigzag"), (5, 16, 8000, True, "zigzag"),
                                                    (5, 64, 12000, False, "zigzag"), (7, 4, 3000, False, "zigzag"),
                                                    (5, 16, 4000, False, "tarjan")])
def test_epaxos_commits_execute_in_the_oracles_order(oracle, n, num_keys, m, fifo, kind):
    """two ticks end to end: GPU pre-accept -> Accept phase for the slow path (GPU) -> every committed triple into the
    product's dependency graph -> execution order; beside it the oracle chain (C oracle EPaxos -> oracle/depgraph.py).
    Executables, component boundaries and blockers are compared after every tick, bit for bit."""
    from frankenpaxos_amd import depgraph as P
    from frankenpaxos_amd.epaxos import EPaxos
    from oracle import depgraph as O

    NI = 2 * m
    gpu, ref = EPaxos(n, num_keys, num_instances=NI), oracle.EPaxos(n, num_keys, num_instances=NI)
    f = (n - 1) // 2
    if kind == "zigzag":
        pg, og = P.DependencyGraph(n, kind=P.FPX_DG_ZIGZAG), O.ZigzagTarjanDependencyGraph(O.InstancePrefixSet(n), n)
    else:
        pg, og = P.DependencyGraph(n, kind=P.FPX_DG_TARJAN), O.TarjanDependencyGraph(O.InstancePrefixSet(n))
    rng = np.random.default_rng(n * 100 + num_keys)
    nxt = [0] * n
    executed = cycles = holes = 0
    for tick in range(2):
        leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 6.0, fifo=fifo)
        triple = np.arange(tick * m, (tick + 1) * m, dtype=np.int32)
        a = gpu.preaccept(leader, number, key, is_set, mask, rank, triple_id=triple)
        b = ref.preaccept(leader, number, key, is_set, mask, rank, triple_id=triple)
        _same(a, b)
        st, fast, deps, ldeps, own = a
        slow, acc_g = _slow_path_accepts(gpu, leader, number, key, is_set, mask, fast, triple, f)
        _, acc_r = _slow_path_accepts(ref, leader, number, key, is_set, mask, b[1], triple, f)
        _same(acc_g, acc_r)
        committed = fast.copy()
        committed[slow] = acc_g[5]
        assert committed.all() and (n == 3 or len(slow) > 0)
        holes += int((own[:, 0] != 0).sum())
        # product: the GPU's outputs as they are; oracle: the oracle's outputs through the reference-shaped sets
        pg.commit_epx(leader, number, deps, own, mask=committed)
        for i in range(m):
            og.commit((int(leader[i]), int(number[i])), 0,
                      O.InstancePrefixSet.from_epx(int(leader[i]), int(number[i]), b[2][i], int(b[4][i, 0])))
        want = O.with_deep_stack(og.execute_by_component)
        got = pg.execute_by_component()
        assert got[0] == [list(c) for c in want[0]] and got[1] == want[1]
        executed += sum(len(c) for c in got[0])
        cycles += sum(1 for c in got[0] if len(c) > 1)
    assert executed == 2 * m and cycles > 0 and (holes > 0) == (not fifo)


@pytest.mark.gpu
def test_config4_commits_execute_at_full_size(oracle):
    """BASELINE.json configs[3] carried through: a 2^20-command tick (n = 5, 1024 keys) pre-accepts on the GPU, the
    slow path goes through the Accept phase at size (K6 at 10^5-10^6 messages, GPU == oracle on every reply), every
    committed triple enters the dependency graph and the whole tick executes: each instance exactly once, no
    dependency after its dependent (check_execution_order).  The zigzag variant only: TarjanDependencyGraph walks every
    vertex's whole dependency prefix (its executed set moves after the pass, TarjanDependencyGraph.scala:266-273) --
    quadratic on prefix sets, in the reference as here, which is why the reference deploys the zigzag one."""
    from frankenpaxos_amd import depgraph as P
    from frankenpaxos_amd.epaxos import EPaxos
    from tests import workloads as W

    n, num_keys, m = 5, 1024, 1 << 20
    NI = m // n + 4096
    gpu, ref = EPaxos(n, num_keys, num_instances=NI), oracle.EPaxos(n, num_keys, num_instances=NI)
    rng = np.random.default_rng(44)
    nxt = [0] * n
    leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 64.0, fifo=False)
    assert max(nxt) <= NI
    key = (W.splitmix64_at(np.arange(m, dtype=np.uint64)) % np.uint64(num_keys)).astype(np.int32)
    triple = np.arange(m, dtype=np.int32)
    a = gpu.preaccept(leader, number, key, is_set, mask, rank, triple_id=triple)
    b = ref.preaccept(leader, number, key, is_set, mask, rank, triple_id=triple)
    _same(a, b)
    st, fast, deps, ldeps, own = a
    slow, acc_g = _slow_path_accepts(gpu, leader, number, key, is_set, mask, fast, triple, 2)
    _, acc_r = _slow_path_accepts(ref, leader, number, key, is_set, mask, fast, triple, 2)
    _same(acc_g, acc_r)
    assert len(slow) > 10_000 and acc_g[5].all()
    for r in range(n):                                    # the Accept phase left every index as the oracle's
        for k in range(0, num_keys, 37):
            ga, sa = gpu.read_index(r, k)
            gb, sb = ref.read_index(r, k)
            assert ga.tolist() == gb.tolist() and sa.tolist() == sb.tolist()
    zz = P.DependencyGraph(n, kind=P.FPX_DG_ZIGZAG)
    zz.commit_epx(leader, number, deps, own)
    el, ei, cs, bl, bi = zz.execute_arrays()
    check_execution_order(n, leader, number, deps, own[:, 0], el, ei, cs)
    assert sorted(zip(bl.tolist(), bi.tolist())) == [(L, nxt[L]) for L in range(n)]   # blocked on the next ids only


# ---- K5, second form (fpx_epaxos_kp.hpp): packed lines, hot keys, both forms on the same ticks -------------------
@pytest.mark.gpu
@pytest.mark.parametrize("n,num_keys,m,fifo", [(5, 64, 20000, True), (3, 8, 6000, False), (7, 32, 9000, False),
                                                (5, 2048, 30000, True), (5, 1, 900, False)])
def test_epaxos_packed_lines_equal_the_four_arrays_and_the_oracle(oracle, n, num_keys, m, fifo):
    """fpx_epx_preaccept_packed_dev: line i = deps | leader_deps | own_values_end | fast, bit for bit what the four
    arrays of fpx_epx_preaccept hold and what the oracle computes, tick after tick (the index carries over)"""
    import torch
    from frankenpaxos_amd.epaxos import EPaxos

    gpu, ref = EPaxos(n, num_keys), oracle.EPaxos(n, num_keys)
    rng = np.random.default_rng(n * 1000 + num_keys)
    nxt = [0] * n
    dev = torch.device("cuda:0")
    stride = gpu.packed_stride()
    assert stride == {3: 12, 5: 16, 7: 20}[n]
    for tick in range(3):
        leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 9.0, fifo=fifo)
        want = ref.preaccept(leader, number, key, is_set, mask, rank)
        assert want[0] == 0
        t = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (leader, number, key, is_set, mask, rank)]
        packed = torch.full((m, stride), -7, dtype=torch.int32, device=dev)
        gpu.preaccept_packed_dev(*t, packed)
        assert gpu.sync() == 0
        fast, deps, ldeps, own = (x.cpu().numpy() for x in gpu.unpack(packed))
        np.testing.assert_array_equal(fast, want[1].astype(np.int32))
        np.testing.assert_array_equal(deps, want[2])
        np.testing.assert_array_equal(ldeps, want[3])
        np.testing.assert_array_equal(own, want[4])
        assert not packed[:, 2 * n + 3:].any()                           # the padding is written (zeros)
    for r in range(n):
        for k in range(min(num_keys, 40)):
            ga, sa = gpu.read_index(r, k)
            gb, sb = ref.read_index(r, k)
            assert ga.tolist() == gb.tolist() and sa.tolist() == sb.tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [3, 5, 7])
def test_epaxos_a_hot_key_goes_the_first_forms_way(oracle, n):
    """one key holds more commands than the on-chip tables of k_epx_key2 take (and one exactly as many as fit): the
    tick is handed to the first form whole, packed or not, with the same results; the next tick is small again"""
    import torch
    from frankenpaxos_amd.epaxos import EPaxos

    num_keys = 16
    tc = {3: 1536, 5: 1152, 7: 832}[n]
    gpu, ref = EPaxos(n, num_keys), oracle.EPaxos(n, num_keys)
    rng = np.random.default_rng(n)
    nxt = [0] * n
    dev = torch.device("cuda:0")
    for tick, hot in enumerate([tc, tc + 1, 3 * tc + 17, 10]):
        m = hot + 3000
        leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 9.0, fifo=tick % 2 == 0)
        key[:] = 1 + rng.integers(0, num_keys - 1, m)       # ~200 commands on each of the other keys
        key[rng.permutation(m)[:hot]] = 0                   # `hot` commands on key 0
        want = ref.preaccept(leader, number, key, is_set, mask, rank)
        assert want[0] == 0
        if tick % 2:
            got = gpu.preaccept(leader, number, key, is_set, mask, rank)
            assert got[0] == 0
            for x, y in zip(got[1:], want[1:]):
                np.testing.assert_array_equal(x, y)
        else:
            t = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (leader, number, key, is_set, mask, rank)]
            packed = torch.zeros((m, gpu.packed_stride()), dtype=torch.int32, device=dev)
            gpu.preaccept_packed_dev(*t, packed)
            assert gpu.sync() == 0
            fast, deps, ldeps, own = (x.cpu().numpy() for x in gpu.unpack(packed))
            np.testing.assert_array_equal(fast, want[1].astype(np.int32))
            np.testing.assert_array_equal(deps, want[2])
            np.testing.assert_array_equal(ldeps, want[3])
            np.testing.assert_array_equal(own, want[4])
    for r in range(n):
        for k in range(num_keys):
            ga, sa = gpu.read_index(r, k)
            gb, sb = ref.read_index(r, k)
            assert ga.tolist() == gb.tolist() and sa.tolist() == sb.tolist()


@pytest.mark.gpu
def test_epaxos_both_forms_agree_and_bad_ranks_are_refused(oracle, monkeypatch):
    """FPX_EPX_V1 forces the first form: the two agree on the same ticks; a rank row with a repeated value, with a value
    out of range, or two rows swapped into each other's multisets -- FPX_EINVAL from both, nothing applied"""
    from frankenpaxos_amd.epaxos import EPaxos
    import frankenpaxos_amd as fa

    n, num_keys, m = 5, 128, 40000
    v2 = EPaxos(n, num_keys)
    monkeypatch.setenv("FPX_EPX_V1", "1")
    v1 = EPaxos(n, num_keys)
    monkeypatch.delenv("FPX_EPX_V1")
    ref = oracle.EPaxos(n, num_keys)
    rng = np.random.default_rng(11)
    nxt = [0] * n
    for tick in range(3):
        args = random_tick(rng, n, num_keys, m, nxt, 12.0, fifo=False)
        a, b, c = v2.preaccept(*args), v1.preaccept(*args), ref.preaccept(*args)
        assert a[0] == b[0] == c[0] == 0
        for x, y, z in zip(a[1:], b[1:], c[1:]):
            np.testing.assert_array_equal(x, y)
            np.testing.assert_array_equal(x, z)
        leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, list(nxt), 12.0)
        for kind in range(3):
            bad = rank.copy()
            if kind == 0:
                bad[3, 17] = bad[3, 18]                      # a repeated position (another one is empty)
            elif kind == 1:
                bad[1, 5] = m                                # out of range
            else:
                lo, hi = (7, 8) if bad[0, 7] < bad[0, 8] else (8, 7)
                bad[0, lo] += 1                              # the same sum of values ...
                bad[0, hi] -= 1                              # ... but two positions now taken twice
                if (np.sort(bad[0]) == np.arange(m)).all():
                    continue
            assert v2.preaccept(leader, number, key, is_set, mask, bad)[0] == fa.FPX_EINVAL
            assert v1.preaccept(leader, number, key, is_set, mask, bad)[0] == fa.FPX_EINVAL
            assert ref.preaccept(leader, number, key, is_set, mask, bad)[0] == 1
    for r in range(n):
        for k in range(num_keys):
            assert [x.tolist() for x in v2.read_index(r, k)] == [x.tolist() for x in ref.read_index(r, k)]
            assert [x.tolist() for x in v1.read_index(r, k)] == [x.tolist() for x in ref.read_index(r, k)]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [3, 5, 7])
def test_epaxos_clumped_ranks_take_the_radix_sort(oracle, n):
    """k_epx_key2 orders a key's commands by rank with a bucket sort that assumes the ranks spread out; two clumps at
    the ends of the delivery order put dozens of commands into one bucket and the key through the LSD radix sort"""
    from frankenpaxos_amd.epaxos import EPaxos

    num_keys, m = 8, 20000
    gpu, ref = EPaxos(n, num_keys), oracle.EPaxos(n, num_keys)
    rng = np.random.default_rng(70 + n)
    nxt = [0] * n
    for tick in range(3):
        leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 2.0, fifo=tick != 1)
        key[:] = 1 + rng.integers(0, num_keys - 1, m)
        key[:150] = 0                                       # delivered first everywhere (skew 2) ...
        key[m - 150:] = 0                                   # ... and last: 150 commands in each end bucket of key 0
        a, b = gpu.preaccept(leader, number, key, is_set, mask, rank), ref.preaccept(leader, number, key, is_set, mask, rank)
        assert a[0] == b[0] == 0
        for x, y in zip(a[1:], b[1:]):
            np.testing.assert_array_equal(x, y)
    for r in range(n):
        for k in range(num_keys):
            assert [x.tolist() for x in gpu.read_index(r, k)] == [x.tolist() for x in ref.read_index(r, k)]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [3, 5, 7])
def test_epaxos_a_burst_of_one_key_takes_the_second_bucket_attempt(oracle, n):
    """k_epx_key2's first sorting attempt spreads a key's ranks over the whole tick's [0, m); a key whose commands all
    arrive in one burst fills a handful of those buckets, and the second attempt buckets them over the key's own rank
    range (a third of the keys here: bursts of 400 - 600 neighbours in the delivery order, the others spread out)"""
    from frankenpaxos_amd.epaxos import EPaxos

    num_keys, m = 12, 30000
    gpu, ref = EPaxos(n, num_keys), oracle.EPaxos(n, num_keys)
    rng = np.random.default_rng(170 + n)
    nxt = [0] * n
    for tick in range(3):
        leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 3.0, fifo=tick != 1)
        key[:] = 4 + rng.integers(0, num_keys - 4, m)
        for k, (at, length) in enumerate([(100, 600), (9000, 400), (20000, 500), (m - 450, 450)]):
            key[at:at + length] = k
        a, b = gpu.preaccept(leader, number, key, is_set, mask, rank), ref.preaccept(leader, number, key, is_set, mask, rank)
        assert a[0] == b[0] == 0
        for x, y in zip(a[1:], b[1:]):
            np.testing.assert_array_equal(x, y)
    for r in range(n):
        for k in range(num_keys):
            assert [x.tolist() for x in gpu.read_index(r, k)] == [x.tolist() for x in ref.read_index(r, k)]


# ---- K8: Replica.handlePrepareOk, the recovering replica's decision (Replica.scala:1759-1884) ---------------------
def test_oracle_handle_prepare_oks_by_hand(oracle):
    """n = 5 (f = 2, slow quorum 3), instance X = (0, 0), recovered by replica 4.
      * PreAccept(X, Ballot(0, 0), triple 7) processed at replicas 1, 2, 3 (identical answers: empty indexes)
      * replica 4 prepares X in Ballot(1, 4) at {1, 2, 3}: three PrepareOk(PreAccepted, voteBallot (0, 0), triple 7)
          two of them in hand            -> wait
          as the reference evaluates it  -> pre-accept triple 7's command again (its popularItems filter looks at the
                                            Prepare's ballot (1, 4), never the default ballot: nothing is "popular")
          as its comments intend         -> f = 2 identical default-ballot pre-accepts not from replica 4: Accept phase
      * replica 3 runs the Accept phase of X in Ballot(2, 3) with triple 9 at replica 2 only (2 of the 3 it needs)
      * replica 4 prepares again in Ballot(3, 4): replica 1 PreAccepted in (0, 0); replicas 2, 3 Accepted in (2, 3)
          as the reference evaluates it  -> Noop!  (status == Some(Accepted) compares an enum with an Option: never true;
                                            no PreAccepted response at the highest voteBallot is left)
          as its comments intend         -> Accept phase with triple 9
      * only NotSeen responses -> Noop either way"""
    e = oracle.EPaxos(5, 4, num_instances=16)
    z = np.zeros((1, 5), np.int32)
    out = e.handle_preaccept([0], [0], [0], [0], [1], [1], [7], z, None, [0b01110])
    assert out[0] == 0 and out[1][0] == 0b01110
    st, ok, nack, com, nb, rs, rv, rt = e.prepare([0], [0], [1], [4], [0b01110])
    assert st == 0 and ok[0] == 0b01110 and rs[0].tolist() == [-1, 2, 2, 2, -1] and rv[0].tolist() == [-1, 0, 0, 0, -1]
    dec = lambda mask, bo, intended: tuple(int(x[0]) for x in e.handle_prepare_oks([0], [0], [bo], [4], [mask], rs, rv, rt, intended)[1:])
    assert dec(0b00110, 1, False) == (0, -1, -1) and dec(0b00110, 1, True) == (0, -1, -1)
    assert dec(0b01110, 1, False) == (2, 1, 7)
    assert dec(0b01110, 1, True) == (1, 1, 7)
    # a response the recovering replica did not get is not a response: status -1 inside the mask is a malformed call
    assert e.handle_prepare_oks([0], [0], [1], [4], [0b11110], rs, rv, rt)[0] == 1
    a = e.accept([0], [0], [2], [3], [9], [0b00100], [1], [1])
    assert a[0] == 0 and a[1][0] == 0b01100 and a[5][0] == 0            # AcceptOk from 2 and from 3 itself: 2 of the 3 needed
    st, ok, nack, com, nb, rs, rv, rt = e.prepare([0], [0], [3], [4], [0b01110])
    assert ok[0] == 0b01110 and rs[0].tolist() == [-1, 2, 3, 3, -1] and rv[0].tolist() == [-1, 0, 19, 19, -1] and rt[0].tolist() == [-1, 7, 9, 9, -1]
    assert dec(0b01110, 3, False) == (3, -1, -1)
    assert dec(0b01110, 3, True) == (1, 2, 9)
    fresh = oracle.EPaxos(5, 4, num_instances=16)
    st, ok, nack, com, nb, rs, rv, rt = fresh.prepare([0], [0], [1], [4], [0b00111])
    assert rs[0].tolist() == [0, 0, 0, -1, -1]
    for intended in (False, True):
        assert tuple(int(x[0]) for x in fresh.handle_prepare_oks([0], [0], [1], [4], [0b00111], rs, rv, rt, intended)[1:]) == (3, -1, -1)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [3, 5, 7])
def test_epaxos_handle_prepare_oks_matches_oracle(oracle, n):
    """random PrepareOk sets (statuses, vote ballots and triples drawn from small ranges so that ties and agreements
    happen) over a command log filled by ticks and re-sent PreAccepts: both evaluations, GPU == oracle; then a recovery
    end to end -- tick, Prepare by another replica, its decision, the pre-accept phase again in the recovery ballot
    (avoiding the fast path), the Accept phase, the commit -- step by step against the oracle"""
    from frankenpaxos_amd.epaxos import EPaxos
    import frankenpaxos_amd as fa

    NI, K = 256, 8
    gpu, ref = EPaxos(n, K, num_instances=NI), oracle.EPaxos(n, K, num_instances=NI)
    rng = np.random.default_rng(n)
    nxt = [0] * n
    leader, number, key, is_set, mask, rank = random_tick(rng, n, K, 300, nxt, 40.0, fifo=False)
    tr = np.arange(300, dtype=np.int32)
    a, b = gpu.preaccept(leader, number, key, is_set, mask, rank, triple_id=tr), ref.preaccept(leader, number, key, is_set, mask, rank, triple_id=tr)
    _same(a, b)
    slow = np.nonzero(a[1] == 0)[0]
    assert len(slow) > 20 or n == 3     # (n = 3: one counted answer is always "identical" -- every commit is a fast one)
    # 1. random reply sets on the tick's instances
    m = 300
    f = (n - 1) // 2
    seen = {0: 0, 1: 0, 2: 0, 3: 0}
    for intended in (False, True):
        rs = rng.choice([0, 2, 2, 3], size=(m, n)).astype(np.int32)
        rv = np.where(rs == 0, -1, rng.choice([0, 1, 2, 9], size=(m, n)) + rng.integers(0, 2, (m, n)) * leader[:, None]).astype(np.int32)
        rt = np.where(rs == 0, -1, rng.integers(0, 3, (m, n)) + tr[:, None]).astype(np.int32)
        msk = rng.integers(0, 1 << n, m).astype(np.uint8)
        b_ord, b_rep = rng.integers(0, 3, m).astype(np.int32), rng.integers(0, n, m).astype(np.int32)
        x, y = (e.handle_prepare_oks(leader, number, b_ord, b_rep, msk, rs, rv, rt, intended) for e in (gpu, ref))
        _same(x, y)
        for k in seen:
            seen[k] += int((x[1] == k).sum())
    assert all(v > 0 for v in seen.values()), seen
    bad = rs.copy()
    bad[5, 0] = -1
    msk[5] |= 1
    assert gpu.handle_prepare_oks(leader, number, b_ord, b_rep, msk, bad, rv, rt)[0] == fa.FPX_EINVAL
    assert ref.handle_prepare_oks(leader, number, b_ord, b_rep, msk, bad, rv, rt)[0] == 1
    if n == 3:
        return
    # 2. recovery of the slow-path instances by replica P = (leader + 1) % n in Ballot(1, P)
    sl, sx = leader[slow], number[slow]
    P = ((sl + 1) % n).astype(np.int32)
    one = np.ones(len(slow), np.int32)
    everyone_else = (((1 << n) - 1) & ~(1 << P.astype(np.int64))).astype(np.uint8)
    x, y = (e.prepare(sl, sx, one, P, everyone_else) for e in (gpu, ref))
    _same(x, y)
    st, ok, nack, com, nb, rs, rv, rt = x
    assert (ok == everyone_else).all()
    x, y = (e.handle_prepare_oks(sl, sx, one, P, ok, rs, rv, rt) for e in (gpu, ref))
    _same(x, y)
    st, act, src, trp = x
    assert (act == 2).all() and (trp == tr[slow]).all()     # every slow-path instance is PreAccepted somewhere: its command again
    deps0 = np.zeros((len(slow), n), np.int32)
    x, y = (e.handle_preaccept(sl, sx, one, P, key[slow], is_set[slow], trp, deps0, None, everyone_else) for e in (gpu, ref))
    _same(x, y)
    assert (x[1] == everyone_else).all()                    # processed afresh everywhere (a higher ballot)
    two = np.zeros(len(slow), np.uint8)
    for j in range(f):
        two |= (1 << ((P + 1 + j) % n)).astype(np.uint8)
    x, y = (e.accept(sl, sx, one, P, trp, two, key[slow], is_set[slow]) for e in (gpu, ref))
    _same(x, y)
    assert x[5].all()                                       # f + 1 with the proposer: committed
    for r in range(n):
        for j in range(0, len(slow), 5):
            assert gpu.read_cmdlog(r, int(sl[j]), int(sx[j])) == ref.read_cmdlog(r, int(sl[j]), int(sx[j]))
        for k in range(K):
            for u, v in zip(gpu.read_index(r, k), ref.read_index(r, k)):
                np.testing.assert_array_equal(u, v)


@pytest.mark.gpu
def test_config4_size_properties_without_the_oracle(monkeypatch):
    """BASELINE.json configs[3] at size, the GPU alone: (1) a command's decision does not depend on where it stands
    in the tick's arrays (a replica's arrival order is the rank row, not the array order): the tick with its commands
    shuffled gives every command the same outputs and every replica the same conflict index; (2) the two forms of the
    tick (partition by key / radix sort of pairs, FPX_EPX_V1) are independent implementations of the same rule and
    agree on 2^20 commands; (3) a second tick on top gives the same outputs on both again."""
    from frankenpaxos_amd.epaxos import EPaxos

    n, num_keys, m = 5, 1024, 1 << 20
    a, b = EPaxos(n, num_keys), EPaxos(n, num_keys)
    monkeypatch.setenv("FPX_EPX_V1", "1")
    c = EPaxos(n, num_keys)
    monkeypatch.delenv("FPX_EPX_V1")
    rng = np.random.default_rng(2026)
    nxt = [0] * n
    for tick in range(2):
        leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 64.0, fifo=tick == 0)
        perm = rng.permutation(m)
        x = a.preaccept(leader, number, key, is_set, mask, rank)
        y = b.preaccept(leader[perm], number[perm], key[perm], is_set[perm], mask[perm], np.ascontiguousarray(rank[:, perm]))
        z = c.preaccept(leader, number, key, is_set, mask, rank)
        assert x[0] == y[0] == z[0] == 0
        assert 0 < int(x[1].sum()) < m
        for u, v, w in zip(x[1:], y[1:], z[1:]):
            np.testing.assert_array_equal(u[perm], v)
            np.testing.assert_array_equal(u, w)
    for r in range(n):
        for k in range(0, num_keys, 7):
            ia, ib, ic = a.read_index(r, k), b.read_index(r, k), c.read_index(r, k)
            assert ia[0].tolist() == ib[0].tolist() == ic[0].tolist()
            assert ia[1].tolist() == ib[1].tolist() == ic[1].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("n,num_keys,m", [(5, 64, 20000), (3, 16, 9000), (5, 1024, 60000)])
def test_epaxos_ticks_queued_back_to_back(oracle, n, num_keys, m):
    """Six ticks enqueued back to back without a host wait in between (the claim counters, the verdict word and the
    segments of the partition are reused from tick to tick) == the oracle, tick by tick; then a queue with a malformed
    tick in the middle: the ticks before it are applied, the tick and everything behind it are not (the status is
    sticky until fpx_epx_sync)."""
    import torch
    import frankenpaxos_amd as fa
    from frankenpaxos_amd.epaxos import EPaxos

    gpu, ref = EPaxos(n, num_keys), oracle.EPaxos(n, num_keys)
    rng = np.random.default_rng(n * 77 + num_keys)
    nxt = [0] * n
    dev = torch.device("cuda:0")
    up = lambda args: [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in args]
    host, devs, outs = [], [], []
    for tick in range(6):
        host.append(random_tick(rng, n, num_keys, m - 37 * tick, nxt, 9.0, fifo=tick % 2 == 0))
        devs.append(up(host[-1]))
        outs.append(torch.full((m - 37 * tick, gpu.packed_stride()), -7, dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    for t, o in zip(devs, outs):
        gpu.preaccept_packed_dev(*t, o)
    assert gpu.sync() == 0
    for h, o in zip(host, outs):
        want = ref.preaccept(*h)
        assert want[0] == 0
        fast, deps, ldeps, own = (x.cpu().numpy() for x in gpu.unpack(o))
        np.testing.assert_array_equal(fast, want[1].astype(np.int32))
        np.testing.assert_array_equal(deps, want[2])
        np.testing.assert_array_equal(ldeps, want[3])
        np.testing.assert_array_equal(own, want[4])
    # a malformed tick (a leader that does not exist) third in a queue of five
    host, devs, outs = [], [], []
    keep = None
    for tick in range(5):
        args = list(random_tick(rng, n, num_keys, m // 2, nxt, 9.0))
        if tick == 1:
            keep = list(nxt)                        # the instance numbers after the last tick that will be applied
        if tick == 2:
            args[0] = args[0].copy()
            args[0][m // 5] = n
        host.append(args)
        devs.append(up(args))
        outs.append(torch.full((m // 2, gpu.packed_stride()), -7, dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    for t, o in zip(devs, outs):
        gpu.preaccept_packed_dev(*t, o)
    assert gpu.sync() == fa.FPX_EINVAL
    for tick in range(5):
        if tick < 2:
            want = ref.preaccept(*host[tick])
            assert want[0] == 0
            fast, deps, _, _ = (x.cpu().numpy() for x in gpu.unpack(outs[tick]))
            np.testing.assert_array_equal(fast, want[1].astype(np.int32))
            np.testing.assert_array_equal(deps, want[2])
        else:
            assert bool((outs[tick] == -7).all()), "tick %d behind a rejected one was applied" % tick
    assert ref.preaccept(*host[2])[0] != 0
    # the context works on after the sync, from the state two applied ticks left
    args = random_tick(rng, n, num_keys, 3000, keep, 9.0)
    a, b = gpu.preaccept(*args), ref.preaccept(*args)
    assert a[0] == b[0] == 0
    for x, y in zip(a[1:], b[1:]):
        np.testing.assert_array_equal(x, y)
    for r in range(n):
        for k in range(min(num_keys, 40)):
            ga, sa = gpu.read_index(r, k)
            gb, sb = ref.read_index(r, k)
            assert ga.tolist() == gb.tolist() and sa.tolist() == sb.tolist()
