"""Pins the CPU oracle against every known-answer test the reference holds for this path
(SURVEY.md section 8c).  Each test names the reference test it transcribes; paths are relative to
/root/reference/shared/src/test/scala/.

These are the ONLY golden vectors in the reference for the hot path: the quorum predicates (a5),
the round system (a7) and the replica log container (f1).  CPU only.
"""
import itertools

import pytest


# ---------------------------------------------------------------- quorums/GridTest.scala ------
def grid23(oracle):
    return oracle.QuorumSystem.grid([[1, 2, 3], [4, 5, 6]])


def test_grid_read_quorums(oracle):
    """quorums/GridTest.scala:11-27"""
    qs = grid23(oracle)
    assert qs.is_read_quorum([]) is False
    for i in range(1, 7):
        assert qs.is_read_quorum([i]) is False
    for i, j in itertools.product(range(1, 7), repeat=2):
        assert qs.is_read_quorum([i, j]) is False
    assert qs.is_read_quorum([1, 2, 4]) is False
    assert qs.is_read_quorum([4, 5, 3]) is False
    assert qs.is_read_quorum([1, 2, 3]) is True
    assert qs.is_read_quorum([4, 5, 6]) is True
    assert qs.is_read_quorum([1, 2, 3, 4]) is True
    assert qs.is_read_quorum([1, 2, 3, 4, 5]) is True
    assert qs.is_read_quorum([1, 2, 3, 4, 5, 6]) is True


GRID_WRITE_CASES = [
    ([], False), ([1, 2], False), ([1, 2, 3], False), ([4, 5], False), ([4, 5, 6], False),
    ([1, 4], True), ([2, 4], True), ([2, 5], True), ([1, 2, 4], True), ([1, 2, 4, 5], True),
    ([1, 2, 3, 4, 5], True), ([1, 2, 3, 4, 5, 6], True),
]


def test_grid_write_quorums(oracle):
    """quorums/GridTest.scala:29-46"""
    qs = grid23(oracle)
    for i in range(1, 7):
        assert qs.is_write_quorum([i]) is False
    for xs, want in GRID_WRITE_CASES:
        assert qs.is_write_quorum(xs) is want, xs


def test_grid_read_quorum_supersets(oracle):
    """quorums/GridTest.scala:48-72"""
    qs = grid23(oracle)
    assert qs.is_superset_of_read_quorum([]) is False
    for i in range(1, 7):
        assert qs.is_superset_of_read_quorum([i]) is False
    for i, j in itertools.product(range(1, 7), repeat=2):
        assert qs.is_superset_of_read_quorum([i, j]) is False
    cases = [([1, 2, 4], False), ([4, 5, 3], False), ([1, 2, 3], True), ([4, 5, 6], True),
             ([1, 2, 3, 4], True), ([1, 2, 3, 4, 5], True), ([1, 2, 3, 4, 5, 6], True)]
    for xs, want in cases:
        assert qs.is_superset_of_read_quorum(xs) is want
        assert qs.is_superset_of_read_quorum([9001] + xs) is want


def test_grid_write_quorum_supersets(oracle):
    """quorums/GridTest.scala:74-102"""
    qs = grid23(oracle)
    for i in range(1, 7):
        assert qs.is_superset_of_write_quorum([i]) is False
    for xs, want in GRID_WRITE_CASES:
        assert qs.is_superset_of_write_quorum(xs) is want
        if xs:
            assert qs.is_superset_of_write_quorum([9001] + xs) is want


def test_grid_require_on_foreign_node(oracle):
    """Grid.scala:37-40,44-47: require(xs.subsetOf(nodes))"""
    qs = grid23(oracle)
    with pytest.raises(ValueError):
        qs.is_write_quorum([9001, 1, 4])
    with pytest.raises(ValueError):
        qs.is_read_quorum([9001, 1, 2, 3])


# ------------------------------------------------------ quorums/SimpleMajorityTest.scala ------
def test_simple_majority_quorums(oracle):
    """quorums/SimpleMajorityTest.scala:11-29"""
    qs = oracle.QuorumSystem.simple_majority([0, 1, 2, 3, 4])
    cases = [([], False), ([0], False), ([0, 1], False), ([0, 1, 2], True), ([0, 1, 2, 3], True),
             ([0, 1, 2, 3, 4], True)]
    for xs, want in cases:
        assert qs.is_read_quorum(xs) is want
        assert qs.is_write_quorum(xs) is want


def test_simple_majority_supersets(oracle):
    """quorums/SimpleMajorityTest.scala:31-63"""
    qs = oracle.QuorumSystem.simple_majority([0, 1, 2, 3, 4])
    cases = [([], False), ([0], False), ([0, 1], False), ([0, 1, 2], True), ([0, 1, 2, 3], True),
             ([0, 1, 2, 3, 4], True), ([5], False), ([0, 5], False), ([0, 1, 5], False),
             ([0, 1, 2, 5], True), ([0, 1, 2, 3, 5], True), ([0, 1, 2, 3, 4, 5], True)]
    for xs, want in cases:
        assert qs.is_superset_of_read_quorum(xs) is want
        assert qs.is_superset_of_write_quorum(xs) is want


# -------------------------------------------------------- quorums/UnanimousWrites.scala ------
def test_unanimous_writes_quorums(oracle):
    """quorums/UnanimousWrites.scala:11-33"""
    qs = oracle.QuorumSystem.unanimous_writes([0, 1, 2, 3, 4])
    assert qs.is_read_quorum([]) is False
    for xs in ([0], [1], [2], [3], [4], [0, 1], [0, 1, 2], [0, 1, 2, 3], [0, 1, 2, 3, 4]):
        assert qs.is_read_quorum(xs) is True
    for xs in ([], [0], [0, 1], [0, 1, 2], [0, 1, 2, 3]):
        assert qs.is_write_quorum(xs) is False
    assert qs.is_write_quorum([0, 1, 2, 3, 4]) is True


def test_unanimous_writes_supersets(oracle):
    """quorums/UnanimousWrites.scala:35-75"""
    qs = oracle.QuorumSystem.unanimous_writes([0, 1, 2, 3, 4])
    assert qs.is_superset_of_read_quorum([]) is False
    for xs in ([0], [1], [2], [3], [4], [0, 1], [0, 1, 2], [0, 1, 2, 3], [0, 1, 2, 3, 4]):
        assert qs.is_superset_of_read_quorum(xs) is True
    assert qs.is_superset_of_read_quorum([5]) is False
    for xs in ([0, 5], [1, 5], [2, 5], [3, 5], [4, 5], [0, 1, 5], [0, 1, 2, 5], [0, 1, 2, 3, 5],
               [0, 1, 2, 3, 4, 5]):
        assert qs.is_superset_of_read_quorum(xs) is True
    for xs in ([], [0], [0, 1], [0, 1, 2], [0, 1, 2, 3]):
        assert qs.is_superset_of_write_quorum(xs) is False
    assert qs.is_superset_of_write_quorum([0, 1, 2, 3, 4]) is True
    for xs in ([5], [0, 5], [0, 1, 5], [0, 1, 2, 5], [0, 1, 2, 3, 5]):
        assert qs.is_superset_of_write_quorum(xs) is False
    assert qs.is_superset_of_write_quorum([0, 1, 2, 3, 4, 5]) is True


# ------------------------------------------------------ quorums/QuorumSystemTest.scala ------
def all_systems(oracle):
    """quorums/QuorumSystemTest.scala:11-31: majority n in 1..9, unanimous n in 1..9, grids 2..5 x 2..5"""
    for i in range(1, 10):
        yield "SimpleMajority %d" % i, oracle.QuorumSystem.simple_majority(range(i))
    for i in range(1, 10):
        yield "UnanimousWrites %d" % i, oracle.QuorumSystem.unanimous_writes(range(i))
    for rows in range(2, 6):
        for cols in range(2, 6):
            ids = list(range(rows * cols))
            yield ("Grid %dx%d" % (rows, cols),
                   oracle.QuorumSystem.grid([ids[r * cols:(r + 1) * cols] for r in range(rows)]))


def test_read_and_write_quorums_intersect(oracle):
    """quorums/QuorumSystemTest.scala:33-47"""
    rng = [0xC0FFEE]
    for name, qs in all_systems(oracle):
        for _ in range(100):
            r, w = qs.random_read_quorum(rng), qs.random_write_quorum(rng)
            assert r & w, (name, r, w)


def test_random_read_quorums_are_read_quorums(oracle):
    """quorums/QuorumSystemTest.scala:49-59"""
    rng = [1]
    for name, qs in all_systems(oracle):
        for _ in range(100):
            q = qs.random_read_quorum(rng)
            assert qs.is_read_quorum(q), (name, q)
            assert qs.is_superset_of_read_quorum(q), (name, q)


def test_random_write_quorums_are_write_quorums(oracle):
    """quorums/QuorumSystemTest.scala:61-72"""
    rng = [2]
    for name, qs in all_systems(oracle):
        for _ in range(100):
            q = qs.random_write_quorum(rng)
            assert qs.is_write_quorum(q), (name, q)
            assert qs.is_superset_of_write_quorum(q), (name, q)


# -------------------------------------------------- roundsystem/RoundSystemTest.scala ------
def test_classic_round_robin_leader(oracle):
    """roundsystem/RoundSystemTest.scala:13-24"""
    L = oracle.lib()
    for rnd, want in enumerate([0, 1, 2, 0, 1, 2, 0, 1, 2]):
        assert L.fpo_round_leader(3, rnd) == want


NEXT_CLASSIC_ROUND = {
    # roundsystem/RoundSystemTest.scala:33-62: leader -> expected for round = -1..6
    0: [0, 3, 3, 3, 6, 6, 6, 9],
    1: [1, 1, 4, 4, 4, 7, 7, 7],
    2: [2, 2, 2, 5, 5, 5, 8, 8],
}


def test_classic_round_robin_next_classic_round(oracle):
    L = oracle.lib()
    for leader, wants in NEXT_CLASSIC_ROUND.items():
        for rnd, want in zip(range(-1, 7), wants):
            assert L.fpo_next_classic_round(3, leader, rnd) == want, (leader, rnd)


# --------------------------------------------------------- util/BufferMapTest.scala ------
def test_buffer_map_put_get(oracle):
    """util/BufferMapTest.scala:7-37"""
    m = oracle.Log(10)
    assert m.get(0) is None
    m.put(0, 0)
    assert m.get(0) == 0
    for grow in (10, 0):
        m = oracle.Log(grow)
        assert m.get(100) is None
        m.put(100, 100)
        assert m.get(100) == 100
    m = oracle.Log(10)
    m.put(100, 100)
    assert m.get(1000) is None
    m.put(1000, 1000)
    assert m.get(100) == 100 and m.get(1000) == 1000


def test_buffer_map_garbage_collect(oracle):
    """util/BufferMapTest.scala:39-119"""
    m = oracle.Log(10)
    m.put(0, 0)
    m.garbage_collect(0)
    assert m.get(0) == 0

    def four():
        m = oracle.Log(10)
        for k in range(4):
            m.put(k, k)
        return m

    m = four()
    m.garbage_collect(2)
    assert [m.get(k) for k in range(4)] == [None, None, 2, 3]
    m = four()
    m.garbage_collect(1)
    assert [m.get(k) for k in range(4)] == [None, 1, 2, 3]
    m.garbage_collect(3)
    assert [m.get(k) for k in range(4)] == [None, None, None, 3]
    m = four()
    m.garbage_collect(3)
    assert [m.get(k) for k in range(4)] == [None, None, None, 3]
    m.garbage_collect(1)
    assert [m.get(k) for k in range(4)] == [None, None, None, 3]
    m = oracle.Log(10)
    m.garbage_collect(100)
    m.put(200, 200)
    assert m.get(200) == 200
    m = oracle.Log(10)
    m.put(10, 10)
    m.put(20, 20)
    m.garbage_collect(15)
    m.put(30, 30)
    m.put(40, 40)
    m.garbage_collect(35)
    m.put(50, 50)
    m.put(60, 60)
    assert (m.get(40), m.get(50), m.get(60)) == (40, 50, 60)


def test_buffer_map_iterators(oracle):
    """util/BufferMapTest.scala:121-260"""
    m = oracle.Log(10)
    assert m.iterator_from(0) == []
    m.put(0, 0)
    assert m.iterator_from(0) == [(0, 0)]
    assert m.iterator_from(1) == [] and m.iterator_from(2) == []
    m = oracle.Log(10)
    m.put(10, 10)
    for k in (0, 1, 9, 10):
        assert m.iterator_from(k) == [(10, 10)]
    assert m.iterator_from(11) == [] and m.iterator_from(12) == []
    m = oracle.Log(10)
    keys = [0, 5, 6, 7, 10, 20]
    for k in keys:
        m.put(k, k)
    for start in (0, 1, 4, 5, 6, 7, 8, 10, 11, 20, 21):
        assert m.iterator_from(start) == [(k, k) for k in keys if k >= start]
    m = oracle.Log(10)
    keys = [0, 5, 6, 7, 10, 15, 20]
    for k in keys:
        m.put(k, k)
    m.garbage_collect(10)
    for start in (0, 1, 9, 10, 11, 15, 16, 20, 21):
        assert m.iterator_from(start) == [(k, k) for k in keys if k >= max(start, 10)]


def test_replica_log_prefix(oracle):
    """multipaxos/Replica.scala:572-590 + 394-404: Chosen -> log.put -> execute contiguous prefix."""
    log = oracle.Log(8)
    assert log.chosen(1, 11) == 0          # hole at 0
    assert log.chosen(2, 12) == 0
    assert log.chosen(0, 10) == 3          # prefix 0..2 executes
    assert log.chosen(0, 99) == 3          # redundantly chosen: ignored (:580-584)
    assert log.get(0) == 10
    assert log.chosen(4, 14) == 3
    assert log.chosen(3, 13) == 5
    assert log.executed_watermark == 5
