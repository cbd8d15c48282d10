"""fpx_epx_execute_dev -- dependency-graph execution on the device (csrc/fpx_depgraph_dev.hpp) against the host graph
(csrc/fpx_depgraph.cpp, itself pinned on the reference's DependencyGraphTest / ZigzagTarjanDependencyGraphTest vectors in
tests/test_depgraph.py): the same SET of strongly connected components, the same set of executables, and a valid
execution order (no dependency's component after its dependent's; inside a component by (leader, id)).  The reference
leaves the order of unrelated components to hash iteration (DependencyGraphTest.scala:188-191 accepts every outcome), so
the sequences themselves are not compared."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def labels(n, leader, number, order_leader, order_id, comp_of_pos):
    """canonical component label (the smallest member's key) of every executed vertex, as {key: label}"""
    key = order_leader.astype(np.int64) * (1 << 22) + order_id
    starts = np.nonzero(np.diff(np.concatenate([[-1], comp_of_pos])))[0]
    lab = np.minimum.reduceat(key, starts) if len(key) else key
    return dict(zip(key.tolist(), np.repeat(lab, np.diff(np.concatenate([starts, [len(key)]]))).tolist()))


def check_valid_order(n, first, leader, number, deps, own_end, order, comp):
    """every executed vertex: whatever it depends on (and is not executed earlier than `first`) sits in a component that does
    not come after its own; members of one component are neighbours, in (leader, id) order"""
    pos_comp = {}
    for p, i in enumerate(order):
        pos_comp[(int(leader[i]), int(number[i]))] = int(comp[p])
    assert np.all(np.diff(comp) >= 0) and (len(comp) == 0 or comp[0] == 0) and np.all(np.diff(comp) <= 1)
    for p in range(1, len(order)):
        if comp[p] == comp[p - 1]:
            a, b = order[p - 1], order[p]
            assert (leader[a], number[a]) < (leader[b], number[b])
    col_max = [np.full(int(number[leader == L].max(initial=first[L] - 1)) - first[L] + 1, -1, np.int64) for L in range(n)]
    for (L, x), c in pos_comp.items():
        col_max[L][x - first[L]] = c
    # prefix max of component numbers per column; an instance that did not execute poisons the prefix
    pm = []
    for L in range(n):
        c = np.where(col_max[L] < 0, np.iinfo(np.int64).max, col_max[L])
        pm.append(np.concatenate([[-1], np.maximum.accumulate(c)]))
    for p, i in enumerate(order):
        L, x = int(leader[i]), int(number[i])
        for l in range(n):
            w = int(deps[i, l]) if l != L else min(int(deps[i, l]), x)
            w = min(max(w - first[l], 0), len(pm[l]) - 1)
            assert pm[l][w] <= comp[p], (i, l)
        for y in range(x + 1, int(own_end[i])):
            assert 0 <= col_max[L][y - first[L]] <= comp[p]


def random_prefix_graph(rng, n, m, jitter, holes, from_zero=False):
    """m instances spread over n leaders, dense columns from random first ids; vertex (L, x) made at global time t depends on
    every column up to (that column's progress at t) +- jitter: jitter > 0 makes cycles"""
    leader = rng.integers(0, n, m).astype(np.int32)
    first = (np.zeros(n) if from_zero else rng.integers(0, 50, n)).astype(np.int32)
    number = np.zeros(m, np.int32)
    prog = np.zeros((m, n), np.int64)
    cnt = np.zeros(n, np.int64)
    for i in range(m):
        prog[i] = cnt
        number[i] = first[leader[i]] + cnt[leader[i]]
        cnt[leader[i]] += 1
    deps = np.zeros((m, n), np.int32)
    own = np.zeros((m, 2), np.int32)
    for i in range(m):
        L = leader[i]
        for l in range(n):
            w = first[l] + prog[i, l] + rng.integers(-jitter, jitter + 1)
            deps[i, l] = max(0, min(w, first[l] + cnt[l]))           # never beyond the column (everything is committed)
        x = number[i]
        if deps[i, L] > x:                                           # the own column never names the instance itself
            if holes and deps[i, L] > x + 1:
                own[i, 0] = deps[i, L]                               # explicit ids x + 1 .. end - 1, watermark x
            deps[i, L] = x
        if rng.random() < 0.1:
            deps[i] = np.minimum(deps[i], first + 3)                 # an old-looking instance: nearly no dependencies
            own[i, 0] = 0
            deps[i, L] = min(deps[i, L], x)
    return leader, number, first, cnt.astype(np.int32), deps, own


def run_device_only(n, leader, number, first, count, deps, own, committed=None):
    """the device call alone on resident inputs (a second call: buffers at their size), synchronised"""
    import torch
    from frankenpaxos_amd.epaxos import EPaxos

    m = len(leader)
    dev = torch.device("cuda:0")
    epx = EPaxos(n, 4)
    packed = np.zeros((m, epx.packed_stride()), np.int32)
    packed[:, :n] = deps
    packed[:, 2 * n] = own[:, 0]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    order, comp = torch.full((m,), -1, dtype=torch.int32, device=dev), torch.full((m,), -1, dtype=torch.int32, device=dev)
    args = (t(leader), t(number), t(packed), first, count, order, comp)
    cm = None if committed is None else t(committed.astype(np.uint8))
    epx.execute_dev(*args, committed=cm)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    out = epx.execute_dev(*args, committed=cm)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return out, dt, order.cpu().numpy()[:out[0]], comp.cpu().numpy()[:out[0]]


def run_both(n, leader, number, first, count, deps, own, committed=None, kind="tarjan"):
    import torch
    from frankenpaxos_amd import depgraph as P
    from frankenpaxos_amd.epaxos import EPaxos

    m = len(leader)
    dev = torch.device("cuda:0")
    epx = EPaxos(n, 4)
    stride = epx.packed_stride()
    packed = np.zeros((m, stride), np.int32)
    packed[:, :n] = deps
    packed[:, 2 * n] = own[:, 0]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    order, comp = torch.full((m,), -1, dtype=torch.int32, device=dev), torch.full((m,), -1, dtype=torch.int32, device=dev)
    ne, nc, nh = epx.execute_dev(t(leader), t(number), t(packed), first, count, order, comp,
                                 committed=None if committed is None else t(committed.astype(np.uint8)))
    order, comp = order.cpu().numpy()[:ne], comp.cpu().numpy()[:ne]
    # the host graph: everything below `first` executed, the committed vertices committed
    # (the zigzag variant walks its columns from the watermarks ITS OWN executions left, ZigzagTarjanDependencyGraph.scala:
    # 455-470: it serves the ticks that start at id 0; the plain variant takes an executed prefix as given)
    g = P.DependencyGraph(n, kind=P.FPX_DG_ZIGZAG if kind == "zigzag" else P.FPX_DG_TARJAN)
    if first.any():
        assert kind == "tarjan"
        g.update_executed(first)
    sel = np.ones(m, bool) if committed is None else committed.astype(bool)
    g.commit_epx(leader[sel], number[sel], deps[sel], own[sel])
    el, ei, cs, bl, bi = g.execute_arrays()
    return (ne, nc, nh, order, comp), (el, ei, cs)


@pytest.fixture(params=["packed", "wide"])
def dg_path(request, monkeypatch):
    """fpx_epx_execute_dev has two forms of its closure rounds: 16-byte rows with the prefix kept per workgroup
    (csrc/fpx_depgraph_pk.hpp; n <= 5) and 32-byte rows with a prefix pass per round (fpx_depgraph_dev.hpp; every n).
    FPX_DG_WIDE=1 sends everything the second way: results must not depend on it"""
    if request.param == "wide":
        monkeypatch.setenv("FPX_DG_WIDE", "1")
    else:
        monkeypatch.delenv("FPX_DG_WIDE", raising=False)
    return request.param


@pytest.mark.parametrize("n,m,jitter,holes", [(3, 300, 0, False), (5, 2000, 3, False), (5, 3000, 12, True), (7, 2500, 40, True),
                                              (3, 5000, 200, True), (5, 1, 0, False), (5, 40000, 6, True)])
def test_device_components_equal_the_host_graphs(n, m, jitter, holes, dg_path):
    rng = np.random.default_rng(n * 1000 + m + jitter)
    big = m > 10000
    leader, number, first, count, deps, own = random_prefix_graph(rng, n, m, jitter, holes, from_zero=big)
    (ne, nc, nh, order, comp), (el, ei, cs) = run_both(n, leader, number, first, count, deps, own, kind="zigzag" if big else "tarjan")
    assert not nh
    assert ne == m == len(el) and nc == len(cs)
    host = labels(n, leader, number, el, ei, np.repeat(np.arange(len(cs)), cs))
    mine = labels(n, leader, number, leader[order], number[order], comp)
    assert mine == host
    check_valid_order(n, first, leader, number, deps, own[:, 0], order, comp)
    if jitter >= 3:
        assert nc < m                                       # there were cycles


@pytest.mark.parametrize("n,m,dg_path", [(5, 3000, "packed"), (5, 3000, "wide"), (3, 800, "packed"), (3, 800, "wide")], indirect=["dg_path"])
def test_device_waits_for_what_is_not_committed(n, m, dg_path):
    """a tenth of the instances is not committed yet: they, and whatever reaches them, stay; the rest executes -- the same
    set the host graph executes"""
    rng = np.random.default_rng(m)
    leader, number, first, count, deps, own = random_prefix_graph(rng, n, m, 0, False)     # (no cycles: a giant component would wait whole)
    committed = rng.random(m) > 0.1
    committed[:3 * m // 4] = True                            # (the early instances are all there: a prefix executes)
    (ne, nc, nh, order, comp), (el, ei, cs) = run_both(n, leader, number, first, count, deps, own, committed)
    assert not nh and ne == len(el) and 0 < ne < m
    host = labels(n, leader, number, el, ei, np.repeat(np.arange(len(cs)), cs))
    mine = labels(n, leader, number, leader[order], number[order], comp)
    assert mine == host


def test_uncommitted_instances_at_size_do_not_walk_their_columns(monkeypatch):
    """an uncommitted vertex used to walk the rest of its column in k_dg_keys, one load per step -- O(m^2) per
    tick, seconds at this size; the walk is now asked of executable vertices only.  400 000 instances, 2.5 % of them not
    committed: the call stays in the tens of milliseconds, and the two forms of the closure rounds (16- and 32-byte rows)
    execute the same instances in the same components.  (The host graph takes minutes at this size with a committed mask;
    the small cases above hold the device to it.)"""
    n, m = 5, 400000
    rng = np.random.default_rng(m)
    leader, number, first, count, deps, own = random_prefix_graph(rng, n, m, 0, False)
    committed = rng.random(m) > 0.1
    committed[:3 * m // 4] = True
    monkeypatch.delenv("FPX_DG_WIDE", raising=False)
    (ne, nc, nh), dt, order, comp = run_device_only(n, leader, number, first, count, deps, own, committed)
    assert not nh and 3 * m // 4 <= ne < m and dt < 0.05, (ne, dt)
    assert committed[order].all()
    monkeypatch.setenv("FPX_DG_WIDE", "1")
    (ne2, nc2, nh2), dt2, order2, comp2 = run_device_only(n, leader, number, first, count, deps, own, committed)
    assert (ne2, nc2, nh2) == (ne, nc, nh) and dt2 < 0.05
    assert labels(n, leader, number, leader[order], number[order], comp) == labels(n, leader, number, leader[order2], number[order2], comp2)


def two_families_of_cycles(K):
    """n = 5: {(0, k), (1, k)} and {(2, k), (3, k)} are cycles of two for every k < K, each reaching everything before it in its
    own two columns: closures (k + 1, k + 1, 0, 0, 0) and (0, 0, k + 1, k + 1, 0) -- DIFFERENT closures with the SAME sum, i.e.
    the same sort key, K times over; column 4 is a chain of singletons"""
    n, m = 5, 5 * K
    leader = np.tile(np.arange(5, dtype=np.int32), K)
    number = np.repeat(np.arange(K, dtype=np.int32), 5)
    deps = np.zeros((m, n), np.int32)
    for i in range(m):
        L, k = int(leader[i]), int(number[i])
        deps[i, L] = k
        if L < 4:
            deps[i, L ^ 1] = k + 1
    return n, leader, number, np.zeros(n, np.int32), np.full(n, K, np.int32), deps, np.zeros((m, 2), np.int32)


def test_a_hash_collision_between_closures_of_one_key_is_reported(dg_path, monkeypatch):
    """needs_host_path = two different cyclic closures with one sort key AND one hash met in the sorted order
    (their members may interleave, so the call's components are not to be used and the caller takes fpx_depgraph_commit_epx
    + execute instead).  With the library's 22 hash bits that does not happen in any test; FPX_DG_HASH_BITS=2 leaves four
    hash values for 64 pairs of closures that share their keys: the flag must come up, on both row widths -- and with
    the full hash the same graph comes back as the host graph has it."""
    K = 64
    n, leader, number, first, count, deps, own = two_families_of_cycles(K)
    monkeypatch.delenv("FPX_DG_HASH_BITS", raising=False)
    (ne, nc, nh, order, comp), (el, ei, cs) = run_both(n, leader, number, first, count, deps, own)
    assert not nh and ne == 5 * K == len(el) and nc == 3 * K == len(cs)
    host = labels(n, leader, number, el, ei, np.repeat(np.arange(len(cs)), cs))
    assert labels(n, leader, number, leader[order], number[order], comp) == host
    check_valid_order(n, first, leader, number, deps, own[:, 0], order, comp)
    monkeypatch.setenv("FPX_DG_HASH_BITS", "2")
    (ne, nc, nh), _, _, _ = run_device_only(n, leader, number, first, count, deps, own)
    assert nh == 1                                          # (whatever else the call returned is not used)
    # a graph without cycles has nothing to group by the hash: no flag however few bits
    rng = np.random.default_rng(5)
    l2, n2, f2, c2, d2, o2 = random_prefix_graph(rng, 5, 2000, 0, False)
    (ne, nc, nh), _, order, comp = run_device_only(5, l2, n2, f2, c2, d2, o2)
    assert not nh and ne == nc == 2000


def test_device_refuses_columns_that_are_not_dense():
    import torch
    import frankenpaxos_amd as fa
    from frankenpaxos_amd.epaxos import EPaxos

    rng = np.random.default_rng(1)
    leader, number, first, count, deps, own = random_prefix_graph(rng, 5, 500, 2, False)
    number = number.copy()
    number[7] = number[8] if leader[7] == leader[8] else number[7] + 10_000        # twice, or outside its column
    dev = torch.device("cuda:0")
    epx = EPaxos(5, 4)
    packed = np.zeros((500, epx.packed_stride()), np.int32)
    packed[:, :5] = deps
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    o, c = torch.zeros(500, dtype=torch.int32, device=dev), torch.zeros(500, dtype=torch.int32, device=dev)
    with pytest.raises(fa.FpxError):
        epx.execute_dev(t(leader), t(number), t(packed), first, count, o, c)


@pytest.mark.parametrize("fifo,dg_path", [(True, "packed"), (False, "packed"), (False, "wide")], indirect=["dg_path"])
def test_config4_tick_executes_on_the_device(oracle, fifo, dg_path):
    """BASELINE.json configs[3] carried through on the device: a 2^20-command tick (n = 5, 1024 keys) pre-accepts (K5), and
    what it commits -- the agreed dependencies of the fast path, the union the slow path's Accept carries -- executes through
    fpx_epx_execute_dev: every instance once, the components of the host graph, a valid order; with reordering channels the
    own-column explicit ids take part."""
    import time
    import torch
    from frankenpaxos_amd import depgraph as P
    from frankenpaxos_amd.epaxos import EPaxos
    from tests import workloads as W
    from tests.workloads import random_tick

    n, num_keys, m = 5, 1024, 1 << 20
    dev = torch.device("cuda:0")
    epx = EPaxos(n, num_keys)
    rng = np.random.default_rng(45)
    nxt = [0] * n
    leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 64.0, fifo=fifo)
    key = (W.splitmix64_at(np.arange(m, dtype=np.uint64)) % np.uint64(num_keys)).astype(np.int32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dl, dn = t(leader), t(number)
    packed = torch.zeros((m, epx.packed_stride()), dtype=torch.int32, device=dev)
    epx.preaccept_packed_dev(dl, dn, t(key), t(is_set), t(mask), t(rank), packed)
    assert epx.sync() == 0
    order, comp = torch.zeros(m, dtype=torch.int32, device=dev), torch.zeros(m, dtype=torch.int32, device=dev)
    first, count = np.zeros(n, np.int32), np.asarray(nxt, np.int32)
    best = None
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ne, nc, nh = epx.execute_dev(dl, dn, packed, first, count, order, comp)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print("device dependency graph: %d commands, %d components, %.3f ms = %.3e commands/s" % (ne, nc, best * 1e3, ne / best))
    assert ne == m and not nh
    fast, deps, ldeps, own = (x.cpu().numpy() for x in epx.unpack(packed))
    if fifo:
        assert not own.any()
    else:
        assert own[:, 0].any()
    order, comp = order.cpu().numpy(), comp.cpu().numpy()
    g = P.DependencyGraph(n, kind=P.FPX_DG_ZIGZAG)
    g.commit_epx(leader, number, deps, own)
    el, ei, cs, bl, bi = g.execute_arrays()
    assert len(el) == m and nc == len(cs)
    host = labels(n, leader, number, el, ei, np.repeat(np.arange(len(cs)), cs))
    mine = labels(n, leader, number, leader[order], number[order], comp)
    assert mine == host
    from tests.test_epaxos import check_execution_order
    check_execution_order(n, leader, number, deps, own[:, 0], leader[order], number[order], np.bincount(comp))
    assert ne / best > (1.6e9 if dg_path == "packed" else 1e9)


@pytest.mark.parametrize("n,m,jitter", [(5, 3000, 12), (5, 3000, 0), (3, 900, 4), (7, 2500, 0)])
def test_host_array_entry_point_and_the_jni_native_execute_as_the_host_graph_does(n, m, jitter):
    """fpx_epx_execute (host arrays in, order out) and Native.epxExecute over it on the mock JVM -- what GpuEPaxosReplica
    calls with `deviceExecution` (jni/EPaxosNative.scala): the SET of components of the host graph, a valid execution
    order (cycles: jitter > 0, everything committed), instances without a Commit and what reaches them left waiting
    (jitter = 0: no cycles, a giant component would wait whole) -- and the same answer through both doors"""
    import ctypes as C
    from frankenpaxos_amd import depgraph as P
    from frankenpaxos_amd.epaxos import EPaxos
    from tests.test_jni_shim import build_shim

    rng = np.random.default_rng(77 + n + m + jitter)
    leader, number, first, count, deps, own = random_prefix_graph(rng, n, m, jitter, holes=jitter > 0)
    committed = np.ones(m, bool)
    if jitter == 0:
        committed = (np.arange(m) < 3 * m // 4) | (rng.random(m) > 0.1)     # some of the later instances have no Commit yet
    epx = EPaxos(n, 4)
    order, comp, nc, nh = epx.execute(leader, number, deps, first, count, deps_values_end=own[:, 0], committed=committed)
    assert not nh and 3 * m // 4 <= len(order) <= m and (len(order) < m) == (jitter == 0)
    g = P.DependencyGraph(n, kind=P.FPX_DG_TARJAN)
    g.update_executed(first)
    g.commit_epx(leader[committed], number[committed], deps[committed], own[committed])
    el, ei, cs, bl, bi = g.execute_arrays()
    assert len(order) == len(el) and nc == len(cs)
    assert labels(n, leader, number, leader[order], number[order], comp) == labels(n, leader, number, el, ei, np.repeat(np.arange(len(cs)), cs))
    if jitter > 0:
        check_valid_order(n, first, leader, number, deps, own[:, 0], order, comp)
    else:
        assert committed[order].all()
    # the native on the mock JVM: the same arrays as Java arrays
    L = C.CDLL(build_shim())
    L.mock_env.restype = C.c_void_p
    L.mock_new_array.restype = C.c_void_p
    L.mock_new_array.argtypes = [C.c_int, C.c_int64, C.c_void_p]
    L.mock_data.restype = C.c_void_p
    L.mock_data.argtypes = [C.c_void_p]
    env = C.c_void_p(L.mock_env())
    arr = lambda a: C.c_void_p(L.mock_new_array(np.ascontiguousarray(a).dtype.itemsize, np.ascontiguousarray(a).size, np.ascontiguousarray(a).ctypes.data))
    read = lambda o, k: np.ctypeslib.as_array(C.cast(L.mock_data(o), C.POINTER(C.c_int32)), (k,)).copy()
    j_order, j_comp, j_counts = arr(np.full(m, -1, np.int32)), arr(np.full(m, -1, np.int32)), arr(np.zeros(3, np.int32))
    fn = L.Java_frankenpaxos_gpu_Native_epxExecute
    fn.restype = C.c_int32
    st = fn(env, None, C.c_int64(epx._h if isinstance(epx._h, int) else epx._h.value), C.c_int32(m), C.c_int32(n), arr(leader), arr(number),
            arr(deps.astype(np.int32)), arr(own[:, 0].astype(np.int32)), arr(committed.astype(np.int8)), arr(first.astype(np.int32)),
            arr(count.astype(np.int32)), j_order, j_comp, j_counts)
    assert st == 0
    ne, nc2, nh2 = read(j_counts, 3)
    assert (ne, nc2, nh2) == (len(order), nc, 0)
    np.testing.assert_array_equal(read(j_order, m)[:ne], order)
    np.testing.assert_array_equal(read(j_comp, m)[:ne], comp)
    # short arrays are refused before anything is touched
    assert fn(env, None, C.c_int64(epx._h if isinstance(epx._h, int) else epx._h.value), C.c_int32(m), C.c_int32(n), arr(leader[:-1]), arr(number),
              arr(deps.astype(np.int32)), None, None, arr(first.astype(np.int32)), arr(count.astype(np.int32)), j_order, j_comp, j_counts) == 1
