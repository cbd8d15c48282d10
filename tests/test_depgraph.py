"""dependency-graph execution (depgraph/TarjanDependencyGraph.scala,
depgraph/ZigzagTarjanDependencyGraph.scala; driven by epaxos/Replica.scala:859-917).

1. every known-answer test of the reference's own
   shared/src/test/scala/depgraph/DependencyGraphTest.scala and ZigzagTarjanDependencyGraphTest.scala, transcribed,
   run against BOTH the oracle (oracle/depgraph.py) and the product (libfpx.so through the C ABI of
   include/fpx_depgraph.h);
2. the reference's property test ("All dep graph implementations should agree", DependencyGraphTest.scala:331-462)
   with an independent SCC computation (scipy) standing in for the Jgrapht / ScalaGraph implementations;
3. product == oracle, bit for bit (executables, their order, component boundaries, blockers), on random graphs in
   both variants, incremental commits, updateExecuted, numBlockers, InstancePrefixSet dependencies with explicit ids.

All of this is host code: it runs without a GPU.  (The GPU tick -> commit -> execute chain is in test_epaxos.py.)"""
import random

import numpy as np
import pytest

from oracle import depgraph as O


@pytest.fixture(scope="module")
def P():
    import os

    import frankenpaxos_amd
    from frankenpaxos_amd import depgraph

    if not os.path.exists(frankenpaxos_amd._lib.SO_PATH):
        frankenpaxos_amd.build()
    return depgraph


# ---------------------------------------------------------------------------------------------------------------
# adapters: one face for the oracle and the product, in the vocabulary of the reference's tests
# ---------------------------------------------------------------------------------------------------------------
class IntGraph:
    """DependencyGraph[Int, Int, IntPrefixSet] as DependencyGraphTest builds it"""

    def __init__(self, impl, P=None):
        self.impl = impl
        if impl == "oracle":
            self.g = O.TarjanDependencyGraph(O.IntSetAsKeys())
        else:
            self.g = P.DependencyGraph(1, kind=P.FPX_DG_TARJAN)

    def commit(self, key, seq, deps):
        """deps: a python set of ints = IntPrefixSet(Set(...))"""
        if self.impl == "oracle":
            self.g.commit(key, seq, O.IntSetAsKeys.of(deps))
        else:
            self.g.commit([0], [key], [seq], [[0]], [[(0, d) for d in deps]])

    def update_executed(self, keys):
        if self.impl == "oracle":
            self.g.update_executed(O.IntSetAsKeys.of(keys))
        else:
            self.g.update_executed(None, [(0, k) for k in keys])

    def execute_by_component(self, num_blockers=None):
        if self.impl == "oracle":
            return self.g.execute_by_component(num_blockers)
        comps, blockers = self.g.execute_by_component(num_blockers)
        return [[k[1] for k in c] for c in comps], {b[1] for b in blockers}

    def ebc(self, num_blockers=None):
        return self.execute_by_component(num_blockers)[0]


class Zigzag:
    """ZigzagTarjanDependencyGraph[(Int, Int), Int, FakeCompactSet[(Int, Int)]], numLeaders = 3"""

    def __init__(self, impl, P=None, gc_every=100):
        self.impl = impl
        if impl == "oracle":
            self.g = O.ZigzagTarjanDependencyGraph(O.FakeCompactSet(), 3, vertices_grow_size=10,
                                                   garbage_collect_every_n_commands=gc_every)
        else:
            self.g = P.DependencyGraph(3, kind=P.FPX_DG_ZIGZAG, gc_every_n=gc_every)

    def commit(self, key, seq, deps):
        if self.impl == "oracle":
            self.g.commit(key, seq, O.FakeCompactSet(deps))
        else:
            self.g.commit([key[0]], [key[1]], [seq], [[0, 0, 0]], [sorted(deps)])

    def update_executed(self, keys):
        if self.impl == "oracle":
            self.g.update_executed(O.FakeCompactSet(keys))
        else:
            self.g.update_executed(None, sorted(keys))

    def execute_by_component(self):
        comps, blockers = self.g.execute_by_component(None)
        return [list(c) for c in comps], set(blockers)


IMPLS = ["oracle", "product"]


@pytest.fixture(params=IMPLS)
def tarjan(request, P):
    return lambda: IntGraph(request.param, P)


@pytest.fixture(params=IMPLS)
def zigzag(request, P):
    return lambda gc_every=100: Zigzag(request.param, P, gc_every)


# ---------------------------------------------------------------------------------------------------------------
# DependencyGraphTest.scala, "A dep graph should ..." (the `tarjan` instances of every test(...) call)
# ---------------------------------------------------------------------------------------------------------------
def test_ref_commit_a_command_with_no_dependencies(tarjan):  # :29-38
    g = tarjan()
    g.commit(0, 0, set())
    assert g.ebc() == [[0]]
    assert g.ebc() == []


def test_ref_ignore_repeated_commands(tarjan):  # :40-53
    g = tarjan()
    g.commit(0, 0, set())
    assert g.ebc() == [[0]]
    g.commit(0, 1, set())
    assert g.ebc() == []
    g.commit(0, 2, {1})
    assert g.ebc() == []


def test_ref_chain_of_commands(tarjan):  # :55-72
    g = tarjan()
    g.commit(0, 0, set())
    assert g.ebc() == [[0]]
    g.commit(1, 0, {0})
    assert g.ebc() == [[1]]
    g.commit(2, 0, {1})
    assert g.ebc() == [[2]]
    g.commit(3, 0, {2})
    assert g.ebc() == [[3]]
    assert g.ebc() == []


@pytest.mark.parametrize("seqs", [(0, 0, 0, 0), (0, 1, 2, 3)])  # :74-94, :96-116 (with sequence numbers)
def test_ref_reverse_chain_of_commands(tarjan, seqs):
    g = tarjan()
    g.commit(3, seqs[0], {2})
    assert g.ebc() == []
    g.commit(2, seqs[1], {1})
    assert g.ebc() == []
    g.commit(1, seqs[2], {0})
    assert g.ebc() == []
    g.commit(0, seqs[3], set())
    assert g.ebc() == [[0], [1], [2], [3]]
    assert g.ebc() == []


def test_ref_two_cycle(tarjan):  # :118-129
    g = tarjan()
    g.commit(0, 0, {1})
    assert g.ebc() == []
    g.commit(1, 0, {0})
    assert g.ebc() == [[0, 1]]


def test_ref_two_cycle_with_sequence_numbers(tarjan):  # :131-142
    g = tarjan()
    g.commit(0, 1, {1})
    assert g.ebc() == []
    g.commit(1, 0, {0})
    assert g.ebc() == [[1, 0]]


def test_ref_three_cycle(tarjan):  # :144-157
    g = tarjan()
    g.commit(0, 0, {1})
    assert g.ebc() == []
    g.commit(1, 0, {2})
    assert g.ebc() == []
    g.commit(2, 0, {0})
    assert g.ebc() == [[0, 1, 2]]


def test_ref_three_cycle_with_sequence_numbers(tarjan):  # :159-172
    g = tarjan()
    g.commit(0, 1, {1})
    assert g.ebc() == []
    g.commit(1, 0, {2})
    assert g.ebc() == []
    g.commit(2, 2, {0})
    assert g.ebc() == [[1, 0, 2]]


def test_ref_complex_graph_in_order(tarjan):  # :174-205
    g = tarjan()
    g.commit(0, 0, set())
    assert g.ebc() == [[0]]
    g.commit(1, 0, {0, 2})
    assert g.ebc() == []
    g.commit(2, 1, {1})
    assert g.ebc() == [[1, 2]]
    g.commit(3, 0, {1, 2})
    assert g.ebc() == [[3]]
    g.commit(4, 0, {2})
    assert g.ebc() == [[4]]
    g.commit(5, 0, {3, 4, 6})
    assert g.ebc() == []
    g.commit(6, 1, {4, 5})
    assert g.ebc() == [[5, 6]]


def test_ref_complex_graph_in_reverse_order(tarjan):  # :207-233
    g = tarjan()
    g.commit(6, 1, {4, 5})
    assert g.ebc() == []
    g.commit(5, 0, {3, 4, 6})
    assert g.ebc() == []
    g.commit(4, 0, {2})
    assert g.ebc() == []
    g.commit(3, 0, {1, 2})
    assert g.ebc() == []
    g.commit(2, 1, {1})
    assert g.ebc() == []
    g.commit(1, 0, {0, 2})
    assert g.ebc() == []
    g.commit(0, 0, set())
    assert g.ebc() in ([[0], [1, 2], [3], [4], [5, 6]], [[0], [1, 2], [4], [3], [5, 6]])


def test_ref_complex_graph_in_random_order(tarjan):  # :235-256
    g = tarjan()
    g.commit(6, 1, {4, 5})
    assert g.ebc() == []
    g.commit(4, 0, {2})
    assert g.ebc() == []
    g.commit(0, 0, set())
    assert g.ebc() == [[0]]
    g.commit(2, 1, {1})
    assert g.ebc() == []
    g.commit(5, 0, {3, 4, 6})
    assert g.ebc() == []
    g.commit(1, 0, {0, 2})
    assert g.ebc() == [[1, 2], [4]]
    g.commit(3, 0, {1, 2})
    assert g.ebc() == [[3], [5, 6]]


def test_ref_hard_tarjan_test_case(tarjan):  # :258-281
    g = tarjan()
    g.commit(0, 0, {3, 1})
    g.commit(1, 1, {2})
    g.commit(2, 2, {1})
    g.commit(3, 3, {4})
    g.commit(4, 4, {2, 3, 5, 6, 7, 8})
    g.commit(5, 5, {6})
    g.commit(6, 6, set())
    g.commit(7, 7, {4})
    g.commit(8, 8, {7})
    assert g.ebc() in (
        [[1, 2], [6], [5], [3, 4, 7, 8], [0]],
        [[6], [1, 2], [5], [3, 4, 7, 8], [0]],
        [[6], [5], [1, 2], [3, 4, 7, 8], [0]],
    )


def test_ref_simple_update_executed(tarjan):  # :283-296
    g = tarjan()
    g.commit(1, 1, {0})
    assert g.ebc() == []
    g.update_executed({0})
    assert g.ebc() == [[1]]


def test_ref_chain_of_update_executed(tarjan):  # :298-313
    g = tarjan()
    g.commit(1, 1, {0})
    g.commit(2, 2, {1})
    g.commit(3, 3, {2})
    assert g.ebc() == []
    g.update_executed({0})
    assert g.ebc() == [[1], [2], [3]]


def test_ref_star_of_update_executed(tarjan):  # :315-337
    g = tarjan()
    g.commit(1, 1, {0})
    g.commit(2, 2, {0})
    g.commit(3, 3, {0})
    assert g.ebc() == []
    g.update_executed({0})
    out = g.ebc()
    assert sorted(out) == [[1], [2], [3]] and len(out) == 3


def test_ref_update_executed_on_a_command_in_the_graph(tarjan):  # :339-354
    g = tarjan()
    g.commit(1, 1, {0})
    g.commit(2, 2, {1})
    g.commit(3, 3, {2})
    assert g.ebc() == []
    g.update_executed({0, 1})
    assert g.ebc() == [[2], [3]]


def test_ref_complex_update_executed_example(tarjan):  # :356-373
    g = tarjan()
    g.commit(4, 4, {0, 1, 5})
    g.commit(5, 5, {4, 2})
    g.commit(6, 6, {7})
    g.commit(7, 7, {3, 5, 6})
    assert g.ebc() == []
    g.update_executed({0, 1, 2, 3, 4, 5})
    assert g.ebc() == [[6, 7]]


# "A TarjanDependencyGraph graph should ..." :465-545
@pytest.mark.parametrize("num_blockers", [None, 1, 10])
def test_ref_tarjan_report_no_blockers(tarjan, num_blockers):  # :465-476
    g = tarjan()
    g.commit(0, 0, set())
    g.commit(1, 1, {0})
    g.commit(2, 2, {1})
    assert g.execute_by_component(num_blockers) == ([[0], [1], [2]], set())


def test_ref_tarjan_report_all_blockers(tarjan):  # :478-486
    g = tarjan()
    g.commit(0, 0, {10})
    g.commit(1, 1, {20})
    g.commit(2, 2, {30})
    assert g.execute_by_component(None) == ([], {10, 20, 30})


def test_ref_tarjan_report_blockers_chain(tarjan):  # :488-496
    g = tarjan()
    g.commit(0, 0, {1})
    g.commit(1, 1, {2})
    g.commit(2, 2, {3})
    assert g.execute_by_component(None) == ([], {3})


@pytest.mark.parametrize("num_blockers,allowed", [
    (1, [{10}, {20}, {30}]),                       # :498-506
    (2, [{10, 20}, {10, 30}, {20, 30}]),           # :508-516
    (3, [{10, 20, 30}]), (4, [{10, 20, 30}]), (100, [{10, 20, 30}]),  # :518-528
])
def test_ref_tarjan_report_n_blockers(tarjan, num_blockers, allowed):
    g = tarjan()
    g.commit(0, 0, {10})
    g.commit(1, 1, {20})
    g.commit(2, 2, {30})
    executables, blockers = g.execute_by_component(num_blockers)
    assert executables == [] and blockers in allowed


def test_ref_tarjan_report_one_shared_blocker(tarjan):  # :530-538
    g = tarjan()
    g.commit(0, 0, {10})
    g.commit(1, 1, {10})
    g.commit(2, 2, {10})
    assert g.execute_by_component(None) == ([], {10})


# ---------------------------------------------------------------------------------------------------------------
# ZigzagTarjanDependencyGraphTest.scala
# ---------------------------------------------------------------------------------------------------------------
def test_ref_zigzag_execute_no_commands(zigzag):  # :52-55
    g = zigzag()
    assert g.execute_by_component() == ([], {(0, 0), (1, 0), (2, 0)})


def test_ref_zigzag_execute_one_command(zigzag):  # :57-62
    g = zigzag()
    g.commit((0, 0), 0, set())
    assert g.execute_by_component() == ([[(0, 0)]], {(0, 1), (1, 0), (2, 0)})


def test_ref_zigzag_chain_of_commands(zigzag):  # :64-82
    g = zigzag()
    g.commit((0, 0), 0, set())
    g.commit((1, 0), 0, {(0, 0)})
    g.commit((2, 0), 0, {(1, 0)})
    g.commit((0, 1), 0, {(2, 0)})
    g.commit((1, 1), 0, {(0, 1)})
    g.commit((2, 1), 0, {(1, 1)})
    assert g.execute_by_component() == (
        [[(0, 0)], [(1, 0)], [(2, 0)], [(0, 1)], [(1, 1)], [(2, 1)]], {(0, 2), (1, 2), (2, 2)})


def test_ref_zigzag_region_with_back_edges(zigzag):  # :84-102
    g = zigzag()
    g.commit((0, 0), 0, set())
    g.commit((1, 0), 0, {(0, 0)})
    g.commit((2, 0), 0, {(1, 0), (0, 0)})
    g.commit((0, 1), 0, {(2, 0)})
    g.commit((1, 1), 0, set())
    g.commit((2, 1), 0, {(1, 1), (2, 0)})
    assert g.execute_by_component() == (
        [[(0, 0)], [(1, 0)], [(2, 0)], [(0, 1)], [(1, 1)], [(2, 1)]], {(0, 2), (1, 2), (2, 2)})


def test_ref_zigzag_forward_edge(zigzag):  # :104-110
    g = zigzag()
    g.commit((0, 0), 0, {(1, 0)})
    g.commit((1, 0), 0, set())
    assert g.execute_by_component() == ([[(1, 0)], [(0, 0)]], {(0, 1), (1, 1), (2, 0)})


def test_ref_zigzag_cycle(zigzag):  # :112-118
    g = zigzag()
    g.commit((0, 0), 0, {(1, 0)})
    g.commit((1, 0), 1, {(0, 0)})
    assert g.execute_by_component() == ([[(0, 0), (1, 0)]], {(0, 1), (1, 1), (2, 0)})


def test_ref_zigzag_forward_edge_with_gap(zigzag):  # :120-131
    g = zigzag()
    g.commit((0, 0), 0, {(2, 0)})
    g.commit((2, 0), 1, set())
    assert g.execute_by_component() == ([[(2, 0)], [(0, 0)]], {(0, 1), (1, 0), (2, 1)})
    g.commit((1, 0), 0, set())
    g.commit((0, 1), 1, set())
    assert g.execute_by_component() == ([[(0, 1)], [(1, 0)]], {(0, 2), (1, 1), (2, 1)})


def test_ref_zigzag_garbage_collect(zigzag):  # :133-149
    g = zigzag(gc_every=3)
    g.commit((0, 0), 0, set())
    g.commit((1, 0), 0, set())
    g.commit((2, 0), 0, set())
    assert g.execute_by_component() == ([[(0, 0)], [(1, 0)], [(2, 0)]], {(0, 1), (1, 1), (2, 1)})
    g.commit((0, 1), 0, {(0, 0), (1, 0), (2, 0)})
    assert g.execute_by_component() == ([[(0, 1)]], {(0, 2), (1, 1), (2, 1)})


def test_ref_zigzag_unmet_dep(zigzag):  # :151-163
    g = zigzag()
    g.commit((0, 0), 0, {(0, 1)})
    assert g.execute_by_component() == ([], {(0, 1), (1, 0), (2, 0)})
    g.commit((0, 1), 0, {(0, 2)})
    assert g.execute_by_component() == ([], {(0, 2), (1, 0), (2, 0)})
    g.commit((0, 2), 0, {(0, 3)})
    assert g.execute_by_component() == ([], {(0, 3), (1, 0), (2, 0)})
    g.commit((0, 3), 0, set())
    assert g.execute_by_component() == ([[(0, 3)], [(0, 2)], [(0, 1)], [(0, 0)]], {(0, 4), (1, 0), (2, 0)})


def test_ref_zigzag_tall_column(zigzag):  # :165-176
    g = zigzag()
    g.commit((0, 0), 0, set())
    for i in range(4):
        g.commit((1, i), 0, set())
    assert g.execute_by_component() == (
        [[(0, 0)], [(1, 0)], [(1, 1)], [(1, 2)], [(1, 3)]], {(0, 1), (1, 4), (2, 0)})


def test_ref_zigzag_update_executed(zigzag):  # :178-187
    g = zigzag()
    g.commit((0, 0), 0, {(0, 10)})
    g.commit((1, 0), 0, {(0, 10)})
    g.commit((2, 0), 0, {(0, 10)})
    assert g.execute_by_component() == ([], {(0, 10)})
    g.update_executed({(0, 0), (1, 0), (2, 0)})
    assert g.execute_by_component() == ([], {(0, 1), (1, 1), (2, 1)})


def test_ref_zigzag_compute_eligibility(zigzag):  # :189-194
    g = zigzag()
    g.commit((0, 0), 0, {(2, 0), (1, 0)})
    g.commit((2, 0), 0, {(0, 0)})
    assert g.execute_by_component() == ([], {(1, 0)})


# ---------------------------------------------------------------------------------------------------------------
# "All dep graph implementations should agree" (DependencyGraphTest.scala:331-462): same generator shape, with an
# independent answer -- strongly connected components by scipy, eligibility by reachability -- in the role of the
# library-backed JgraphtDependencyGraph / ScalaGraphDependencyGraph
# ---------------------------------------------------------------------------------------------------------------
def independent_components(nodes):
    """nodes: list of (key, seq, deps).  The set of eligible components, each sorted by (seq, key)."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components

    keys = [k for k, _, _ in nodes]
    committed = {}
    for k, s, d in nodes:
        committed.setdefault(k, (s, d))  # a repeated key is ignored
    every = sorted(set(keys) | {x for _, (_, d) in committed.items() for x in d})
    idx = {k: i for i, k in enumerate(every)}
    rows, cols = [], []
    for k, (_, d) in committed.items():
        for x in d:
            rows.append(idx[k])
            cols.append(idx[x])
    n = len(every)
    adj = csr_matrix((np.ones(len(rows), np.int8), (rows, cols)), shape=(n, n))
    # eligible(v): everything reachable from v is committed
    bad = {idx[k] for k in every if k not in committed}
    radj = adj.T.tocsr()
    seen, todo = set(bad), list(bad)
    while todo:
        u = todo.pop()
        for w in radj.indices[radj.indptr[u]:radj.indptr[u + 1]]:
            if w not in seen:
                seen.add(int(w))
                todo.append(int(w))
    _, label = connected_components(adj, directed=True, connection="strong")
    comps = {}
    for k in committed:
        if idx[k] not in seen:
            comps.setdefault(int(label[idx[k]]), []).append(k)
    return sorted(sorted(c, key=lambda k: (committed[k][0], k)) for c in comps.values())


def gen_nodes(rng, num_vertices, max_vertex):  # :340-352
    keys = rng.sample(range(max_vertex), num_vertices)
    out = []
    for k in keys:
        deps = set(rng.sample(range(max_vertex), rng.randint(0, num_vertices)))
        out.append((k, rng.randint(0, 1000), deps - {k}))
    return out


@pytest.mark.parametrize("num_vertices,max_vertex", [(5, 5), (10, 20), (10, 10), (100, 200), (100, 100), (50, 100)])
def test_all_implementations_agree(P, num_vertices, max_vertex):
    rng = random.Random(num_vertices * 1000 + max_vertex)
    for _ in range(40):
        nodes = gen_nodes(rng, num_vertices, max_vertex)
        want = independent_components(nodes)
        outs = []
        for impl in IMPLS:
            g = IntGraph(impl, P)
            for k, s, d in nodes:
                g.commit(k, s, d)
            comps, blockers = g.execute_by_component(None)
            assert sorted(comps) == want, impl  # "contain theSameElementsAs" :414-417
            # reverse topological: a dependency's component never comes later
            pos = {k: i for i, c in enumerate(comps) for k in c}
            deps = {}
            for k, _, d in nodes:
                deps.setdefault(k, d)
            for k in pos:
                assert all(pos[x] <= pos[k] for x in deps[k])
            outs.append((comps, blockers))
        assert outs[0] == outs[1]  # product == oracle, order included


# ---------------------------------------------------------------------------------------------------------------
# product == oracle on EPaxos-shaped input: (leader, id) keys, InstancePrefixSet dependencies (watermarks + explicit
# ids), commits interleaved with executes, updateExecuted, re-commits
# ---------------------------------------------------------------------------------------------------------------
def random_instance_deps(rng, n, horizon, explicit_p, i):
    """EPaxos-shaped: instance i of a leader saw roughly the first i +- spread instances of every other leader, so
    neighbours depend on each other (cycles) and everything stays inside the horizon (everything executes in the
    end)"""
    spread = rng.choice([0, 1, 2, 5])
    wm = [min(horizon, max(0, i + rng.randint(-spread, spread + 1))) if rng.random() < 0.8 else 0 for _ in range(n)]
    vals = []
    if rng.random() < explicit_p:
        for _ in range(rng.randint(1, 4)):
            vals.append((rng.randrange(n), rng.randrange(horizon)))
    return wm, vals


def oracle_deps(n, wm, vals, self_key=None):
    s = O.InstancePrefixSet(n, [O.IntPrefixSet(w, [x for (l2, x) in vals if l2 == l]) for l, w in enumerate(wm)])
    return s


COVER = {"executed": 0, "cycles": 0, "blockers": 0}


@pytest.mark.parametrize("kind", ["tarjan", "zigzag"])
@pytest.mark.parametrize("seed", range(8))
def test_product_equals_oracle_on_instance_graphs(P, kind, seed):
    rng = random.Random(seed * 7 + (kind == "zigzag"))
    n = rng.choice([3, 5, 7])
    horizon = rng.choice([6, 20, 60])
    if kind == "tarjan":
        og = O.TarjanDependencyGraph(O.InstancePrefixSet(n))
        pg = P.DependencyGraph(n, kind=P.FPX_DG_TARJAN)
    else:
        gc = rng.choice([3, 1000])
        og = O.ZigzagTarjanDependencyGraph(O.InstancePrefixSet(n), n, vertices_grow_size=4,
                                           garbage_collect_every_n_commands=gc)
        pg = P.DependencyGraph(n, kind=P.FPX_DG_ZIGZAG, gc_every_n=gc)
    pending = [(l, i) for l in range(n) for i in range(horizon)]
    rng.shuffle(pending)
    executed_total, committed = [], []
    while pending:
        batch = [pending.pop() for _ in range(min(len(pending), rng.randint(1, 12)))]
        if rng.random() < 0.15 and executed_total:  # a key committed again
            batch.append(rng.choice(executed_total))
        if rng.random() < 0.15 and len(batch) > 1:
            batch.append(batch[0])
        L, I, S, W, V = [], [], [], [], []
        for (l, i) in batch:
            wm, vals = random_instance_deps(rng, n, horizon, 0.3, i)
            vals = [v for v in vals if v != (l, i)]
            if wm[l] > i:  # a dependency set never contains its own instance (Replica.scala:582): K5's encoding
                vals += [(l, x) for x in range(i + 1, wm[l])]
                wm[l] = i
            seq = rng.randint(0, 3)
            og.commit((l, i), seq, oracle_deps(n, wm, vals))
            L.append(l), I.append(i), S.append(seq), W.append(wm), V.append(vals)
        pg.commit(L, I, S, W, V)
        committed += batch
        if rng.random() < 0.1:
            if kind == "tarjan":
                wm = [rng.randint(0, 3) for _ in range(n)]
                keys = [(rng.randrange(n), rng.randrange(horizon)) for _ in range(rng.randint(0, 3))]
            else:  # zigzag walks its columns through the VERTICES: only keys it holds (see the wedge test below)
                wm = [0] * n
                keys = rng.sample(committed, min(len(committed), rng.randint(0, 3)))
            og.update_executed(O.InstancePrefixSet(n, [O.IntPrefixSet(w, [x for (l2, x) in keys if l2 == l])
                                                       for l, w in enumerate(wm)]))
            pg.update_executed(wm, keys)
        if rng.random() < 0.6 or not pending:
            nb = rng.choice([None, None, 1, 3]) if kind == "tarjan" else None
            want = og.execute_by_component(nb)
            got = pg.execute_by_component(nb)
            assert got[0] == [list(c) for c in want[0]]
            assert got[1] == want[1]
            COVER["executed"] += sum(len(c) for c in got[0])
            COVER["cycles"] += sum(1 for c in got[0] if len(c) > 1)
            COVER["blockers"] += len(got[1])
            executed_total += [k for c in want[0] for k in c]
            assert list(pg.executed_watermark()) == [s.get_watermark() for s in og.executed.sets]
            if kind == "tarjan":
                assert pg.num_vertices == og.num_vertices
    # everything was committed (or declared executed): every instance is executed, and none twice
    assert sorted(executed_total) == sorted(set(executed_total))
    assert list(pg.executed_watermark()) == [horizon] * n


def test_zigzag_update_executed_of_a_key_it_never_held_wedges_the_column(P):
    """executeKeyImpl asks the vertex column before the executed set (ZigzagTarjanDependencyGraph.scala:510-519) and
    commit ignores an executed key (:350): a key declared executed that was never committed stops its column for
    good.  The reference's behaviour, kept (and the reason the random test above only declares held keys)."""
    og = O.ZigzagTarjanDependencyGraph(O.InstancePrefixSet(2), 2)
    pg = P.DependencyGraph(2, kind=P.FPX_DG_ZIGZAG)
    og.update_executed(O.InstancePrefixSet(2, [O.IntPrefixSet(0, [0]), O.IntPrefixSet()]))
    pg.update_executed(None, [(0, 0)])
    for key in [(0, 0), (0, 1), (1, 0)]:
        og.commit(key, 0, O.InstancePrefixSet(2))
        pg.commit([key[0]], [key[1]], None, [[0, 0]])
    want = og.execute_by_component()
    assert want == ([[(1, 0)]], {(0, 0), (1, 1)})
    assert pg.execute_by_component() == want


def test_instance_graphs_reached_the_interesting_cases():
    assert COVER["executed"] > 1000 and COVER["cycles"] > 10 and COVER["blockers"] > 100, COVER


def test_commit_epx_is_commit_with_the_own_column_run(P):
    """fpx_depgraph_commit_epx == fpx_depgraph_commit with the explicit ids id+1 .. end-1 of the own column"""
    rng = random.Random(5)
    n = 5
    a = P.DependencyGraph(n, kind=P.FPX_DG_ZIGZAG)
    b = P.DependencyGraph(n, kind=P.FPX_DG_ZIGZAG)
    keys = [(l, i) for l in range(n) for i in range(30)]
    rng.shuffle(keys)
    L, I, D, E, V, M = [], [], [], [], [], []
    for (l, i) in keys:
        wm = [rng.randint(0, 30) for _ in range(n)]
        end = 0
        if wm[l] > i:
            end = wm[l] if wm[l] > i + 1 else 0
            wm[l] = i
        L.append(l), I.append(i), D.append(wm), E.append([end, 77]), M.append(rng.random() < 0.8)
        V.append([(l, x) for x in range(i + 1, end)])
    a.commit_epx(L, I, D, E, mask=M)
    sel = [k for k in range(len(keys)) if M[k]]
    b.commit([L[k] for k in sel], [I[k] for k in sel], None, [D[k] for k in sel], [V[k] for k in sel])
    assert a.execute_by_component() == b.execute_by_component()


def test_long_chain_is_a_loop_not_a_recursion(P):
    """one hot key: instance i depends on everything before it; committed newest first so that a single
    strongConnect walks the whole chain"""
    m = 300_000
    g = P.DependencyGraph(1, kind=P.FPX_DG_ZIGZAG)
    ids = np.arange(m - 1, -1, -1, dtype=np.int32)
    g.commit(np.zeros(m, np.int32), ids, None, ids.reshape(m, 1))  # watermark = own id: every earlier instance
    el, ei, cs, bl, bi = g.execute_arrays()
    assert np.array_equal(ei, np.arange(m)) and np.all(cs == 1) and list(zip(bl, bi)) == [(0, m)]


def test_bad_arguments_are_einval(P):
    import frankenpaxos_amd as fa

    g = P.DependencyGraph(3)
    for args in [([3], [0], [0], [[0, 0, 0]]), ([0], [-1], [0], [[0, 0, 0]]), ([0], [0], [0], [[0, -1, 0]])]:
        with pytest.raises(fa.FpxError) as e:
            g.commit(*args)
        assert e.value.status == fa.FPX_EINVAL
    with pytest.raises(fa.FpxError):
        g.commit([0], [0], [0], [[0, 0, 0]], [[(5, 1)]])
    with pytest.raises(fa.FpxError):
        P.DependencyGraph(0)
    assert g.execute_by_component() == ([], {(0, 0), (1, 0), (2, 0)})  # nothing was committed


def test_a_key_far_ahead_of_the_executed_watermark_is_refused():
    """the vertex columns and the executed sets are dense from the watermark on; one key near 2^31 would
    allocate gigabytes.  FPX_ECAPACITY, nothing applied; after the watermark has moved the same key is welcome"""
    import frankenpaxos_amd as fa
    from frankenpaxos_amd import depgraph as P

    for kind in (P.FPX_DG_TARJAN, P.FPX_DG_ZIGZAG):
        g = P.DependencyGraph(2, kind=kind)
        far = (1 << 26) + 5
        with pytest.raises(fa.FpxError) as e:
            g.commit_epx([1], [far], [[0, 0]])
        assert e.value.status == fa.FPX_ECAPACITY
        with pytest.raises(fa.FpxError):
            g.update_executed(None, [(0, 2_000_000_000)])
        assert g.num_vertices == 0
        g.commit_epx([0, 1], [0, 1 << 20], [[0, 0], [0, 0]])        # a megabyte away is fine
        assert g.num_vertices == 2
        g.update_executed([0, far - 10])
        g.commit_epx([1], [far], [[0, 0]])
        assert g.num_vertices >= 2
