"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's dependency-graph execution.

Only tests/, __graft_entry__.smoke() and bench's cpu_baseline leg may import this; the product
(frankenpaxos_amd/csrc/fpx_depgraph.cpp behind include/fpx_depgraph.h) never does.

What is restated, with the reference's own shapes (objects, maps, explicit sets, RECURSIVE strongConnect), paths
relative to /root/reference/shared/src/main/scala/frankenpaxos/:

  compact/IntPrefixSet.scala            IntPrefixSet        (watermark + explicit values; add :185, contains :198,
                                                             addAll :264-301, materializedDiff :233-259,
                                                             diffIterator :38-50 with WatermarkIterator :105-145 and
                                                             ValuesIterator :83-103, compact :386-391)
  epaxos/InstancePrefixSet.scala        InstancePrefixSet   (one IntPrefixSet per leader; diffIterator :118-126,
                                                             materializedDiff :105-116, leaderIndexWatermark :181)
  compact/FakeCompactSet.scala          FakeCompactSet      (a plain set; what ZigzagTarjanDependencyGraphTest uses)
  util/BufferMap.scala                  BufferMap           (get :29, put :37-51, garbageCollect :55-63)
  depgraph/TarjanDependencyGraph.scala  TarjanDependencyGraph  (commit :225-239, updateExecuted :241-244,
                                                             execute :257-276, appendExecute :278-296,
                                                             executeByComponent :298-321, executeImpl :323-356,
                                                             strongConnect :358-462)
  depgraph/ZigzagTarjanDependencyGraph.scala  ZigzagTarjanDependencyGraph (commit :341-360, executeImpl :465-500,
                                                             executeKeyImpl :502-566, strongConnect :568-720)
  epaxos/Replica.scala:859-917          how EPaxos drives it: commit(instance, sequenceNumber, dependencies) for every
                                                             committed triple, then appendExecute.

Parity status: PINNED.  tests/test_depgraph.py runs every known-answer test of the reference's own
shared/src/test/scala/depgraph/DependencyGraphTest.scala and ZigzagTarjanDependencyGraphTest.scala (transcribed)
against this restatement and, through the C ABI, against the product.

One thing is NOT the reference's, and it is unobservable in its tests: wherever the reference iterates a HASH
collection (the roots `for ((key, vertex) <- vertices)` of TarjanDependencyGraph :329 -- a mutable.HashMap; the
explicit `values` of an IntPrefixSet / a FakeCompactSet -- mutable.HashSet), the JVM's bucket order decides between
orders the reference itself treats as equally right (DependencyGraphTest.scala:188-191, 226-230, 274-281 accept sets
of answers).  This restatement and the product both take ASCENDING key order there.  ZigzagTarjanDependencyGraph --
what epaxos/ReplicaMain.scala:127 deploys -- has no hash-ordered roots at all (column round-robin from the executed
watermarks), so with watermark-only dependencies (K5-K7's top-one dependencies on FIFO channels) nothing is left
to choose.
"""
import sys


# ---------------------------------------------------------------------------------------------------------------
# compact sets
# ---------------------------------------------------------------------------------------------------------------
class IntPrefixSet:
    """compact/IntPrefixSet.scala: {0 .. watermark-1} U values"""

    def __init__(self, watermark=0, values=()):
        self.watermark = watermark
        self.values = set(values)
        self._compact()

    @classmethod
    def of(cls, values):  # IntPrefixSet(Set(...)) :18-19
        return cls(0, values)

    def _compact(self):  # :386-391
        while self.watermark in self.values:
            self.values.remove(self.watermark)
            self.watermark += 1

    def clone(self):
        return IntPrefixSet(self.watermark, self.values)

    def add(self, x):  # :185-196
        assert x >= 0
        if x < self.watermark:
            return False
        fresh = x not in self.values
        self.values.add(x)
        self._compact()
        return fresh

    def contains(self, x):  # :198-201
        assert x >= 0
        return x < self.watermark or x in self.values

    def add_all(self, other):  # :264-301 (the four cases collapse to this; _compact() where the reference compacts)
        if other.watermark > self.watermark:
            self.watermark = other.watermark
        self.values = {v for v in (self.values | other.values) if v >= self.watermark}
        self._compact()
        return self

    def get_watermark(self):
        return self.watermark

    def materialize(self):  # :371
        return set(self.values) | set(range(self.watermark))

    def materialized_diff(self, other):  # :233-259 (a lazy view there; nothing changes under it in Tarjan)
        out = [x for x in range(other.watermark, self.watermark) if x not in other.values]
        out += [x for x in sorted(self.values) if x >= other.watermark and x not in other.values]
        return out

    def diff_iterator(self, other):
        """DiffIterator :38-50 -- LAZY against a LIVE `other`: every element is looked for when it is asked for.
        WatermarkIterator.getNext :124-145, then ValuesIterator.getNext :85-101 (hash order there, ascending here)."""
        x = 0
        to = self.watermark
        while True:
            if x >= to or to <= other.watermark:
                break
            start = max(x, other.watermark)
            if other.values:
                while start in other.values:
                    start += 1
                    if start >= to:
                        start = None
                        break
                if start is None:
                    break
            x = start + 1
            yield start
        for v in sorted(self.values):
            if v < other.watermark or v in other.values:
                continue
            yield v


class InstancePrefixSet:
    """epaxos/InstancePrefixSet.scala; keys are (replicaIndex, instanceNumber) tuples"""

    def __init__(self, num_replicas, sets=None):
        self.n = num_replicas
        self.sets = sets if sets is not None else [IntPrefixSet() for _ in range(num_replicas)]

    @classmethod
    def from_watermarks(cls, watermarks):  # :19-25
        return cls(len(watermarks), [IntPrefixSet(w) for w in watermarks])

    @classmethod
    def from_epx(cls, leader, number, deps, values_end):
        """K5-K7's encoding (include/fpx.h): n watermarks + the own-leader column's explicit values
        number+1 .. values_end-1 (0 = none)"""
        s = cls.from_watermarks([int(w) for w in deps])
        if values_end:
            s.sets[leader] = IntPrefixSet(int(deps[leader]), range(number + 1, values_end))
        return s

    def add(self, key):
        return self.sets[key[0]].add(key[1])

    def contains(self, key):
        return self.sets[key[0]].contains(key[1])

    def add_all(self, other):
        for a, b in zip(self.sets, other.sets):
            a.add_all(b)
        return self

    def materialized_diff(self, other):  # :105-116
        return [(i, x) for i in range(self.n) for x in self.sets[i].materialized_diff(other.sets[i])]

    def diff_iterator(self, other):  # :118-126
        for i in range(self.n):
            for x in self.sets[i].diff_iterator(other.sets[i]):
                yield (i, x)

    def leader_index_watermark(self, leader_index):  # :181
        return self.sets[leader_index].get_watermark()

    def materialize(self):
        return {(i, x) for i in range(self.n) for x in self.sets[i].materialize()}


class IntSetAsKeys(IntPrefixSet):
    """IntPrefixSet with plain Int keys (DependencyGraphTest's KeySet)"""

    def leader_index_watermark(self, leader_index):
        raise NotImplementedError  # CompactSet.scala:55: ???


class FakeCompactSet:
    """compact/FakeCompactSet.scala"""

    def __init__(self, values=()):
        self.values = set(values)

    def add(self, key):
        fresh = key not in self.values
        self.values.add(key)
        return fresh

    def contains(self, key):
        return key in self.values

    def add_all(self, other):
        self.values |= other.values
        return self

    def materialized_diff(self, other):  # :28-29
        return [v for v in sorted(self.values) if v not in other.values]

    def diff_iterator(self, other):  # :30-31, lazy filter against the live other
        for v in sorted(self.values):
            if v not in other.values:
                yield v

    def leader_index_watermark(self, leader_index):  # :45
        return 0

    def materialize(self):
        return set(self.values)


class BufferMap:
    """util/BufferMap.scala"""

    def __init__(self, grow_size=5000):
        self.grow_size = grow_size
        self.buffer = [None] * grow_size
        self.watermark = 0
        self.largest_key = -1

    def get(self, key):  # :29-35
        k = key - self.watermark
        if k < 0 or k >= len(self.buffer):
            return None
        return self.buffer[k]

    def put(self, key, value):  # :37-51
        self.largest_key = max(self.largest_key, key)
        k = key - self.watermark
        if k < 0:
            return
        if k < len(self.buffer):
            self.buffer[k] = value
            return
        self.buffer += [None] * (k + 1 + self.grow_size - len(self.buffer))
        self.buffer[k] = value

    def garbage_collect(self, watermark):  # :55-63
        if watermark <= self.watermark:
            return
        del self.buffer[: min(watermark - self.watermark, len(self.buffer))]
        self.watermark = watermark


# ---------------------------------------------------------------------------------------------------------------
# graphs
# ---------------------------------------------------------------------------------------------------------------
class _Vertex:
    __slots__ = ("key", "sequence_number", "dependencies")

    def __init__(self, key, sequence_number, dependencies):
        self.key, self.sequence_number, self.dependencies = key, sequence_number, dependencies


class _Meta:
    __slots__ = ("number", "low_link", "stack_index", "eligible")

    def __init__(self, number, low_link, stack_index, eligible):
        self.number, self.low_link, self.stack_index, self.eligible = number, low_link, stack_index, eligible


class TarjanDependencyGraph:
    """depgraph/TarjanDependencyGraph.scala"""

    def __init__(self, empty_key_set):
        self.vertices = {}  # :205
        self.executed = empty_key_set  # :208
        self.metadatas, self.stack = {}, []

    def commit(self, key, sequence_number, dependencies):  # :225-239
        if key in self.vertices or self.executed.contains(key):
            return
        self.vertices[key] = _Vertex(key, sequence_number, dependencies)

    def update_executed(self, keys):  # :241-244
        self.executed.add_all(keys)
        self.vertices = {k: v for k, v in self.vertices.items() if not self.executed.contains(k)}

    @property
    def num_vertices(self):  # :464
        return len(self.vertices)

    def execute_by_component(self, num_blockers=None):  # :298-321
        self.metadatas, self.stack = {}, []
        executables, blockers = [], set()
        self._execute_impl(num_blockers, executables, blockers)
        for component in executables:
            for key in component:
                del self.vertices[key]
                self.executed.add(key)
        return executables, blockers

    def execute(self, num_blockers=None):  # :257-276
        components, blockers = self.execute_by_component(num_blockers)
        return [k for c in components for k in c], blockers

    def append_execute(self, num_blockers, executables, blockers):  # :278-296
        e, b = self.execute(num_blockers)
        executables.extend(e)
        blockers |= b

    def _execute_impl(self, num_blockers, executables, blockers):  # :323-356
        for key in sorted(self.vertices):  # the reference: hash order (module docstring)
            if key not in self.metadatas:
                self._strong_connect(key, executables, blockers)
                if not self.metadatas[key].eligible:
                    self.stack.clear()
                if num_blockers is not None and len(blockers) >= num_blockers:
                    return

    def _strong_connect(self, v, executables, blockers):  # :358-462
        metadatas, stack, vertices = self.metadatas, self.stack, self.vertices
        number = len(metadatas)
        mv = metadatas[v] = _Meta(number, number, len(stack), True)
        stack.append(v)
        for w in vertices[v].dependencies.materialized_diff(self.executed):
            if w not in vertices:  # :380-389 uncommitted child
                mv.eligible = False
                blockers.add(w)
                return
            elif w not in metadatas:  # :390-405 unexplored child
                self._strong_connect(w, executables, blockers)
                mw = metadatas[w]
                if not mw.eligible:
                    mv.eligible = False
                    return
                mv.low_link = min(mv.low_link, mw.low_link)
            elif not metadatas[w].eligible:  # :406-413
                mv.eligible = False
                return
            elif metadatas[w].stack_index != -1:  # :414-418 on stack
                mv.low_link = min(mv.low_link, metadatas[w].number)
            # else off stack :419-422
        if mv.low_link != mv.number:  # :427-429
            return
        if mv.stack_index == len(stack) - 1:  # :434-439
            component = stack.pop()
            metadatas[component].stack_index = -1
            executables.append([component])
        else:  # :440-451
            component = stack[mv.stack_index:]
            del stack[mv.stack_index:]
            for w in component:
                metadatas[w].stack_index = -1
            executables.append(sorted(component, key=lambda k: (vertices[k].sequence_number, k)))


class ZigzagTarjanDependencyGraph:
    """depgraph/ZigzagTarjanDependencyGraph.scala.  Keys are (leaderIndex, id) tuples (VertexIdLike)."""

    def __init__(self, empty_key_set, num_leaders, vertices_grow_size=1000, garbage_collect_every_n_commands=1000):
        self.num_leaders = num_leaders
        self.gc_every = garbage_collect_every_n_commands
        self.vertices = [BufferMap(vertices_grow_size) for _ in range(num_leaders)]  # :289-291
        self.executed_watermark = [0] * num_leaders  # :296
        self.num_commands_since_last_gc = 0
        self.executed = empty_key_set  # :304
        self.metadatas, self.stack = {}, []

    def _get_vertex(self, key):  # :322-323
        return self.vertices[key[0]].get(key[1])

    def _garbage_collect(self):  # :333-337
        for i in range(self.num_leaders):
            self.vertices[i].garbage_collect(self.executed_watermark[i])

    def commit(self, key, sequence_number, dependencies):  # :341-360
        # `vertices.contains(key)` :350 asks a Buffer[BufferMap] for a Key: always false.  A committed, not yet
        # executed key committed again REPLACES its vertex.
        if self.executed.contains(key):
            return
        self.vertices[key[0]].put(key[1], _Vertex(key, sequence_number, dependencies))

    def update_executed(self, keys):  # :362
        self.executed.add_all(keys)

    def execute_by_component(self, num_blockers=None):  # :418-445 (numBlockers is ignored there too)
        self.metadatas, self.stack = {}, []
        executables, blockers = [], set()
        self._execute_impl(executables, blockers)
        self.num_commands_since_last_gc += sum(len(c) for c in executables)
        if self.num_commands_since_last_gc >= self.gc_every:
            self._garbage_collect()
            self.num_commands_since_last_gc = 0
        return executables, blockers

    def execute(self, num_blockers=None):  # :364-388
        components, blockers = self.execute_by_component(num_blockers)
        return [k for c in components for k in c], blockers

    def append_execute(self, num_blockers, executables, blockers):  # :390-416
        e, b = self.execute(num_blockers)
        executables.extend(e)
        blockers |= b

    def _execute_impl(self, executables, blockers):  # :465-500
        eligible_columns = list(range(self.num_leaders))
        index = 0
        while eligible_columns:
            leader_index = eligible_columns[index]
            id_ = self.executed_watermark[leader_index]
            if self._execute_key_impl(executables, blockers, leader_index, id_):
                self.executed_watermark[leader_index] = max(self.executed_watermark[leader_index] + 1,
                                                            self.executed.leader_index_watermark(leader_index))
                index += 1
                if index >= len(eligible_columns):
                    index = 0
            else:
                del eligible_columns[index]
                if index >= len(eligible_columns):
                    index = 0

    def _execute_key_impl(self, executables, blockers, leader_index, id_):  # :502-566
        v = (leader_index, id_)
        vertex = self.vertices[leader_index].get(id_)
        if vertex is None:
            blockers.add(v)
            return False
        if self.executed.contains(v):
            return True
        mv = self.metadatas.get(v)
        if mv is None:
            mv = self._strong_connect(v, vertex, executables, blockers)
            if not mv.eligible:
                for u in self.stack:
                    self.metadatas[u].eligible = False
                    self.metadatas[u].stack_index = -1
                self.stack.clear()
                return False
            return True
        return mv.eligible

    def _strong_connect(self, v, vertex, executables, blockers):  # :568-720
        metadatas, stack = self.metadatas, self.stack
        number = len(metadatas)
        iterator = vertex.dependencies.diff_iterator(self.executed)
        nxt = next(iterator, None)
        if nxt is None:  # :583-599 nothing left to wait for
            mv = metadatas[v] = _Meta(number, number, -1, True)
            executables.append([v])
            self.executed.add(v)
            return mv
        mv = metadatas[v] = _Meta(number, number, len(stack), True)
        stack.append(v)
        while nxt is not None:
            w = nxt
            wertex = self._get_vertex(w)
            if wertex is None:  # :619-629
                mv.eligible = False
                mv.stack_index = -1
                blockers.add(w)
                return mv
            mw = metadatas.get(w)
            if mw is None:  # :633-650
                mw = self._strong_connect(w, wertex, executables, blockers)
                if not mw.eligible:
                    mv.eligible = False
                    mv.stack_index = -1
                    return mv
                mv.low_link = min(mv.low_link, mw.low_link)
            elif not mw.eligible:  # :653-663
                mv.eligible = False
                mv.stack_index = -1
                return mv
            elif mw.stack_index != -1:  # :664-671
                mv.low_link = min(mv.low_link, mw.number)
            nxt = next(iterator, None)  # hasNext at the loop head: looked for NOW, against the live executed set
        if mv.low_link != mv.number:  # :686-688
            return mv
        if mv.stack_index == len(stack) - 1:  # :694-699
            stack.pop()
            mv.stack_index = -1
            executables.append([v])
            self.executed.add(v)
        else:  # :700-716
            component = stack[mv.stack_index:]
            del stack[mv.stack_index:]
            for w in component:
                metadatas[w].stack_index = -1
                self.executed.add(w)
            executables.append(sorted(component, key=lambda k: (self._get_vertex(k).sequence_number, k)))
        return mv


def with_deep_stack(fn, *args, stack_mb=1024, depth=10_000_000):
    """run fn(*args) on a thread with a big C stack: strongConnect recurses once per vertex of a dependency chain,
    as the reference does"""
    import threading

    out = {}

    def run():
        sys.setrecursionlimit(depth)
        try:
            out["v"] = fn(*args)
        except BaseException as e:  # noqa: BLE001 -- handed to the caller's thread
            out["e"] = e

    old = threading.stack_size(stack_mb << 20)
    try:
        t = threading.Thread(target=run)
        t.start()
        t.join()
    finally:
        threading.stack_size(old)
    if "e" in out:
        raise out["e"]
    return out["v"]
