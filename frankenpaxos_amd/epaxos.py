"""EPaxos pre-accept fast path (K5) on the GPU: python handle on an fpx_epx context.

Every result is computed by the HIP kernels of csrc/fpx_epaxos.hip; there is no CPU path here."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FpxError


class FpxEpxConfig(C.Structure):
    _fields_ = [("num_replicas", C.c_int32), ("num_keys", C.c_int32), ("device", C.c_int32),
                ("flags", C.c_uint32), ("num_instances", C.c_int32)]


def _bind(L):
    if getattr(L, "_epx_bound", False):
        return
    VP = C.c_void_p
    L.fpx_epx_create.argtypes = [C.POINTER(FpxEpxConfig), C.POINTER(VP)]
    L.fpx_epx_destroy.argtypes = [VP]
    L.fpx_epx_set_stream.argtypes = [VP, VP]
    L.fpx_epx_sync.argtypes = [VP]
    # fpx_epx_preaccept[_dev]: the table in _lib.SIGNATURES (one place, checked against the header)
    L.fpx_epx_read_index.argtypes = [VP, C.c_int32, C.c_int32, VP, VP]
    L._epx_bound = True


class EPaxos:
    def __init__(self, num_replicas, num_keys, device=0, num_instances=0):
        """num_instances > 0: every replica keeps its command log for instances (leader, number < num_instances):
        pre-accept records it, prepare() / accept() run the per-instance Paxos on it"""
        self.L = _lib.lib()
        _bind(self.L)
        self.n, self.num_keys = num_replicas, num_keys
        cfg = FpxEpxConfig(num_replicas, num_keys, device, 0, num_instances)
        h = C.c_void_p()
        st = self.L.fpx_epx_create(C.byref(cfg), C.byref(h))
        if st:
            raise FpxError(st, "fpx_epx_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.L.fpx_epx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, s):
        st = self.L.fpx_epx_set_stream(self._h, C.c_void_p(-1 if s is None else int(s)))
        if st:
            raise FpxError(st, "fpx_epx_set_stream")

    def sync(self):
        return self.L.fpx_epx_sync(self._h)

    def preaccept(self, leader, number, key, is_set, resp_mask, rank, seen_mask=None, triple_id=None):
        """seen_mask: the other replicas that process the PreAccept (None = resp_mask, a thrifty
        deployment; all n-1 others with the reference's default ThriftySystem.NotThrifty)"""
        a32 = lambda x: np.ascontiguousarray(x, dtype=np.int32)
        a8 = lambda x: np.ascontiguousarray(x, dtype=np.uint8)
        leader, number, key, rank = a32(leader), a32(number), a32(key), a32(rank)
        is_set, resp_mask = a8(is_set), a8(resp_mask)
        seen_mask = None if seen_mask is None else a8(seen_mask)
        triple_id = None if triple_id is None else a32(triple_id)
        m = len(leader)
        fast = np.zeros(m, np.uint8)
        deps = np.zeros((m, self.n), np.int32)
        ldeps = np.zeros((m, self.n), np.int32)
        own = np.zeros((m, 2), np.int32)
        p = lambda a: None if a is None else a.ctypes.data
        st = self.L.fpx_epx_preaccept(self._h, m, p(leader), p(number), p(key), p(is_set), p(resp_mask),
                                      p(seen_mask), p(rank), p(triple_id), p(fast), p(deps), p(ldeps), p(own))
        return st, fast, deps, ldeps, own

    def prepare(self, leader, number, ballot_ordering, ballot_replica, target_mask):
        """Replica.handlePrepare at the replicas of target_mask: (status, ok_bits, nack_bits, commit_bits,
        nack_ballot, reply_status[m, n], reply_vote[m, n], reply_triple[m, n])"""
        a32 = lambda x: np.ascontiguousarray(x, dtype=np.int32)
        leader, number, bo, br = a32(leader), a32(number), a32(ballot_ordering), a32(ballot_replica)
        tgt = np.ascontiguousarray(target_mask, dtype=np.uint8)
        m = len(leader)
        ok, nack, com = (np.zeros(m, np.uint8) for _ in range(3))
        nb = np.full(m, -1, np.int32)
        rs, rv, rt = (np.full((m, self.n), -1, np.int32) for _ in range(3))
        p = lambda a: a.ctypes.data
        st = self.L.fpx_epx_prepare(self._h, m, p(leader), p(number), p(bo), p(br), p(tgt), p(ok), p(nack), p(com),
                                    p(nb), p(rs), p(rv), p(rt))
        return st, ok, nack, com, nb, rs, rv, rt

    def handle_prepare_oks(self, leader, number, ballot_ordering, ballot_replica, resp_mask, reply_status, reply_vote,
                           reply_triple, as_intended=False):
        """Replica.handlePrepareOk, the recovering replica's decision: (status, action, source, triple) -- action 0 wait,
        1 Accept phase with `source`'s triple, 2 pre-accept again with its command, 3 pre-accept a Noop"""
        a32 = lambda x: np.ascontiguousarray(x, dtype=np.int32)
        leader, number, bo, br = a32(leader), a32(number), a32(ballot_ordering), a32(ballot_replica)
        mask = np.ascontiguousarray(resp_mask, dtype=np.uint8)
        rs, rv, rt = a32(reply_status), a32(reply_vote), a32(reply_triple)
        m = len(leader)
        act, src, tr = (np.full(m, -9, np.int32) for _ in range(3))
        p = lambda a: a.ctypes.data
        st = self.L.fpx_epx_handle_prepare_oks(self._h, m, p(leader), p(number), p(bo), p(br), p(mask), p(rs), p(rv), p(rt),
                                               int(as_intended), p(act), p(src), p(tr))
        return st, act, src, tr

    def accept(self, leader, number, ballot_ordering, ballot_replica, triple_id, target_mask, key=None, is_set=None):
        """the Accept phase: (status, ok_bits, nack_bits, commit_bits, nack_ballot, committed).  key / is_set: the
        triples' commands (updateConflictIndex wherever the triple is stored); key None = every triple is a Noop"""
        a32 = lambda x: np.ascontiguousarray(x, dtype=np.int32)
        leader, number, bo, br, tr = a32(leader), a32(number), a32(ballot_ordering), a32(ballot_replica), a32(triple_id)
        tgt = np.ascontiguousarray(target_mask, dtype=np.uint8)
        m = len(leader)
        key = np.full(m, -1, np.int32) if key is None else a32(key)
        is_set = np.zeros(m, np.uint8) if is_set is None else np.ascontiguousarray(is_set, dtype=np.uint8)
        ok, nack, com, done = (np.zeros(m, np.uint8) for _ in range(4))
        nb = np.full(m, -1, np.int32)
        p = lambda a: a.ctypes.data
        st = self.L.fpx_epx_accept(self._h, m, p(leader), p(number), p(bo), p(br), p(tr), p(key), p(is_set), p(tgt),
                                   p(ok), p(nack), p(com), p(nb), p(done))
        return st, ok, nack, com, nb, done

    def handle_commit(self, leader, number, triple_id, target_mask, key=None, is_set=None, deps=None, deps_values_end=None):
        """Replica.handleCommit at the replicas of target_mask: CommittedEntry(triple) whatever was there, the conflict
        index learns the command.  key None = Noops; deps None = the triple is known by its id alone.  -> status"""
        a32 = lambda x: None if x is None else np.ascontiguousarray(x, dtype=np.int32)
        leader, number, tr = a32(leader), a32(number), a32(triple_id)
        m = len(leader)
        tgt = np.ascontiguousarray(target_mask, dtype=np.uint8)
        key = np.full(m, -1, np.int32) if key is None else a32(key)
        is_set = np.zeros(m, np.uint8) if is_set is None else np.ascontiguousarray(is_set, dtype=np.uint8)
        d, de = a32(deps), a32(deps_values_end)
        p = lambda a: None if a is None else a.ctypes.data
        return self.L.fpx_epx_handle_commit(self._h, m, p(leader), p(number), p(tr), p(key), p(is_set), p(d), p(de), p(tgt))

    def handle_preaccept(self, leader, number, ballot_ordering, ballot_replica, key, is_set, triple_id, deps_in,
                         deps_in_values_end, target_mask):
        """Replica.handlePreAccept in full at the replicas of target_mask (key -1 = Noop): (status, ok_bits,
        resend_bits, nack_bits, commit_bits, nack_ballot, reply_deps[m, n, n], reply_values_end[m, n],
        reply_triple[m, n])"""
        a32 = lambda x: None if x is None else np.ascontiguousarray(x, dtype=np.int32)
        leader, number, bo, br, key = a32(leader), a32(number), a32(ballot_ordering), a32(ballot_replica), a32(key)
        is_set = np.ascontiguousarray(is_set, dtype=np.uint8)
        tr, din, dend = a32(triple_id), a32(deps_in), a32(deps_in_values_end)
        tgt = np.ascontiguousarray(target_mask, dtype=np.uint8)
        m = len(leader)
        ok, resend, nack, com = (np.zeros(m, np.uint8) for _ in range(4))
        nb = np.full(m, -1, np.int32)
        rd = np.zeros((m, self.n, self.n), np.int32)
        re = np.zeros((m, self.n), np.int32)
        rt = np.full((m, self.n), -1, np.int32)
        p = lambda a: None if a is None else a.ctypes.data
        st = self.L.fpx_epx_handle_preaccept(self._h, m, p(leader), p(number), p(bo), p(br), p(key), p(is_set), p(tr),
                                             p(din), p(dend), p(tgt), p(ok), p(resend), p(nack), p(com), p(nb), p(rd),
                                             p(re), p(rt))
        return st, ok, resend, nack, com, nb, rd, re, rt

    def read_cmdlog_deps(self, replica, leader, number):
        """the dependencies kept with a command-log entry: (watermarks[n], values_end)"""
        deps, end = np.zeros(self.n, np.int32), np.zeros(1, np.int32)
        st = self.L.fpx_epx_read_cmdlog_deps(self._h, replica, leader, number, deps.ctypes.data, end.ctypes.data)
        if st:
            raise FpxError(st, "fpx_epx_read_cmdlog_deps")
        return deps, int(end[0])

    def read_cmdlog(self, replica, leader, number):
        """(kind, ballot, voteBallot, triple id, the replica's largestBallot); ballots encoded ordering * 8 + index"""
        out = np.zeros(5, np.int32)
        st = self.L.fpx_epx_read_cmdlog(self._h, replica, leader, number, out.ctypes.data)
        if st:
            raise FpxError(st, "fpx_epx_read_cmdlog")
        return tuple(int(x) for x in out)

    def preaccept_dev(self, leader, number, key, is_set, resp_mask, rank, fast=None, deps=None,
                      leader_deps=None, seen_mask=None, own_values_end=None, triple_id=None):
        d = lambda t: None if t is None else t.data_ptr()
        st = self.L.fpx_epx_preaccept_dev(self._h, leader.numel(), d(leader), d(number), d(key), d(is_set),
                                          d(resp_mask), d(seen_mask), d(rank), d(triple_id), d(fast), d(deps),
                                          d(leader_deps), d(own_values_end))
        if st:
            raise FpxError(st, "fpx_epx_preaccept_dev")

    def packed_stride(self):
        return int(self.L.fpx_epx_packed_stride(self.n))

    def preaccept_packed_dev(self, leader, number, key, is_set, resp_mask, rank, packed, seen_mask=None, triple_id=None):
        """one line of packed_stride() ints per command: deps | leader_deps | own_values_end[2] | fast (see unpack)"""
        d = lambda t: None if t is None else t.data_ptr()
        st = self.L.fpx_epx_preaccept_packed_dev(self._h, leader.numel(), d(leader), d(number), d(key), d(is_set),
                                                 d(resp_mask), d(seen_mask), d(rank), d(triple_id), d(packed))
        if st:
            raise FpxError(st, "fpx_epx_preaccept_packed_dev")

    def execute_dev(self, leader, number, packed, first, count, order, component, committed=None):
        """fpx_epx_execute_dev: the tick's commits through the device dependency graph.  Returns (num_executed,
        num_components, needs_host_path); order / component are filled on the device"""
        d = lambda t: None if t is None else t.data_ptr()
        f = np.ascontiguousarray(first, np.int32)
        c = np.ascontiguousarray(count, np.int32)
        ne, nc, nh = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        st = self.L.fpx_epx_execute_dev(self._h, leader.numel(), d(leader), d(number), d(packed), d(committed), f.ctypes.data,
                                        c.ctypes.data, d(order), d(component), C.addressof(ne), C.addressof(nc), C.addressof(nh))
        if st:
            raise FpxError(st, "fpx_epx_execute_dev")
        return ne.value, nc.value, nh.value

    def execute(self, leader, number, deps, first, count, deps_values_end=None, committed=None):
        """fpx_epx_execute: the same on host (numpy) arrays.  Returns (order, component, num_components, needs_host_path);
        order / component hold the num_executed entries"""
        leader, number = np.ascontiguousarray(leader, np.int32), np.ascontiguousarray(number, np.int32)
        deps = np.ascontiguousarray(deps, np.int32)
        m = len(leader)
        ends = None if deps_values_end is None else np.ascontiguousarray(deps_values_end, np.int32)
        cm = None if committed is None else np.ascontiguousarray(committed, np.uint8)
        f, c = np.ascontiguousarray(first, np.int32), np.ascontiguousarray(count, np.int32)
        order, comp = np.full(m, -1, np.int32), np.full(m, -1, np.int32)
        ne, nc, nh = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        p = lambda a: None if a is None else a.ctypes.data
        st = self.L.fpx_epx_execute(self._h, m, p(leader), p(number), p(deps), p(ends), p(cm), p(f), p(c), p(order), p(comp),
                                    C.addressof(ne), C.addressof(nc), C.addressof(nh))
        if st:
            raise FpxError(st, "fpx_epx_execute")
        return order[:ne.value], comp[:ne.value], nc.value, nh.value

    def unpack(self, packed):
        """packed [m, stride] (numpy or torch) -> fast [m], deps [m, n], leader_deps [m, n], own_values_end [m, 2]"""
        n = self.n
        return packed[:, 2 * n + 2], packed[:, :n], packed[:, n:2 * n], packed[:, 2 * n:2 * n + 2]

    def read_index(self, replica, key):
        g = np.zeros(self.n, np.int32)
        s = np.zeros(self.n, np.int32)
        st = self.L.fpx_epx_read_index(self._h, replica, key, g.ctypes.data, s.ctypes.data)
        if st:
            raise FpxError(st, "fpx_epx_read_index")
        return g, s
