"""Dependency-graph execution (include/fpx_depgraph.h): python handle on an fpx_depgraph.

Host code of libfpx.so (csrc/fpx_depgraph.cpp) -- the step after an EPaxos commit
(epaxos/Replica.scala:859-917).  Nothing is computed in python."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FpxError

FPX_DG_TARJAN = 0
FPX_DG_ZIGZAG = 1


class FpxDepgraphConfig(C.Structure):
    _fields_ = [("kind", C.c_int32), ("num_leaders", C.c_int32), ("gc_every_n", C.c_int32)]


VP = C.c_void_p
I64P = C.POINTER(C.c_int64)
# every symbol include/fpx_depgraph.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "fpx_depgraph_create": (C.c_int32, [C.POINTER(FpxDepgraphConfig), C.POINTER(VP)]),
    "fpx_depgraph_destroy": (C.c_int32, [VP]),
    "fpx_depgraph_commit": (C.c_int32, [VP, C.c_int32] + [VP] * 7),
    "fpx_depgraph_commit_epx": (C.c_int32, [VP, C.c_int32, VP, VP, VP, VP, VP, C.c_int32, VP]),
    "fpx_depgraph_update_executed": (C.c_int32, [VP, VP, C.c_int32, VP, VP]),
    "fpx_depgraph_execute": (C.c_int32, [VP, C.c_int32, I64P, I64P, I64P]),
    "fpx_depgraph_read_result": (C.c_int32, [VP] * 6),
    "fpx_depgraph_num_vertices": (C.c_int64, [VP]),
    "fpx_depgraph_executed_watermark": (C.c_int32, [VP, VP]),
}


def _bind(L):
    if getattr(L, "_dg_bound", False):
        return
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L._dg_bound = True


def _a32(x):
    return None if x is None else np.ascontiguousarray(x, dtype=np.int32)


def _p(a):
    return None if a is None else a.ctypes.data


class DependencyGraph:
    """depgraph.DependencyGraph[(leader, id), Int, InstancePrefixSet] (DependencyGraph.scala:126-192)"""

    def __init__(self, num_leaders, kind=FPX_DG_ZIGZAG, gc_every_n=0):
        self.L = _lib.lib()
        _bind(self.L)
        self.num_leaders = num_leaders
        cfg = FpxDepgraphConfig(kind, num_leaders, gc_every_n)
        h = VP()
        st = self.L.fpx_depgraph_create(C.byref(cfg), C.byref(h))
        if st:
            raise FpxError(st, "fpx_depgraph_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.L.fpx_depgraph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def commit(self, leader, id, seq, dep_watermark, dep_values=None):
        """n vertices; dep_watermark [n, num_leaders]; dep_values: None or a list of n lists of (leader, id)"""
        leader, id, seq, wm = _a32(leader), _a32(id), _a32(seq), _a32(dep_watermark)
        off = vl = vi = None
        if dep_values is not None:
            off = np.zeros(len(leader) + 1, np.int64)
            off[1:] = np.cumsum([len(v) for v in dep_values])
            flat = [p for v in dep_values for p in v]
            vl = np.array([p[0] for p in flat], np.int32)
            vi = np.array([p[1] for p in flat], np.int32)
        st = self.L.fpx_depgraph_commit(self._h, len(leader), _p(leader), _p(id), _p(seq), _p(wm), _p(off), _p(vl),
                                        _p(vi))
        if st:
            raise FpxError(st, "fpx_depgraph_commit")

    def commit_epx(self, leader, id, deps, own_values_end=None, mask=None, seq=None):
        """the outputs of EPaxos.preaccept / handle_preaccept as they are: deps [m, n], own_values_end [m] or
        [m, 2] (column 0 is used), mask = fast / committed flags"""
        leader, id, deps, seq = _a32(leader), _a32(id), _a32(deps), _a32(seq)
        own = _a32(own_values_end)
        stride = 0 if own is None else (1 if own.ndim == 1 else own.shape[1])
        mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        st = self.L.fpx_depgraph_commit_epx(self._h, len(leader), _p(leader), _p(id), _p(seq), _p(deps), _p(own),
                                            stride, _p(mask))
        if st:
            raise FpxError(st, "fpx_depgraph_commit_epx")

    def update_executed(self, watermark=None, keys=()):
        wm = _a32(watermark)
        keys = list(keys)
        kl, ki = _a32([k[0] for k in keys]), _a32([k[1] for k in keys])
        st = self.L.fpx_depgraph_update_executed(self._h, _p(wm), len(keys), _p(kl), _p(ki))
        if st:
            raise FpxError(st, "fpx_depgraph_update_executed")

    def execute_arrays(self, num_blockers=None):
        """(exec_leader, exec_id, component_size, blocker_leader, blocker_id) as int32 arrays"""
        ne, nc, nb = C.c_int64(), C.c_int64(), C.c_int64()
        st = self.L.fpx_depgraph_execute(self._h, -1 if num_blockers is None else num_blockers, C.byref(ne),
                                         C.byref(nc), C.byref(nb))
        if st:
            raise FpxError(st, "fpx_depgraph_execute")
        el, ei = np.zeros(ne.value, np.int32), np.zeros(ne.value, np.int32)
        cs = np.zeros(nc.value, np.int32)
        bl, bi = np.zeros(nb.value, np.int32), np.zeros(nb.value, np.int32)
        st = self.L.fpx_depgraph_read_result(self._h, _p(el), _p(ei), _p(cs), _p(bl), _p(bi))
        if st:
            raise FpxError(st, "fpx_depgraph_read_result")
        return el, ei, cs, bl, bi

    def execute_by_component(self, num_blockers=None):
        """(components: list of lists of (leader, id), blockers: set of (leader, id))"""
        el, ei, cs, bl, bi = self.execute_arrays(num_blockers)
        keys = list(zip(el.tolist(), ei.tolist()))
        comps, at = [], 0
        for c in cs.tolist():
            comps.append(keys[at:at + c])
            at += c
        return comps, set(zip(bl.tolist(), bi.tolist()))

    def execute(self, num_blockers=None):
        comps, blockers = self.execute_by_component(num_blockers)
        return [k for c in comps for k in c], blockers

    @property
    def num_vertices(self):
        return int(self.L.fpx_depgraph_num_vertices(self._h))

    def executed_watermark(self):
        wm = np.zeros(self.num_leaders, np.int32)
        st = self.L.fpx_depgraph_executed_watermark(self._h, _p(wm))
        if st:
            raise FpxError(st, "fpx_depgraph_executed_watermark")
        return wm
