"""ctypes loader for the CPU oracle (oracle/libfpx_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py -- never by the frankenpaxos_amd package.  See oracle/fpx_oracle.h for the parity status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfpx_oracle.so")


class Config(C.Structure):
    """Layout-identical to fpx_config / fpo_config."""

    _fields_ = [
        ("num_slots", C.c_int32),
        ("num_replicas", C.c_int32),
        ("num_groups", C.c_int32),
        ("num_leader_groups", C.c_int32),
        ("f", C.c_int32),
        ("quorum_kind", C.c_int32),
        ("grid_rows", C.c_int32),
        ("grid_cols", C.c_int32),
        ("num_leaders", C.c_int32),
        ("ballot_mode", C.c_int32),
        ("tally_ways", C.c_int32),
        ("replica_base", C.c_int32),
        ("replicas_total", C.c_int32),
        ("device", C.c_int32),
        ("flags", C.c_uint32),
    ]


def make_config(num_slots, num_replicas, num_groups=1, num_leader_groups=1, f=0, quorum_kind=0,
                grid_rows=0, grid_cols=0, num_leaders=2, ballot_mode=0, tally_ways=4,
                replica_base=0, replicas_total=0, device=0, flags=0):
    return Config(num_slots, num_replicas, num_groups, num_leader_groups, f, quorum_kind, grid_rows,
                  grid_cols, num_leaders, ballot_mode, tally_ways, replica_base, replicas_total,
                  device, flags)


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("fpx_oracle.c", "fpx_oracle_epaxos.c", "fpx_oracle.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libfpx_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _SO


_lib = None

I32P = C.POINTER(C.c_int32)
U64P = C.POINTER(C.c_uint64)
U8P = C.POINTER(C.c_uint8)
INTP = C.POINTER(C.c_int)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    L.fpo_round_leader.argtypes = [C.c_int, C.c_int]
    L.fpo_next_classic_round.argtypes = [C.c_int, C.c_int, C.c_int]
    for name in ("fpo_qs_simple_majority", "fpo_qs_unanimous_writes"):
        getattr(L, name).argtypes = [INTP, C.c_int]
        getattr(L, name).restype = C.c_void_p
    L.fpo_qs_grid.argtypes = [INTP, C.c_int, C.c_int]
    L.fpo_qs_grid.restype = C.c_void_p
    L.fpo_qs_free.argtypes = [C.c_void_p]
    L.fpo_qs_nodes.argtypes = [C.c_void_p, INTP]
    for name in ("fpo_qs_is_read_quorum", "fpo_qs_is_write_quorum",
                 "fpo_qs_is_superset_of_read_quorum", "fpo_qs_is_superset_of_write_quorum"):
        getattr(L, name).argtypes = [C.c_void_p, INTP, C.c_int]
    for name in ("fpo_qs_random_read_quorum", "fpo_qs_random_write_quorum"):
        getattr(L, name).argtypes = [C.c_void_p, U64P, INTP]
    L.fpo_splitmix64.argtypes = [U64P]
    L.fpo_splitmix64.restype = C.c_uint64
    L.fpo_log_new.argtypes = [C.c_int]
    L.fpo_log_new.restype = C.c_void_p
    L.fpo_log_free.argtypes = [C.c_void_p]
    L.fpo_log_get.argtypes = [C.c_void_p, C.c_int, INTP]
    L.fpo_log_put.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.fpo_log_garbage_collect.argtypes = [C.c_void_p, C.c_int]
    L.fpo_log_chosen.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.fpo_log_executed_watermark.argtypes = [C.c_void_p]
    L.fpo_log_largest_key.argtypes = [C.c_void_p]
    L.fpo_config_check.argtypes = [C.POINTER(Config)]
    L.fpo_sys_new.argtypes = [C.POINTER(Config)]
    L.fpo_sys_new.restype = C.c_void_p
    L.fpo_sys_free.argtypes = [C.c_void_p]
    L.fpo_sys_reset.argtypes = [C.c_void_p]
    L.fpo_group_of_slot.argtypes = [C.POINTER(Config), C.c_int]
    L.fpo_acceptor_handle_phase2a.argtypes = [C.c_void_p] + [C.c_int] * 5 + [INTP]
    L.fpo_acceptor_handle_phase1a.argtypes = [C.c_void_p] + [C.c_int] * 4 + [INTP]
    L.fpo_proxy_handle_phase2a.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.fpo_proxy_handle_phase2b.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, INTP]
    L.fpo_sys_is_write_quorum.argtypes = [C.POINTER(Config), U64P, C.c_int]
    L.fpo_sys_is_read_quorum.argtypes = [C.POINTER(Config), U64P, C.c_int]
    L.fpo_acceptor_phase2a.argtypes = [C.c_void_p, C.c_int32, I32P, I32P, I32P, U64P, U64P, U64P,
                                       I32P]
    L.fpo_acceptor_phase1a.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, U64P, U64P, U64P]
    L.fpo_proxy_open.argtypes = [C.c_void_p, C.c_int32, I32P, I32P, I32P, U8P]
    L.fpo_proxy_phase2b.argtypes = [C.c_void_p, C.c_int32, I32P, I32P, U64P, U8P, I32P, I32P]
    for name in ("fpo_phase2_fused", "fpo_phase2_fifo_pump"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_int32, I32P, I32P, I32P, U64P, U8P, I32P, I32P,
                                     I32P]
    L.fpo_acceptor_phase2a_noop_range.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, U64P, U64P,
                                                  U64P, I32P]
    L.fpo_proxy_open_noop_range.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, U8P]
    L.fpo_proxy_phase2b_noop_range.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, U64P, U8P]
    L.fpo_acceptor_phase2a_noop_ranges.argtypes = [C.c_void_p, C.c_int32, I32P, I32P, I32P, U64P, U64P, U64P, I32P]
    L.fpo_proxy_open_noop_ranges.argtypes = [C.c_void_p, C.c_int32, I32P, I32P, I32P, U8P]
    L.fpo_proxy_phase2b_noop_ranges.argtypes = [C.c_void_p, C.c_int32, I32P, I32P, I32P, U64P, U8P]
    L.fpo_noop_ranges_fused.argtypes = [C.c_void_p, C.c_int32, I32P, I32P, I32P, U64P, U64P, U64P, I32P, U8P, U8P]
    L.fpo_read_range_tally.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, I32P, U64P]
    L.fpo_recycle_slots.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.fpo_proxy_forget.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.fpo_replica_chosen.argtypes = [C.c_void_p, C.c_int32, I32P, I32P, U8P, I32P, I32P]
    L.fpo_replica_chosen_noop_range.argtypes = [C.c_void_p, C.c_int32, C.c_int32, I32P, I32P]
    L.fpo_replica_read_log.argtypes = [C.c_void_p, C.c_int32, C.c_int32, I32P, U8P]
    L.fpo_leader_phase1b_scan.argtypes = [C.c_void_p, C.c_int32, U64P, C.c_int32, I32P, I32P, I32P]
    L.fpo_acceptor_phase1b_info.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, I32P, I32P, I32P, I32P]
    L.fpo_error_detail.argtypes = [C.c_void_p, I32P, I32P, I32P]
    L.fpo_read_acceptor.argtypes = [C.c_void_p, C.c_int32, C.c_int32, I32P, I32P, I32P, I32P, I32P]
    L.fpo_read_state.argtypes = [C.c_void_p, I32P, I32P, I32P]
    L.fpo_read_scalars.argtypes = [C.c_void_p, I32P, I32P]
    L.fpo_read_tally.argtypes = [C.c_void_p, C.c_int32, I32P, I32P, I32P, I32P, U64P]
    L.fpo_state_digest.argtypes = [C.c_void_p, U64P]
    _lib = L
    return L


def _p(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def _u64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint64)


def ints(xs):
    xs = list(xs)
    arr = (C.c_int * max(len(xs), 1))(*xs)
    return arr, len(xs)


class QuorumSystem:
    """Set[Int]-based quorum system of the oracle (SimpleMajority / UnanimousWrites / Grid)."""

    def __init__(self, handle, max_nodes):
        if not handle:
            raise ValueError("require() failed constructing the quorum system")
        self._h = handle
        self._max = max_nodes

    @classmethod
    def simple_majority(cls, members):
        a, n = ints(members)
        return cls(lib().fpo_qs_simple_majority(a, n), n)

    @classmethod
    def unanimous_writes(cls, members):
        a, n = ints(members)
        return cls(lib().fpo_qs_unanimous_writes(a, n), n)

    @classmethod
    def grid(cls, rows):
        rows = [list(r) for r in rows]
        flat = [x for r in rows for x in r]
        a, _ = ints(flat)
        return cls(lib().fpo_qs_grid(a, len(rows), len(rows[0]) if rows else 0), len(flat))

    def __del__(self):
        try:
            lib().fpo_qs_free(self._h)
        except Exception:
            pass

    def _call(self, name, xs):
        a, n = ints(xs)
        r = getattr(lib(), name)(self._h, a, n)
        if r < 0:
            raise ValueError("IllegalArgumentException (require failed)")
        return bool(r)

    def nodes(self):
        out = (C.c_int * self._max)()
        n = lib().fpo_qs_nodes(self._h, out)
        return set(out[:n])

    def is_read_quorum(self, xs):
        return self._call("fpo_qs_is_read_quorum", xs)

    def is_write_quorum(self, xs):
        return self._call("fpo_qs_is_write_quorum", xs)

    def is_superset_of_read_quorum(self, xs):
        return self._call("fpo_qs_is_superset_of_read_quorum", xs)

    def is_superset_of_write_quorum(self, xs):
        return self._call("fpo_qs_is_superset_of_write_quorum", xs)

    def _rand(self, name, rng):
        out = (C.c_int * self._max)()
        st = C.c_uint64(rng[0])
        n = getattr(lib(), name)(self._h, C.byref(st), out)
        rng[0] = st.value
        return set(out[:n])

    def random_read_quorum(self, rng):
        return self._rand("fpo_qs_random_read_quorum", rng)

    def random_write_quorum(self, rng):
        return self._rand("fpo_qs_random_write_quorum", rng)


class Log:
    """util.BufferMap + Replica executed-watermark."""

    def __init__(self, grow_size=5000):
        self._h = lib().fpo_log_new(grow_size)

    def __del__(self):
        try:
            lib().fpo_log_free(self._h)
        except Exception:
            pass

    def get(self, key):
        v = C.c_int(0)
        return v.value if lib().fpo_log_get(self._h, key, C.byref(v)) else None

    def put(self, key, value):
        lib().fpo_log_put(self._h, key, value)

    def garbage_collect(self, watermark):
        lib().fpo_log_garbage_collect(self._h, watermark)

    def chosen(self, slot, value):
        return lib().fpo_log_chosen(self._h, slot, value)

    def iterator_from(self, key=0):
        """BufferMap.iteratorFrom (BufferMap.scala:64-103): (key, value) pairs from `key` to largestKey."""
        out = []
        for k in range(key, lib().fpo_log_largest_key(self._h) + 1):
            v = self.get(k)
            if v is not None:
                out.append((k, v))
        return out

    @property
    def executed_watermark(self):
        return lib().fpo_log_executed_watermark(self._h)


class System:
    """The oracle's Phase-2 system; method-for-method the same surface as frankenpaxos_amd.Context."""

    def __init__(self, cfg):
        self.cfg = cfg
        if lib().fpo_config_check(C.byref(cfg)) != 0:
            raise ValueError("FPX_EINVAL: bad config")
        self._h = lib().fpo_sys_new(C.byref(cfg))
        self.S, self.R = cfg.num_slots, cfg.num_replicas
        self.ngroups = cfg.num_groups * cfg.num_leader_groups

    def __del__(self):
        try:
            lib().fpo_sys_free(self._h)
        except Exception:
            pass

    def reset(self):
        lib().fpo_sys_reset(self._h)

    def group_of_slot(self, slot):
        return lib().fpo_group_of_slot(C.byref(self.cfg), slot)

    # single-message handlers
    def acceptor_handle_phase2a(self, group, replica, slot, round_, value):
        rr = C.c_int(0)
        ok = lib().fpo_acceptor_handle_phase2a(self._h, group, replica, slot, round_, value,
                                               C.byref(rr))
        return bool(ok), rr.value

    def acceptor_handle_phase1a(self, group, replica, round_, watermark=0):
        rr = C.c_int(0)
        ok = lib().fpo_acceptor_handle_phase1a(self._h, group, replica, round_, watermark,
                                               C.byref(rr))
        return bool(ok), rr.value

    def proxy_handle_phase2a(self, slot, round_, value):
        return bool(lib().fpo_proxy_handle_phase2a(self._h, slot, round_, value))

    def proxy_handle_phase2b(self, acceptor_bit, slot, round_):
        v = C.c_int(-1)
        rc = lib().fpo_proxy_handle_phase2b(self._h, acceptor_bit, slot, round_, C.byref(v))
        return rc, v.value

    # batch entry points
    def acceptor_phase2a(self, slot, round_, value, target_mask=None):
        slot, round_, value, target_mask = _i32(slot), _i32(round_), _i32(value), _u64(target_mask)
        n = len(slot)
        vb = np.zeros((n, 4), np.uint64)
        nb = np.zeros((n, 4), np.uint64)
        nr = np.zeros(n, np.int32)
        st = lib().fpo_acceptor_phase2a(self._h, n, _p(slot, I32P), _p(round_, I32P),
                                        _p(value, I32P), _p(target_mask, U64P), _p(vb, U64P),
                                        _p(nb, U64P), _p(nr, I32P))
        return st, vb, nb, nr

    def acceptor_phase1a(self, group, round_, watermark=0, target_mask=None):
        target_mask = _u64(target_mask)
        pb = np.zeros(4, np.uint64)
        nb = np.zeros(4, np.uint64)
        st = lib().fpo_acceptor_phase1a(self._h, group, round_, watermark, _p(target_mask, U64P),
                                        _p(pb, U64P), _p(nb, U64P))
        return st, pb, nb

    def proxy_open(self, slot, round_, value):
        slot, round_, value = _i32(slot), _i32(round_), _i32(value)
        n = len(slot)
        new = np.zeros(n, np.uint8)
        st = lib().fpo_proxy_open(self._h, n, _p(slot, I32P), _p(round_, I32P), _p(value, I32P),
                                  _p(new, U8P))
        return st, new

    def proxy_phase2b(self, slot, round_, vote_bits):
        slot, round_, vote_bits = _i32(slot), _i32(round_), _u64(vote_bits)
        n = len(slot)
        ch = np.zeros(n, np.uint8)
        cr = np.zeros(n, np.int32)
        cv = np.zeros(n, np.int32)
        st = lib().fpo_proxy_phase2b(self._h, n, _p(slot, I32P), _p(round_, I32P),
                                     _p(vote_bits, U64P), _p(ch, U8P), _p(cr, I32P), _p(cv, I32P))
        return st, ch, cr, cv

    def _fused(self, fn, slot, round_, value, target_mask):
        slot, round_, value, target_mask = _i32(slot), _i32(round_), _i32(value), _u64(target_mask)
        n = len(slot)
        ch = np.zeros(n, np.uint8)
        cr = np.zeros(n, np.int32)
        cv = np.zeros(n, np.int32)
        nr = np.zeros(n, np.int32)
        st = fn(self._h, n, _p(slot, I32P), _p(round_, I32P), _p(value, I32P),
                _p(target_mask, U64P), _p(ch, U8P), _p(cr, I32P), _p(cv, I32P), _p(nr, I32P))
        return st, ch, cr, cv, nr

    def phase2_fused(self, slot, round_, value, target_mask=None):
        return self._fused(lib().fpo_phase2_fused, slot, round_, value, target_mask)

    def phase2_fifo_pump(self, slot, round_, value, target_mask=None):
        return self._fused(lib().fpo_phase2_fifo_pump, slot, round_, value, target_mask)

    def acceptor_phase2a_noop_range(self, slot_start, slot_end, round_, target_masks=None):
        A = self.cfg.num_groups
        target_masks = _u64(target_masks)
        vb = np.zeros((A, 4), np.uint64)
        nb = np.zeros((A, 4), np.uint64)
        nr = C.c_int32(-1)
        st = lib().fpo_acceptor_phase2a_noop_range(self._h, slot_start, slot_end, round_,
                                                   _p(target_masks, U64P), _p(vb, U64P), _p(nb, U64P),
                                                   C.byref(nr))
        return st, vb, nb, nr.value

    def proxy_open_noop_range(self, slot_start, slot_end, round_):
        new = C.c_uint8(0)
        st = lib().fpo_proxy_open_noop_range(self._h, slot_start, slot_end, round_, C.byref(new))
        return st, new.value

    def proxy_phase2b_noop_range(self, slot_start, slot_end, round_, vote_bits):
        vote_bits = _u64(vote_bits)
        ch = C.c_uint8(0)
        st = lib().fpo_proxy_phase2b_noop_range(self._h, slot_start, slot_end, round_,
                                                _p(vote_bits, U64P), C.byref(ch))
        return st, ch.value

    def _ranges(self, slot_start, slot_end, round_):
        return _i32(np.atleast_1d(slot_start)), _i32(np.atleast_1d(slot_end)), _i32(np.atleast_1d(round_))

    def acceptor_phase2a_noop_ranges(self, slot_start, slot_end, round_, target_masks=None):
        s, e, r = self._ranges(slot_start, slot_end, round_)
        n, A = len(s), self.cfg.num_groups
        target_masks = _u64(target_masks)
        vb = np.zeros((n, A, 4), np.uint64)
        nb = np.zeros((n, A, 4), np.uint64)
        nr = np.full(n, -1, np.int32)
        st = lib().fpo_acceptor_phase2a_noop_ranges(self._h, n, _p(s, I32P), _p(e, I32P), _p(r, I32P),
                                                    _p(target_masks, U64P), _p(vb, U64P), _p(nb, U64P), _p(nr, I32P))
        return st, vb, nb, nr

    def proxy_open_noop_ranges(self, slot_start, slot_end, round_):
        s, e, r = self._ranges(slot_start, slot_end, round_)
        new = np.zeros(len(s), np.uint8)
        st = lib().fpo_proxy_open_noop_ranges(self._h, len(s), _p(s, I32P), _p(e, I32P), _p(r, I32P), _p(new, U8P))
        return st, new

    def proxy_phase2b_noop_ranges(self, slot_start, slot_end, round_, vote_bits):
        s, e, r = self._ranges(slot_start, slot_end, round_)
        vote_bits = _u64(vote_bits)
        ch = np.zeros(len(s), np.uint8)
        st = lib().fpo_proxy_phase2b_noop_ranges(self._h, len(s), _p(s, I32P), _p(e, I32P), _p(r, I32P),
                                                 _p(vote_bits, U64P), _p(ch, U8P))
        return st, ch

    def noop_ranges_fused(self, slot_start, slot_end, round_, target_masks=None):
        s, e, r = self._ranges(slot_start, slot_end, round_)
        n, A = len(s), self.cfg.num_groups
        target_masks = _u64(target_masks)
        vb = np.zeros((n, A, 4), np.uint64)
        nb = np.zeros((n, A, 4), np.uint64)
        nr = np.full(n, -1, np.int32)
        new = np.zeros(n, np.uint8)
        ch = np.zeros(n, np.uint8)
        st = lib().fpo_noop_ranges_fused(self._h, n, _p(s, I32P), _p(e, I32P), _p(r, I32P), _p(target_masks, U64P),
                                         _p(vb, U64P), _p(nb, U64P), _p(nr, I32P), _p(new, U8P), _p(ch, U8P))
        return st, vb, nb, nr, new, ch

    def read_range_tally(self, slot_start, slot_end, round_):
        state = C.c_int32()
        bits = np.zeros((self.cfg.num_groups, 4), np.uint64)
        lib().fpo_read_range_tally(self._h, slot_start, slot_end, round_, C.byref(state), _p(bits, U64P))
        return state.value, bits

    def proxy_forget(self, first_slot, count):
        st = lib().fpo_proxy_forget(self._h, first_slot, count)
        if st:
            raise ValueError("FPX_EINVAL")

    def recycle_slots(self, first_slot, count):
        st = lib().fpo_recycle_slots(self._h, first_slot, count)
        if st:
            raise ValueError("FPX_EINVAL")

    def replica_chosen(self, slot, value, mask=None):
        slot, value = _i32(slot), _i32(value)
        mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        wm, nc = C.c_int32(), C.c_int32()
        st = lib().fpo_replica_chosen(self._h, len(slot), _p(slot, I32P), _p(value, I32P),
                                      _p(mask, U8P), C.byref(wm), C.byref(nc))
        return st, wm.value, nc.value

    def replica_chosen_noop_range(self, slot_start, slot_end):
        wm, nc = C.c_int32(), C.c_int32()
        st = lib().fpo_replica_chosen_noop_range(self._h, slot_start, slot_end, C.byref(wm), C.byref(nc))
        return st, wm.value, nc.value

    def replica_read_log(self, first, count):
        vals = np.zeros(count, np.int32)
        pres = np.zeros(count, np.uint8)
        lib().fpo_replica_read_log(self._h, first, count, _p(vals, I32P), _p(pres, U8P))
        return vals, pres

    def leader_phase1b_scan(self, watermark, quorum_masks, cap):
        q = np.ascontiguousarray(quorum_masks, dtype=np.uint64).reshape(self.ngroups, 4)
        mx = C.c_int32()
        sr = np.full(cap, -7, np.int32)
        sv = np.full(cap, -7, np.int32)
        st = lib().fpo_leader_phase1b_scan(self._h, watermark, _p(q, U64P), cap, C.byref(mx),
                                           _p(sr, I32P), _p(sv, I32P))
        k = max(0, min(cap, mx.value - watermark + 1))
        return st, mx.value, sr[:k], sv[:k]

    def acceptor_phase1b_info(self, group, replica, watermark=0):
        k = C.c_int32()
        cap = self.S
        sl, vr, vv = (np.zeros(cap, np.int32) for _ in range(3))
        st = lib().fpo_acceptor_phase1b_info(self._h, group, replica, watermark, cap, C.byref(k), _p(sl, I32P),
                                             _p(vr, I32P), _p(vv, I32P))
        if st:
            raise ValueError("FPX_EINVAL")
        return sl[:k.value], vr[:k.value], vv[:k.value]

    def error_detail(self):
        i, s, r = C.c_int32(), C.c_int32(), C.c_int32()
        lib().fpo_error_detail(self._h, C.byref(i), C.byref(s), C.byref(r))
        return i.value, s.value, r.value

    # readback
    def read_acceptor(self, group, replica):
        p, m = C.c_int32(), C.c_int32()
        vr = np.zeros(self.S, np.int32)
        vv = np.zeros(self.S, np.int32)
        bl = np.zeros(self.S, np.int32)
        st = lib().fpo_read_acceptor(self._h, group, replica, C.byref(p), C.byref(m), _p(vr, I32P),
                                     _p(vv, I32P), _p(bl, I32P))
        if st:
            raise ValueError("FPX_EINVAL")
        return p.value, m.value, vr, vv, bl

    def read_state(self):
        vr = np.zeros((self.S, self.R), np.int32)
        vv = np.zeros((self.S, self.R), np.int32)
        bl = np.zeros((self.S, self.R), np.int32)
        lib().fpo_read_state(self._h, _p(vr, I32P), _p(vv, I32P), _p(bl, I32P))
        return vr, vv, bl

    def read_scalars(self):
        pr = np.zeros((self.ngroups, self.R), np.int32)
        mv = np.zeros((self.ngroups, self.R), np.int32)
        lib().fpo_read_scalars(self._h, _p(pr, I32P), _p(mv, I32P))
        return pr, mv

    def state_digest(self):
        out = np.zeros(8, np.uint64)
        lib().fpo_state_digest(self._h, _p(out, U64P))
        return out

    def read_tally(self, slot):
        n = C.c_int32()
        rounds = np.zeros(64, np.int32)
        states = np.zeros(64, np.int32)
        values = np.zeros(64, np.int32)
        bits = np.zeros((64, 4), np.uint64)
        lib().fpo_read_tally(self._h, slot, C.byref(n), _p(rounds, I32P), _p(states, I32P),
                             _p(values, I32P), _p(bits, U64P))
        k = n.value
        return [(int(rounds[i]), int(states[i]), int(values[i]), tuple(int(x) for x in bits[i]))
                for i in range(k)]

    def is_write_quorum(self, nodes, strict=True):
        nodes = _u64(nodes)
        r = lib().fpo_sys_is_write_quorum(C.byref(self.cfg), _p(nodes, U64P), int(strict))
        if r < 0:
            raise ValueError("IllegalArgumentException (require failed)")
        return bool(r)

    def is_read_quorum(self, nodes, strict=True):
        nodes = _u64(nodes)
        r = lib().fpo_sys_is_read_quorum(C.byref(self.cfg), _p(nodes, U64P), int(strict))
        if r < 0:
            raise ValueError("IllegalArgumentException (require failed)")
        return bool(r)


def bits_of(indices):
    """256-bit set as 4 little-endian uint64 words."""
    v = np.zeros(4, np.uint64)
    for j in indices:
        v[j >> 6] |= np.uint64(1) << np.uint64(j & 63)
    return v


def indices_of(words):
    return [j for j in range(256) if (int(words[j >> 6]) >> (j & 63)) & 1]


class EPaxos:
    """oracle twin of frankenpaxos_amd.epaxos.EPaxos (fpx_oracle_epaxos.c)"""

    def __init__(self, num_replicas, num_keys, num_instances=0):
        L = lib()
        L.fpo_epx_new2.argtypes = [C.c_int, C.c_int, C.c_int]
        L.fpo_epx_new2.restype = C.c_void_p
        L.fpo_epx_preaccept2.argtypes = [C.c_void_p, C.c_int32, I32P, I32P, I32P, U8P, U8P, U8P, I32P, I32P, U8P, I32P,
                                         I32P, I32P]
        L.fpo_epx_prepare.argtypes = [C.c_void_p, C.c_int32, I32P, I32P, I32P, I32P, U8P, U8P, U8P, U8P, I32P, I32P, I32P, I32P]
        L.fpo_epx_accept.argtypes = [C.c_void_p, C.c_int32, I32P, I32P, I32P, I32P, I32P, I32P, U8P, U8P, U8P, U8P, U8P, I32P, U8P]
        L.fpo_epx_read_cmdlog.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, I32P]
        L.fpo_epx_handle_prepare_oks.argtypes = [C.c_void_p, C.c_int32, I32P, I32P, I32P, I32P, U8P, I32P, I32P, I32P, C.c_int32, I32P, I32P, I32P]
        L.fpo_epx_read_cmdlog_deps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, I32P, I32P]
        L.fpo_epx_handle_preaccept.argtypes = [C.c_void_p, C.c_int32, I32P, I32P, I32P, I32P, I32P, U8P, I32P, I32P, I32P,
                                               U8P, U8P, U8P, U8P, U8P, I32P, I32P, I32P, I32P]
        L.fpo_epx_free.argtypes = [C.c_void_p]
        L.fpo_epx_preaccept.argtypes = [C.c_void_p, C.c_int32, I32P, I32P, I32P, U8P, U8P, U8P, I32P, U8P, I32P,
                                        I32P, I32P]
        L.fpo_epx_read_index.argtypes = [C.c_void_p, C.c_int, C.c_int, I32P, I32P]
        L.fpo_epx_index_put.argtypes = [C.c_void_p] + [C.c_int] * 5
        L.fpo_epx_index_conflicts.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, I32P]
        self.n, self.num_keys = num_replicas, num_keys
        self._h = L.fpo_epx_new2(num_replicas, num_keys, num_instances)
        if not self._h:
            raise ValueError("FPX_EINVAL")

    def __del__(self):
        try:
            lib().fpo_epx_free(self._h)
        except Exception:
            pass

    def prepare(self, leader, number, ballot_ordering, ballot_replica, target_mask):
        leader, number, bo, br = _i32(leader), _i32(number), _i32(ballot_ordering), _i32(ballot_replica)
        tgt = np.ascontiguousarray(target_mask, dtype=np.uint8)
        m = len(leader)
        ok, nack, com = (np.zeros(m, np.uint8) for _ in range(3))
        nb = np.full(m, -1, np.int32)
        rs, rv, rt = (np.full((m, self.n), -1, np.int32) for _ in range(3))
        st = lib().fpo_epx_prepare(self._h, m, _p(leader, I32P), _p(number, I32P), _p(bo, I32P), _p(br, I32P), _p(tgt, U8P),
                                   _p(ok, U8P), _p(nack, U8P), _p(com, U8P), _p(nb, I32P), _p(rs, I32P), _p(rv, I32P),
                                   _p(rt, I32P))
        return st, ok, nack, com, nb, rs, rv, rt

    def handle_prepare_oks(self, leader, number, ballot_ordering, ballot_replica, resp_mask, reply_status, reply_vote,
                           reply_triple, as_intended=False):
        """Replica.handlePrepareOk: (status, action, source, triple)"""
        leader, number, bo, br = _i32(leader), _i32(number), _i32(ballot_ordering), _i32(ballot_replica)
        mask = np.ascontiguousarray(resp_mask, dtype=np.uint8)
        rs, rv, rt = _i32(reply_status), _i32(reply_vote), _i32(reply_triple)
        m = len(leader)
        act, src, tr = (np.full(m, -9, np.int32) for _ in range(3))
        st = lib().fpo_epx_handle_prepare_oks(self._h, m, _p(leader, I32P), _p(number, I32P), _p(bo, I32P), _p(br, I32P),
                                              _p(mask, U8P), _p(rs, I32P), _p(rv, I32P), _p(rt, I32P), int(as_intended),
                                              _p(act, I32P), _p(src, I32P), _p(tr, I32P))
        return st, act, src, tr

    def accept(self, leader, number, ballot_ordering, ballot_replica, triple_id, target_mask, key=None, is_set=None):
        """key / is_set: the triples' commands; key None = every triple is a Noop"""
        leader, number, bo, br, tr = _i32(leader), _i32(number), _i32(ballot_ordering), _i32(ballot_replica), _i32(triple_id)
        tgt = np.ascontiguousarray(target_mask, dtype=np.uint8)
        m = len(leader)
        key = np.full(m, -1, np.int32) if key is None else _i32(key)
        is_set = np.zeros(m, np.uint8) if is_set is None else np.ascontiguousarray(is_set, dtype=np.uint8)
        ok, nack, com, done = (np.zeros(m, np.uint8) for _ in range(4))
        nb = np.full(m, -1, np.int32)
        st = lib().fpo_epx_accept(self._h, m, _p(leader, I32P), _p(number, I32P), _p(bo, I32P), _p(br, I32P), _p(tr, I32P),
                                  _p(key, I32P), _p(is_set, U8P), _p(tgt, U8P), _p(ok, U8P), _p(nack, U8P), _p(com, U8P),
                                  _p(nb, I32P), _p(done, U8P))
        return st, ok, nack, com, nb, done

    def handle_commit(self, leader, number, triple_id, target_mask, key=None, is_set=None, deps=None, deps_values_end=None):
        """Replica.handleCommit at the replicas of target_mask; key None = Noops, deps None = the triple by id alone"""
        leader, number, tr = _i32(leader), _i32(number), _i32(triple_id)
        m = len(leader)
        tgt = np.ascontiguousarray(target_mask, dtype=np.uint8)
        key = np.full(m, -1, np.int32) if key is None else _i32(key)
        is_set = np.zeros(m, np.uint8) if is_set is None else np.ascontiguousarray(is_set, dtype=np.uint8)
        d = None if deps is None else np.ascontiguousarray(deps, dtype=np.int32)
        de = None if deps_values_end is None else _i32(deps_values_end)
        fn = lib().fpo_epx_handle_commit
        fn.argtypes = [C.c_void_p, C.c_int32, I32P, I32P, I32P, I32P, U8P, I32P, I32P, U8P]
        return fn(self._h, m, _p(leader, I32P), _p(number, I32P), _p(tr, I32P), _p(key, I32P), _p(is_set, U8P), _p(d, I32P),
                  _p(de, I32P), _p(tgt, U8P))

    def read_cmdlog(self, replica, leader, number):
        out = np.zeros(5, np.int32)
        if lib().fpo_epx_read_cmdlog(self._h, replica, leader, number, _p(out, I32P)):
            raise ValueError("FPX_EINVAL")
        return tuple(int(x) for x in out)

    def read_cmdlog_deps(self, replica, leader, number):
        deps, end = np.zeros(self.n, np.int32), np.zeros(1, np.int32)
        if lib().fpo_epx_read_cmdlog_deps(self._h, replica, leader, number, _p(deps, I32P), _p(end, I32P)):
            raise ValueError("FPX_EINVAL")
        return deps, int(end[0])

    def handle_preaccept(self, leader, number, ballot_ordering, ballot_replica, key, is_set, triple_id, deps_in,
                         deps_in_values_end, target_mask):
        leader, number, bo, br, key = _i32(leader), _i32(number), _i32(ballot_ordering), _i32(ballot_replica), _i32(key)
        is_set = np.ascontiguousarray(is_set, dtype=np.uint8)
        tr = None if triple_id is None else _i32(triple_id)
        din = np.ascontiguousarray(deps_in, dtype=np.int32)
        dend = None if deps_in_values_end is None else _i32(deps_in_values_end)
        tgt = np.ascontiguousarray(target_mask, dtype=np.uint8)
        m = len(leader)
        ok, resend, nack, com = (np.zeros(m, np.uint8) for _ in range(4))
        nb = np.full(m, -1, np.int32)
        rd = np.zeros((m, self.n, self.n), np.int32)
        re = np.zeros((m, self.n), np.int32)
        rt = np.full((m, self.n), -1, np.int32)
        st = lib().fpo_epx_handle_preaccept(self._h, m, _p(leader, I32P), _p(number, I32P), _p(bo, I32P), _p(br, I32P),
                                            _p(key, I32P), _p(is_set, U8P), _p(tr, I32P), _p(din, I32P), _p(dend, I32P),
                                            _p(tgt, U8P), _p(ok, U8P), _p(resend, U8P), _p(nack, U8P), _p(com, U8P),
                                            _p(nb, I32P), _p(rd, I32P), _p(re, I32P), _p(rt, I32P))
        return st, ok, resend, nack, com, nb, rd, re, rt

    def preaccept(self, leader, number, key, is_set, resp_mask, rank, seen_mask=None, triple_id=None):
        a8 = lambda x: np.ascontiguousarray(x, dtype=np.uint8)
        leader, number, key, rank = _i32(leader), _i32(number), _i32(key), _i32(rank)
        is_set, resp_mask = a8(is_set), a8(resp_mask)
        seen_mask = None if seen_mask is None else a8(seen_mask)
        m = len(leader)
        fast = np.zeros(m, np.uint8)
        deps = np.zeros((m, self.n), np.int32)
        ldeps = np.zeros((m, self.n), np.int32)
        own = np.zeros((m, 2), np.int32)
        triple_id = None if triple_id is None else _i32(triple_id)
        st = lib().fpo_epx_preaccept2(self._h, m, _p(leader, I32P), _p(number, I32P), _p(key, I32P),
                                      _p(is_set, U8P), _p(resp_mask, U8P), _p(seen_mask, U8P), _p(rank, I32P),
                                      _p(triple_id, I32P), _p(fast, U8P),
                                      _p(deps, I32P), _p(ldeps, I32P), _p(own, I32P))
        return st, fast, deps, ldeps, own

    def index_put(self, replica, key, is_set, leader, number):
        """KeyValueStore conflict index put of one single-key command (KeyValueStore.scala:232-253)"""
        lib().fpo_epx_index_put(self._h, replica, key, int(is_set), leader, number)

    def index_conflicts(self, replica, key, is_set):
        """getTopOneConflicts of one single-key command (KeyValueStore.scala:259-302)"""
        out = np.zeros(self.n, np.int32)
        lib().fpo_epx_index_conflicts(self._h, replica, key, int(is_set), _p(out, I32P))
        return out.tolist()

    def read_index(self, replica, key):
        g = np.zeros(self.n, np.int32)
        s = np.zeros(self.n, np.int32)
        lib().fpo_epx_read_index(self._h, replica, key, _p(g, I32P), _p(s, I32P))
        return g, s


def top_one(num_leaders, puts=(), merge_with=None):
    """util.TopOne: vector after `puts` [(leaderIndex, id)...], optionally mergeEquals(other vector)"""
    L = lib()
    L.fpo_top_one_put.argtypes = [I32P, C.c_int, C.c_int]
    L.fpo_top_one_merge.argtypes = [I32P, I32P, C.c_int]
    v = np.zeros(num_leaders, np.int32)
    for li, ident in puts:
        L.fpo_top_one_put(_p(v, I32P), li, ident)
    if merge_with is not None:
        o = _i32(merge_with)
        L.fpo_top_one_merge(_p(v, I32P), _p(o, I32P), num_leaders)
    return v.tolist()


def popular_items(xs, n):
    """Util.popularItems over a list of ints or int tuples: the set of items appearing >= n times"""
    L = lib()
    L.fpo_popular_items.argtypes = [I32P, C.c_int, C.c_int, C.c_int, I32P]
    rows = [x if isinstance(x, (tuple, list)) else (x,) for x in xs]
    if not rows:
        return set()
    a = _i32(rows)
    out = np.zeros(len(rows), np.int32)
    k = L.fpo_popular_items(_p(a, I32P), len(rows), a.shape[1], n, _p(out, I32P))
    res = [tuple(int(v) for v in a[int(out[j])]) for j in range(k)]
    return set(r[0] if len(r) == 1 else r for r in res)
