"""Wire adapter (include/fpx_wire.h): the reference's protobuf messages of the Phase-2 path <-> SoA batches.
Thin ctypes binding of the C functions in libfpx.so; host code only (works without a GPU)."""
import ctypes as C

import numpy as np

from . import _lib

OTHER, PHASE2A, PHASE2B, PHASE1A, CHOSEN, NACK = range(6)

_bound = False


def _L():
    global _bound
    L = _lib.lib()
    if not _bound:
        VP = C.c_void_p
        I32P = C.POINTER(C.c_int32)
        for name in ("fpx_wire_decode_proxy_leader_inbound",):
            getattr(L, name).argtypes = [VP, VP, C.c_int32] + [VP] * 8 + [I32P]
        L.fpx_wire_decode_acceptor_inbound.argtypes = [VP, VP, C.c_int32] + [VP] * 7 + [I32P]
        L.fpx_wire_decode_replica_inbound.argtypes = [VP, VP, C.c_int32] + [VP] * 5 + [I32P]
        L.fpx_wire_phase2b_rows.argtypes = [C.c_int32, VP, VP, VP, VP, VP, C.c_int32, I32P, VP, VP, VP]
        L.fpx_wire_encode_proxy_leader_phase2a.argtypes = [VP, C.c_int64, C.c_int32, C.c_int32, VP, C.c_int32, C.c_int32]
        L.fpx_wire_encode_acceptor_phase2a.argtypes = [VP, C.c_int64, C.c_int32, C.c_int32, VP, C.c_int32, C.c_int32]
        L.fpx_wire_encode_acceptor_phase1a.argtypes = [VP, C.c_int64, C.c_int32, C.c_int32]
        L.fpx_wire_encode_proxy_leader_phase2b.argtypes = [VP, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.fpx_wire_encode_replica_chosen.argtypes = [VP, C.c_int64, C.c_int32, VP, C.c_int32, C.c_int32]
        L.fpx_wire_encode_leader_nack.argtypes = [VP, C.c_int64, C.c_int32]
        L.fpx_wire_encode_phase2b_batch.argtypes = [C.c_int32, VP, VP, VP, VP, C.c_int32, VP, C.c_int64, VP, C.c_int64]
        for name in ("fpx_wire_encode_proxy_leader_phase2a", "fpx_wire_encode_acceptor_phase2a",
                     "fpx_wire_encode_acceptor_phase1a", "fpx_wire_encode_proxy_leader_phase2b",
                     "fpx_wire_encode_replica_chosen", "fpx_wire_encode_leader_nack", "fpx_wire_encode_phase2b_batch"):
            getattr(L, name).restype = C.c_int64
        _bound = True
    return L


def pack(messages):
    """a tick's byte arrays -> (one contiguous uint8 buffer, int64 offsets[n + 1])"""
    offsets = np.zeros(len(messages) + 1, np.int64)
    np.cumsum([len(m) for m in messages], out=offsets[1:])
    buf = np.frombuffer(b"".join(messages), dtype=np.uint8).copy() if messages else np.zeros(1, np.uint8)
    return buf, offsets


def _decode(fn, messages, names):
    buf, off = pack(messages)
    n = len(messages)
    out = {k: (np.zeros(n, np.int64) if k == "value_off" else np.zeros(n, np.int32)) for k in names}
    bad = C.c_int32(-1)
    st = getattr(_L(), fn)(buf.ctypes.data, off.ctypes.data, n, *[out[k].ctypes.data for k in names], C.byref(bad))
    out["status"], out["bad_index"], out["buf"] = st, bad.value, buf
    return out


def decode_proxy_leader_inbound(messages):
    return _decode("fpx_wire_decode_proxy_leader_inbound", messages,
                   ["kind", "slot", "round", "is_noop", "value_off", "value_len", "group_index", "acceptor_index"])


def decode_acceptor_inbound(messages):
    return _decode("fpx_wire_decode_acceptor_inbound", messages,
                   ["kind", "slot", "round", "is_noop", "value_off", "value_len", "chosen_watermark"])


def decode_replica_inbound(messages):
    return _decode("fpx_wire_decode_replica_inbound", messages, ["kind", "slot", "is_noop", "value_off", "value_len"])


def phase2b_rows(d, grid_cols=0):
    """decoded ProxyLeaderInbound batch -> (slot, round, vote_bits[m, 4]) rows for fpx_proxy_phase2b"""
    n = len(d["kind"])
    m = C.c_int32()
    rs, rr = np.zeros(n, np.int32), np.zeros(n, np.int32)
    rb = np.zeros((max(n, 1), 4), np.uint64)
    st = _L().fpx_wire_phase2b_rows(n, d["kind"].ctypes.data, d["group_index"].ctypes.data,
                                    d["acceptor_index"].ctypes.data, d["slot"].ctypes.data, d["round"].ctypes.data,
                                    grid_cols, C.byref(m), rs.ctypes.data, rr.ctypes.data, rb.ctypes.data)
    if st:
        raise ValueError("FPX_EINVAL: Phase2b acceptor outside 0..255")
    return rs[:m.value], rr[:m.value], rb[:m.value]


def _enc(fn, *args):
    out = np.zeros(64, np.uint8)
    n = fn(out.ctypes.data, len(out), *args)
    if n < 0:
        out = np.zeros(-n, np.uint8)
        n = fn(out.ctypes.data, len(out), *args)
    return out[:n].tobytes()


def _val(value):
    if value is None:
        return None, 0, 1
    v = np.frombuffer(bytes(value), dtype=np.uint8)
    return (v.ctypes.data if len(v) else None), len(v), 0


def encode_proxy_leader_phase2a(slot, round_, value):
    """value: the serialised CommandBatchOrNoop, or None for Noop"""
    p, n, noop = _val(value)
    keep = np.frombuffer(bytes(value), dtype=np.uint8) if value else None  # keeps the pointer alive
    return _enc(_L().fpx_wire_encode_proxy_leader_phase2a, slot, round_, keep.ctypes.data if keep is not None else None, n, noop)


def encode_acceptor_phase2a(slot, round_, value):
    p, n, noop = _val(value)
    keep = np.frombuffer(bytes(value), dtype=np.uint8) if value else None
    return _enc(_L().fpx_wire_encode_acceptor_phase2a, slot, round_, keep.ctypes.data if keep is not None else None, n, noop)


def encode_acceptor_phase1a(round_, chosen_watermark):
    return _enc(_L().fpx_wire_encode_acceptor_phase1a, round_, chosen_watermark)


def encode_proxy_leader_phase2b(group_index, acceptor_index, slot, round_):
    return _enc(_L().fpx_wire_encode_proxy_leader_phase2b, group_index, acceptor_index, slot, round_)


def encode_replica_chosen(slot, value):
    p, n, noop = _val(value)
    keep = np.frombuffer(bytes(value), dtype=np.uint8) if value else None
    return _enc(_L().fpx_wire_encode_replica_chosen, slot, keep.ctypes.data if keep is not None else None, n, noop)


def encode_leader_nack(round_):
    return _enc(_L().fpx_wire_encode_leader_nack, round_)


def encode_phase2b_batch(slot, round_, vote_bits, group_of_slot=None, grid_cols=0):
    """the Phase2b replies of a K1 batch as a list of ProxyLeaderInbound byte strings"""
    slot = np.ascontiguousarray(slot, np.int32)
    round_ = np.ascontiguousarray(round_, np.int32)
    vote_bits = np.ascontiguousarray(vote_bits, np.uint64)
    gos = None if group_of_slot is None else np.ascontiguousarray(group_of_slot, np.int32)
    max_msgs = int(sum(bin(int(x)).count("1") for x in vote_bits.reshape(-1)))
    out = np.zeros(max(1, max_msgs * 48), np.uint8)
    off = np.zeros(max_msgs + 1, np.int64)
    k = _L().fpx_wire_encode_phase2b_batch(len(slot), slot.ctypes.data, round_.ctypes.data, vote_bits.ctypes.data,
                                           None if gos is None else gos.ctypes.data, grid_cols, out.ctypes.data,
                                           len(out), off.ctypes.data, max_msgs)
    if k < 0:
        raise ValueError("output too small")
    return [out[off[i]:off[i + 1]].tobytes() for i in range(k)]
