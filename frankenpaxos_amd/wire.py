"""Wire adapter (include/fpx_wire.h): the reference's protobuf messages of the Phase-2 path <-> SoA batches.
Thin ctypes binding of the C functions in libfpx.so; host code only (works without a GPU)."""
import ctypes as C

import numpy as np

from . import _lib

OTHER, PHASE2A, PHASE2B, PHASE1A, CHOSEN, NACK, PHASE2A_NOOP_RANGE, PHASE2B_NOOP_RANGE, CHOSEN_NOOP_RANGE = range(9)
PHASE1B = 9
MAX_SLOT_REQUEST, BATCH_MAX_SLOT_REQUEST = 10, 11      # the acceptor's read path (multipaxos/Acceptor.scala:222-254)
EPX_PRE_ACCEPT, EPX_PRE_ACCEPT_OK, EPX_ACCEPT, EPX_ACCEPT_OK, EPX_COMMIT, EPX_PREPARE, EPX_PREPARE_OK, EPX_NACK = range(16, 24)


class EpxMsg(C.Structure):
    """fpx_wire_epx_msg (include/fpx_wire.h)"""
    _fields_ = [("kind", C.c_int32), ("instance_leader", C.c_int32), ("instance_number", C.c_int32),
                ("ballot_ordering", C.c_int32), ("ballot_replica", C.c_int32), ("replica_index", C.c_int32),
                ("sequence_number", C.c_int32), ("has_sequence_number", C.c_int32),
                ("vote_ballot_ordering", C.c_int32), ("vote_ballot_replica", C.c_int32), ("status", C.c_int32),
                ("is_noop", C.c_int32), ("command", C.c_void_p), ("command_len", C.c_int32),
                ("num_replicas", C.c_int32), ("deps_watermark", C.c_void_p), ("num_values", C.c_int32),
                ("values_leader", C.c_void_p), ("values_id", C.c_void_p)]

_bound = False


def _L():
    global _bound
    L = _lib.lib()
    if not _bound:
        VP = C.c_void_p
        I32P = C.POINTER(C.c_int32)
        for name in ("fpx_wire_decode_proxy_leader_inbound",):
            getattr(L, name).argtypes = [VP, C.c_int64, VP, C.c_int32] + [VP] * 8 + [I32P]
        L.fpx_wire_decode_acceptor_inbound.argtypes = [VP, C.c_int64, VP, C.c_int32] + [VP] * 7 + [I32P]
        # the device decoders (Context.wire_decode_dev): ctx, d_buf, buf_len, d_offsets, n, outputs..., value_id_base, d_value_id
        L.fpx_wire_decode_proxy_leader_inbound_dev.argtypes = [VP, VP, C.c_int64, VP, C.c_int32] + [VP] * 8 + [C.c_int32, VP]
        L.fpx_wire_decode_acceptor_inbound_dev.argtypes = [VP, VP, C.c_int64, VP, C.c_int32] + [VP] * 7 + [C.c_int32, VP]
        L.fpx_wire_decode_replica_inbound.argtypes = [VP, C.c_int64, VP, C.c_int32] + [VP] * 5 + [I32P]
        L.fpx_wire_mencius_decode_proxy_leader_inbound.argtypes = [VP, C.c_int64, VP, C.c_int32] + [VP] * 9 + [I32P]
        L.fpx_wire_mencius_decode_acceptor_inbound.argtypes = [VP, C.c_int64, VP, C.c_int32] + [VP] * 8 + [I32P]
        L.fpx_wire_mencius_decode_replica_inbound.argtypes = [VP, C.c_int64, VP, C.c_int32] + [VP] * 6 + [I32P]
        i32 = C.c_int32
        for name, args in (("proxy_leader_phase2a", [i32, i32, VP, i32, i32]), ("acceptor_phase2a", [i32, i32, VP, i32, i32]),
                           ("proxy_leader_phase2a_noop_range", [i32] * 3), ("acceptor_phase2a_noop_range", [i32] * 3),
                           ("acceptor_phase1a", [i32] * 2), ("proxy_leader_phase2b", [i32] * 3),
                           ("proxy_leader_phase2b_noop_range", [i32] * 5), ("replica_chosen", [i32, VP, i32, i32]),
                           ("replica_chosen_noop_range", [i32] * 2), ("leader_nack", [i32])):
            fn = getattr(L, "fpx_wire_mencius_encode_" + name)
            fn.argtypes = [VP, C.c_int64] + args
            fn.restype = C.c_int64
        L.fpx_wire_epaxos_encode_replica_inbound.argtypes = [VP, C.c_int64, C.POINTER(EpxMsg)]
        L.fpx_wire_epaxos_encode_replica_inbound.restype = C.c_int64
        L.fpx_wire_epaxos_decode_replica_inbound.argtypes = [VP, C.c_int64, VP, C.c_int32, C.c_int32] + [VP] * 16 + \
            [C.c_int64, VP, VP, I32P]
        L.fpx_wire_phase2b_rows.argtypes = [C.c_int32, VP, VP, VP, VP, VP, C.c_int32, I32P, VP, VP, VP]
        L.fpx_wire_encode_proxy_leader_phase2a.argtypes = [VP, C.c_int64, C.c_int32, C.c_int32, VP, C.c_int32, C.c_int32]
        L.fpx_wire_encode_acceptor_phase2a.argtypes = [VP, C.c_int64, C.c_int32, C.c_int32, VP, C.c_int32, C.c_int32]
        L.fpx_wire_encode_acceptor_phase1a.argtypes = [VP, C.c_int64, C.c_int32, C.c_int32]
        L.fpx_wire_encode_proxy_leader_phase2b.argtypes = [VP, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.fpx_wire_encode_replica_chosen.argtypes = [VP, C.c_int64, C.c_int32, VP, C.c_int32, C.c_int32]
        L.fpx_wire_encode_leader_nack.argtypes = [VP, C.c_int64, C.c_int32]
        L.fpx_wire_encode_client_max_slot_reply.argtypes = [VP, C.c_int64, VP, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.fpx_wire_encode_client_max_slot_reply.restype = C.c_int64
        L.fpx_wire_encode_read_batcher_batch_max_slot_reply.argtypes = [VP, C.c_int64] + [C.c_int32] * 4
        L.fpx_wire_encode_read_batcher_batch_max_slot_reply.restype = C.c_int64
        L.fpx_wire_encode_leader_phase1b.argtypes = [VP, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32] + [VP] * 6
        L.fpx_wire_encode_leader_phase1b.restype = C.c_int64
        L.fpx_wire_decode_leader_inbound.argtypes = [VP, C.c_int64, VP, C.c_int32] + [VP] * 6 + [C.c_int32, I32P] + [VP] * 5 + [I32P]
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


def _decode(fn, messages, names, offsets=None):
    """offsets: override the message boundaries (tests of the bounds checks)"""
    buf, off = pack(messages)
    n = len(messages)
    if offsets is not None:
        off = np.ascontiguousarray(offsets, np.int64)
        n = len(off) - 1
    out = {k: (np.zeros(n, np.int64) if k == "value_off" else np.zeros(n, np.int32)) for k in names}
    bad = C.c_int32(-1)
    buf_len = sum(len(m) for m in messages)
    st = getattr(_L(), fn)(buf.ctypes.data, buf_len, off.ctypes.data, n, *[out[k].ctypes.data for k in names], C.byref(bad))
    out["status"], out["bad_index"], out["buf"] = st, bad.value, buf
    return out


def decode_proxy_leader_inbound(messages, offsets=None):
    return _decode("fpx_wire_decode_proxy_leader_inbound", messages,
                   ["kind", "slot", "round", "is_noop", "value_off", "value_len", "group_index", "acceptor_index"], offsets)


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
    """value: the serialised CommandBatchOrNoop / CommandOrNoop, or None for Noop -> (keep-alive array, pointer, length,
    is_noop).  An EMPTY byte string is neither: a oneof with no member set is the reference's
    logger.fatal("Empty CommandBatchOrNoop") (multipaxos/Replica.scala:414-416) and the decoder refuses it."""
    if value is None:
        return None, None, 0, 1
    v = np.frombuffer(bytes(value), dtype=np.uint8)
    if len(v) == 0:
        raise ValueError("an empty CommandBatchOrNoop (no oneof member) cannot be sent; pass None for Noop")
    return v, v.ctypes.data, len(v), 0


def encode_proxy_leader_phase2a(slot, round_, value):
    """value: the serialised CommandBatchOrNoop, or None for Noop"""
    keep, p, n, noop = _val(value)
    return _enc(_L().fpx_wire_encode_proxy_leader_phase2a, slot, round_, p, n, noop)


def encode_acceptor_phase2a(slot, round_, value):
    keep, p, n, noop = _val(value)
    return _enc(_L().fpx_wire_encode_acceptor_phase2a, slot, round_, p, n, noop)


def encode_acceptor_phase1a(round_, chosen_watermark):
    return _enc(_L().fpx_wire_encode_acceptor_phase1a, round_, chosen_watermark)


def encode_proxy_leader_phase2b(group_index, acceptor_index, slot, round_):
    return _enc(_L().fpx_wire_encode_proxy_leader_phase2b, group_index, acceptor_index, slot, round_)


def encode_replica_chosen(slot, value):
    keep, p, n, noop = _val(value)
    return _enc(_L().fpx_wire_encode_replica_chosen, slot, p, n, noop)


def encode_leader_nack(round_):
    return _enc(_L().fpx_wire_encode_leader_nack, round_)


def encode_client_max_slot_reply(command_id, group_index, acceptor_index, slot):
    """ClientInbound{MaxSlotReply}: command_id = the serialised CommandId of the request (decode_acceptor_inbound's
    value_off / value_len), slot = Acceptor.maxVotedSlot (multipaxos/Acceptor.scala:222-237)"""
    keep = np.frombuffer(bytes(command_id), np.uint8).copy() if len(command_id) else np.zeros(1, np.uint8)
    return _enc(_L().fpx_wire_encode_client_max_slot_reply, keep.ctypes.data, len(command_id), group_index, acceptor_index, slot)


def encode_read_batcher_batch_max_slot_reply(read_batcher_index, read_batcher_id, acceptor_index, slot):
    """ReadBatcherInbound{BatchMaxSlotReply} (multipaxos/Acceptor.scala:239-254)"""
    return _enc(_L().fpx_wire_encode_read_batcher_batch_max_slot_reply, read_batcher_index, read_batcher_id, acceptor_index, slot)


def encode_leader_phase1b(group_index, acceptor_index, round_, info):
    """info: [(slot, vote_round, value)] with value = the serialised CommandBatchOrNoop or None for Noop
    -> LeaderInbound{Phase1b} (what Acceptor.handlePhase1a sends, multipaxos/Acceptor.scala:163-181)"""
    n = len(info)
    slot = np.array([x[0] for x in info], np.int32)
    vr = np.array([x[1] for x in info], np.int32)
    blobs = [b"" if x[2] is None else bytes(x[2]) for x in info]
    if any(x[2] is not None and len(x[2]) == 0 for x in info):
        raise ValueError("an empty CommandBatchOrNoop (no oneof member) cannot be sent; pass None for Noop")
    noop = np.array([1 if x[2] is None else 0 for x in info], np.uint8)
    vbuf, voff = pack(blobs)
    vlen = np.diff(voff).astype(np.int32)
    return _enc(_L().fpx_wire_encode_leader_phase1b, group_index, acceptor_index, round_, n, slot.ctypes.data,
                vr.ctypes.data, vbuf.ctypes.data, voff.ctypes.data, vlen.ctypes.data, noop.ctypes.data)


def decode_leader_inbound(messages, info_cap=1 << 16):
    """-> kind (PHASE1B / NACK / OTHER), round, group_index, acceptor_index, info_first, info_count per message and the
    Phase1bSlotInfo entries (info_slot, info_vote_round, info_is_noop, info_value_off, info_value_len)"""
    buf, off = pack(messages)
    n = len(messages)
    per = {k: np.zeros(n, np.int32) for k in ("kind", "round", "group_index", "acceptor_index", "info_first", "info_count")}
    inf = {k: (np.zeros(info_cap, np.int64) if k == "info_value_off" else np.zeros(info_cap, np.int32))
           for k in ("info_slot", "info_vote_round", "info_is_noop", "info_value_off", "info_value_len")}
    bad, total = C.c_int32(-1), C.c_int32(0)
    st = _L().fpx_wire_decode_leader_inbound(buf.ctypes.data, sum(len(m) for m in messages), off.ctypes.data, n,
                                             *[per[k].ctypes.data for k in per], info_cap, C.byref(total),
                                             *[inf[k].ctypes.data for k in inf], C.byref(bad))
    out = dict(per)
    out.update({k: v[:total.value] for k, v in inf.items()})
    out["status"], out["bad_index"], out["buf"] = st, bad.value, buf
    return out


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


# ---- Mencius (mencius/Mencius.proto) --------------------------------------------------------------------------------
def mencius_decode_proxy_leader_inbound(messages, offsets=None):
    return _decode("fpx_wire_mencius_decode_proxy_leader_inbound", messages,
                   ["kind", "slot", "slot_end", "round", "is_noop", "value_off", "value_len", "group_index",
                    "acceptor_index"], offsets)


def mencius_decode_acceptor_inbound(messages, offsets=None):
    return _decode("fpx_wire_mencius_decode_acceptor_inbound", messages,
                   ["kind", "slot", "slot_end", "round", "is_noop", "value_off", "value_len", "chosen_watermark"], offsets)


def mencius_decode_replica_inbound(messages, offsets=None):
    return _decode("fpx_wire_mencius_decode_replica_inbound", messages,
                   ["kind", "slot", "slot_end", "is_noop", "value_off", "value_len"], offsets)


def mencius_encode(what, *args):
    """what: proxy_leader_phase2a(slot, round, value) | acceptor_phase2a(slot, round, value) |
    proxy_leader_phase2a_noop_range(start, end, round) | acceptor_phase2a_noop_range(start, end, round) |
    acceptor_phase1a(round, chosen_watermark) | proxy_leader_phase2b(acceptor_index, slot, round) |
    proxy_leader_phase2b_noop_range(group, acceptor, start, end, round) | replica_chosen(slot, value) |
    replica_chosen_noop_range(start, end) | leader_nack(round)"""
    fn = getattr(_L(), "fpx_wire_mencius_encode_" + what)
    if what in ("proxy_leader_phase2a", "acceptor_phase2a", "replica_chosen"):
        keep, p, n, noop = _val(args[-1])
        return _enc(fn, *args[:-1], p, n, noop)
    return _enc(fn, *args)


# ---- EPaxos (epaxos/EPaxos.proto) -----------------------------------------------------------------------------------
def epaxos_encode(kind, instance, ballot=(-1, -1), replica_index=-1, sequence_number=None, vote_ballot=(-1, -1), status=-1,
                  command=False, deps=None, values=()):
    """one ReplicaInbound.  command: False = the message carries none, None = Noop, bytes = the serialised
    CommandOrNoop; deps: None or the list of per-leader watermarks; values: explicit ids as (leader, id)"""
    m = EpxMsg()
    m.kind, (m.instance_leader, m.instance_number) = kind, instance
    m.ballot_ordering, m.ballot_replica = ballot
    m.vote_ballot_ordering, m.vote_ballot_replica = vote_ballot
    m.replica_index, m.status = replica_index, status
    m.has_sequence_number = 0 if sequence_number is None else 1
    m.sequence_number = -1 if sequence_number is None else sequence_number
    keep = None
    if command is False:
        m.is_noop = -1
    else:
        keep, m.command, m.command_len, m.is_noop = _val(command)
    wm = vl = vi = None
    if deps is None:
        m.num_replicas = -1
    else:
        wm = np.ascontiguousarray(deps, np.int32)
        vl = np.ascontiguousarray([v[0] for v in values], np.int32)
        vi = np.ascontiguousarray([v[1] for v in values], np.int32)
        m.num_replicas, m.deps_watermark, m.num_values = len(wm), wm.ctypes.data, len(vl)
        m.values_leader, m.values_id = vl.ctypes.data, vi.ctypes.data
    fn = _L().fpx_wire_epaxos_encode_replica_inbound
    out = np.zeros(64, np.uint8)
    n = fn(out.ctypes.data, len(out), C.byref(m))
    if n < -(1 << 40):
        raise ValueError("FPX_EINVAL: not an encodable ReplicaInbound")
    if n < 0:
        out = np.zeros(-n, np.uint8)
        n = fn(out.ctypes.data, len(out), C.byref(m))
    return out[:n].tobytes()


def epaxos_decode_replica_inbound(messages, max_replicas=7, values_cap=None, offsets=None):
    buf, off = pack(messages)
    n = len(messages)
    if offsets is not None:
        off = np.ascontiguousarray(offsets, np.int64)
        n = len(off) - 1
    names = ["kind", "instance_leader", "instance_number", "ballot_ordering", "ballot_replica", "replica_index",
             "sequence_number", "vote_ballot_ordering", "vote_ballot_replica", "status", "is_noop", "cmd_off", "cmd_len",
             "deps_num_replicas"]
    out = {k: (np.zeros(n, np.int64) if k == "cmd_off" else np.zeros(n, np.int32)) for k in names}
    out["deps_watermark"] = np.zeros((n, max_replicas), np.int32)
    out["values_off"] = np.zeros(n + 1, np.int64)
    bad = C.c_int32(-1)
    cap = 0 if values_cap is None else values_cap
    while True:
        vl, vi = np.zeros(max(cap, 1), np.int32), np.zeros(max(cap, 1), np.int32)
        st = _L().fpx_wire_epaxos_decode_replica_inbound(
            buf.ctypes.data, sum(len(m) for m in messages), off.ctypes.data, n, max_replicas,
            *[out[k].ctypes.data for k in names], out["deps_watermark"].ctypes.data, out["values_off"].ctypes.data, cap,
            vl.ctypes.data, vi.ctypes.data, C.byref(bad))
        if st == _lib.FPX_ECAPACITY and values_cap is None:
            cap = int(out["values_off"][n])
            continue
        break
    out["values_leader"], out["values_id"] = vl[:min(cap, int(out["values_off"][n]))], vi[:min(cap, int(out["values_off"][n]))]
    out["status_code"], out["bad_index"], out["buf"] = st, bad.value, buf
    return out
