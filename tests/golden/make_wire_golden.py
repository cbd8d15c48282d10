"""Generates tests/golden/wire_vectors.json: the byte strings an INDEPENDENT protobuf runtime (google.protobuf, the
reference C++/Python implementation of the wire format -- ScalaPB's toByteArray emits the same canonical bytes)
produces for the Phase-2 messages of the reference, from descriptors transcribed field by field from
/root/reference/shared/src/main/scala/frankenpaxos/multipaxos/MultiPaxos.proto (Noop :183-186, CommandId
:188-196, Command :198-204, CommandBatch :206-211, CommandBatchOrNoop :213-221, Phase1a :238-253, Phase1bSlotInfo :254-261, Phase1b :263-271, Phase2a
:273-281, Phase2b :283-291, Chosen :293-299, Nack :455-460, LeaderInbound.phase1b = 1 :530, LeaderInbound.nack = 6 :535, ProxyLeaderInbound
:541-549, AcceptorInbound :551-561, ReplicaInbound :563-575).  Round 5: the acceptor's read path -- MaxSlotRequest :316-321,
MaxSlotReply :323-331, BatchMaxSlotRequest :333-339, BatchMaxSlotReply :341-349, ClientInbound.max_slot_reply = 4 :497,
ReadBatcherInbound.batch_max_slot_reply = 4 :521, AcceptorInbound.max_slot_request = 3 / batch_max_slot_request = 4 :558-559.

Round 3: the same for mencius/Mencius.proto (Phase1a :104-117, Phase2a :151-158, Phase2aNoopRange :160-167, Phase2b
:169-176, Phase2bNoopRange :178-187, Chosen :189-195, ChosenNoopRange :197-203, Nack :266-271, LeaderInbound.nack = 7
:322-337, ProxyLeaderInbound :339-350, AcceptorInbound :352-361, ReplicaInbound :363-371) and epaxos/EPaxos.proto
(Instance :35-44, Ballot :46-53, Noop :55-59, Command :61-78, CommandOrNoop :80-89, CommandStatus :91-95,
InstancePrefixSetProto :97-104, PreAccept :113-123, PreAcceptOk :125-135, Accept :137-147, AcceptOk :149-156, Commit
:158-166, Prepare :178-184, PrepareOk :186-209, Nack :211-218, ReplicaInbound :220-235) with
compact/IntPrefixSet.proto:12-15.

Run where google.protobuf is importable:  python tests/golden/make_wire_golden.py
The committed JSON is what tests/test_wire.py checks the C codec against (it needs no protobuf runtime)."""
import json
import os

from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

F = descriptor_pb2.FieldDescriptorProto


def build():
    fd = descriptor_pb2.FileDescriptorProto(name="fpx_multipaxos_subset.proto", package="frankenpaxos.multipaxos",
                                            syntax="proto2")

    def msg(name, *fields):
        m = fd.message_type.add(name=name)
        for fname, number, ftype, label, tname, oneof in fields:
            f = m.field.add(name=fname, number=number, type=ftype, label=label)
            if tname:
                f.type_name = ".frankenpaxos.multipaxos." + tname
            if oneof is not None:
                if not m.oneof_decl:
                    m.oneof_decl.add(name=oneof)
                f.oneof_index = 0
        return m

    REQ, OPT, REP = F.LABEL_REQUIRED, F.LABEL_OPTIONAL, F.LABEL_REPEATED
    I32, BYT, MSG = F.TYPE_INT32, F.TYPE_BYTES, F.TYPE_MESSAGE
    msg("Noop")
    msg("CommandId", ("client_address", 1, BYT, REQ, None, None), ("client_pseudonym", 2, I32, REQ, None, None),
        ("client_id", 3, I32, REQ, None, None))
    msg("Command", ("command_id", 1, MSG, REQ, "CommandId", None), ("command", 2, BYT, REQ, None, None))
    msg("CommandBatch", ("command", 1, MSG, REP, "Command", None))
    msg("CommandBatchOrNoop", ("command_batch", 1, MSG, OPT, "CommandBatch", "value"),
        ("noop", 2, MSG, OPT, "Noop", "value"))
    msg("Phase1a", ("round", 1, I32, REQ, None, None), ("chosen_watermark", 2, I32, REQ, None, None))
    msg("Phase1bSlotInfo", ("slot", 1, I32, REQ, None, None), ("vote_round", 2, I32, REQ, None, None),
        ("vote_value", 3, MSG, REQ, "CommandBatchOrNoop", None))
    msg("Phase1b", ("group_index", 1, I32, REQ, None, None), ("acceptor_index", 2, I32, REQ, None, None),
        ("round", 3, I32, REQ, None, None), ("info", 4, MSG, REP, "Phase1bSlotInfo", None))
    msg("Phase2a", ("slot", 1, I32, REQ, None, None), ("round", 2, I32, REQ, None, None),
        ("command_batch_or_noop", 3, MSG, REQ, "CommandBatchOrNoop", None))
    msg("Phase2b", ("group_index", 1, I32, REQ, None, None), ("acceptor_index", 2, I32, REQ, None, None),
        ("slot", 3, I32, REQ, None, None), ("round", 4, I32, REQ, None, None))
    msg("Chosen", ("slot", 1, I32, REQ, None, None), ("command_batch_or_noop", 2, MSG, REQ, "CommandBatchOrNoop", None))
    msg("Nack", ("round", 1, I32, REQ, None, None))
    msg("ProxyLeaderInbound", ("phase2a", 1, MSG, OPT, "Phase2a", "request"), ("phase2b", 2, MSG, OPT, "Phase2b", "request"))
    msg("MaxSlotRequest", ("command_id", 1, MSG, REQ, "CommandId", None))
    msg("MaxSlotReply", ("command_id", 1, MSG, REQ, "CommandId", None), ("group_index", 2, I32, REQ, None, None),
        ("acceptor_index", 3, I32, REQ, None, None), ("slot", 4, I32, REQ, None, None))
    msg("BatchMaxSlotRequest", ("read_batcher_index", 1, I32, REQ, None, None), ("read_batcher_id", 2, I32, REQ, None, None))
    msg("BatchMaxSlotReply", ("read_batcher_index", 1, I32, REQ, None, None), ("read_batcher_id", 2, I32, REQ, None, None),
        ("acceptor_index", 3, I32, REQ, None, None), ("slot", 4, I32, REQ, None, None))
    msg("ClientInbound", ("max_slot_reply", 4, MSG, OPT, "MaxSlotReply", "request"))
    msg("ReadBatcherInbound", ("batch_max_slot_reply", 4, MSG, OPT, "BatchMaxSlotReply", "request"))
    msg("AcceptorInbound", ("phase1a", 1, MSG, OPT, "Phase1a", "request"), ("phase2a", 2, MSG, OPT, "Phase2a", "request"),
        ("max_slot_request", 3, MSG, OPT, "MaxSlotRequest", "request"),
        ("batch_max_slot_request", 4, MSG, OPT, "BatchMaxSlotRequest", "request"))
    msg("ReplicaInbound", ("chosen", 1, MSG, OPT, "Chosen", "request"))
    msg("LeaderInbound", ("phase1b", 1, MSG, OPT, "Phase1b", "request"), ("nack", 6, MSG, OPT, "Nack", "request"))
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("frankenpaxos.multipaxos." + n))
    return {n: get(n) for n in ("Noop", "CommandId", "Command", "CommandBatch", "CommandBatchOrNoop", "Phase1a", "Phase1bSlotInfo", "Phase1b", "Phase2a",
                                "Phase2b", "Chosen", "Nack", "ProxyLeaderInbound", "AcceptorInbound", "ReplicaInbound",
                                "LeaderInbound", "MaxSlotRequest", "MaxSlotReply", "BatchMaxSlotRequest", "BatchMaxSlotReply",
                                "ClientInbound", "ReadBatcherInbound")}


def _file(name, package, messages, enums=()):
    """messages: {name: [(field, number, type, label, type_name or None, oneof or None)]}"""
    fd = descriptor_pb2.FileDescriptorProto(name=name, package=package, syntax="proto2")
    for ename, values in enums:
        e = fd.enum_type.add(name=ename)
        for vname, num in values:
            e.value.add(name=vname, number=num)
    for mname, fields in messages.items():
        m = fd.message_type.add(name=mname)
        for fname, number, ftype, label, tname, oneof in fields:
            f = m.field.add(name=fname, number=number, type=ftype, label=label)
            if tname:
                f.type_name = "." + package + "." + tname
            if oneof is not None:
                if not m.oneof_decl:
                    m.oneof_decl.add(name=oneof)
                f.oneof_index = 0
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName(package + "." + n)) for n in messages}


def build_mencius():
    REQ, OPT, REP = F.LABEL_REQUIRED, F.LABEL_OPTIONAL, F.LABEL_REPEATED
    I32, BYT, MSG = F.TYPE_INT32, F.TYPE_BYTES, F.TYPE_MESSAGE
    i = lambda n, k: (n, k, I32, REQ, None, None)
    return _file("fpx_mencius_subset.proto", "frankenpaxos.mencius", {
        "Noop": [],
        "CommandId": [("client_address", 1, BYT, REQ, None, None), i("client_pseudonym", 2), i("client_id", 3)],
        "Command": [("command_id", 1, MSG, REQ, "CommandId", None), ("command", 2, BYT, REQ, None, None)],
        "CommandBatch": [("command", 1, MSG, REP, "Command", None)],
        "CommandBatchOrNoop": [("command_batch", 1, MSG, OPT, "CommandBatch", "value"), ("noop", 2, MSG, OPT, "Noop", "value")],
        "Phase1a": [i("round", 1), i("chosen_watermark", 2)],
        "HighWatermark": [i("nextSlot", 1)],
        "Phase2a": [i("slot", 1), i("round", 2), ("command_batch_or_noop", 3, MSG, REQ, "CommandBatchOrNoop", None)],
        "Phase2aNoopRange": [i("slot_start_inclusive", 1), i("slot_end_exclusive", 2), i("round", 3)],
        "Phase2b": [i("acceptor_index", 1), i("slot", 2), i("round", 3)],
        "Phase2bNoopRange": [i("acceptor_group_index", 1), i("acceptor_index", 2), i("slot_start_inclusive", 3),
                             i("slot_end_exclusive", 4), i("round", 5)],
        "Chosen": [i("slot", 1), ("command_batch_or_noop", 2, MSG, REQ, "CommandBatchOrNoop", None)],
        "ChosenNoopRange": [i("slot_start_inclusive", 1), i("slot_end_exclusive", 2)],
        "Nack": [i("round", 1)],
        "LeaderInbound": [("nack", 7, MSG, OPT, "Nack", "request")],
        "ProxyLeaderInbound": [("high_watermark", 1, MSG, OPT, "HighWatermark", "request"),
                               ("phase2a", 2, MSG, OPT, "Phase2a", "request"),
                               ("phase2a_noop_range", 3, MSG, OPT, "Phase2aNoopRange", "request"),
                               ("phase2b", 4, MSG, OPT, "Phase2b", "request"),
                               ("phase2b_noop_range", 5, MSG, OPT, "Phase2bNoopRange", "request")],
        "AcceptorInbound": [("phase1a", 1, MSG, OPT, "Phase1a", "request"), ("phase2a", 2, MSG, OPT, "Phase2a", "request"),
                            ("phase2a_noop_range", 3, MSG, OPT, "Phase2aNoopRange", "request")],
        "ReplicaInbound": [("chosen", 1, MSG, OPT, "Chosen", "request"),
                           ("chosen_noop_range", 2, MSG, OPT, "ChosenNoopRange", "request")],
    })


def build_epaxos():
    REQ, OPT, REP = F.LABEL_REQUIRED, F.LABEL_OPTIONAL, F.LABEL_REPEATED
    I32, BYT, MSG, ENUM = F.TYPE_INT32, F.TYPE_BYTES, F.TYPE_MESSAGE, F.TYPE_ENUM
    i = lambda n, k, lab=REQ: (n, k, I32, lab, None, None)
    m = lambda n, k, t, lab=REQ, one=None: (n, k, MSG, lab, t, one)
    # compact.IntPrefixSetProto lives in its own package in the reference; the bytes do not depend on package names
    return _file("fpx_epaxos_subset.proto", "frankenpaxos.epaxos", {
        "IntPrefixSetProto": [i("watermark", 1), i("value", 2, REP)],
        "Instance": [i("replica_index", 1), i("instance_number", 2)],
        "Ballot": [i("ordering", 1), i("replica_index", 2)],
        "Noop": [],
        "Command": [("client_address", 1, BYT, REQ, None, None), i("client_pseudonym", 2), i("client_id", 3),
                    ("command", 4, BYT, REQ, None, None)],
        "CommandOrNoop": [m("command", 1, "Command", OPT, "value"), m("noop", 2, "Noop", OPT, "value")],
        "InstancePrefixSetProto": [i("numReplicas", 1), m("int_prefix_set", 2, "IntPrefixSetProto", REP)],
        "ClientRequest": [m("command", 1, "Command")],
        "PreAccept": [m("instance", 1, "Instance"), m("ballot", 2, "Ballot"), m("command_or_noop", 3, "CommandOrNoop"),
                      i("sequence_number", 4), m("dependencies", 5, "InstancePrefixSetProto")],
        "PreAcceptOk": [m("instance", 1, "Instance"), m("ballot", 2, "Ballot"), i("replica_index", 3),
                        i("sequence_number", 4), m("dependencies", 5, "InstancePrefixSetProto")],
        "Accept": [m("instance", 1, "Instance"), m("ballot", 2, "Ballot"), m("command_or_noop", 3, "CommandOrNoop"),
                   i("sequence_number", 4), m("dependencies", 5, "InstancePrefixSetProto")],
        "AcceptOk": [m("instance", 2, "Instance"), m("ballot", 3, "Ballot"), i("replica_index", 7)],
        "Commit": [m("instance", 1, "Instance"), m("command_or_noop", 2, "CommandOrNoop"), i("sequence_number", 3),
                   m("dependencies", 4, "InstancePrefixSetProto")],
        "Prepare": [m("instance", 1, "Instance"), m("ballot", 2, "Ballot")],
        "PrepareOk": [m("ballot", 1, "Ballot"), m("instance", 2, "Instance"), i("replica_index", 3),
                      m("vote_ballot", 4, "Ballot"), ("status", 5, ENUM, REQ, "CommandStatus", None),
                      m("command_or_noop", 6, "CommandOrNoop", OPT), i("sequence_number", 7, OPT),
                      m("dependencies", 8, "InstancePrefixSetProto", OPT)],
        "Nack": [m("instance", 1, "Instance"), m("largest_ballot", 2, "Ballot")],
        "ReplicaInbound": [m("client_request", 1, "ClientRequest", OPT, "request"), m("pre_accept", 2, "PreAccept", OPT, "request"),
                           m("pre_accept_ok", 3, "PreAcceptOk", OPT, "request"), m("accept", 4, "Accept", OPT, "request"),
                           m("accept_ok", 5, "AcceptOk", OPT, "request"), m("commit", 6, "Commit", OPT, "request"),
                           m("prepare", 7, "Prepare", OPT, "request"), m("prepare_ok", 8, "PrepareOk", OPT, "request"),
                           m("nack", 9, "Nack", OPT, "request")],
    }, enums=[("CommandStatus", [("NotSeen", 0), ("PreAccepted", 1), ("Accepted", 2)])])


def mencius_vectors():
    M = build_mencius()
    out = []

    def val(commands):
        v = M["CommandBatchOrNoop"]()
        if commands is None:
            v.noop.SetInParent()
        else:
            v.command_batch.SetInParent()
            for addr, pseud, cid, payload in commands:
                c = v.command_batch.command.add()
                c.command_id.client_address, c.command_id.client_pseudonym, c.command_id.client_id = addr, pseud, cid
                c.command = payload
        return v

    vals = {"noop": None, "one": [(b"\x0a\x00\x00\x01:9000", 3, 17, b"set x 1")],
            "batch": [(b"c%d" % i, i, 1000 * i, bytes(range(i, i + 40))) for i in range(5)]}
    for vname, commands in vals.items():
        for slot, rnd in [(0, 0), (5, 1), (1 << 20, 300), (2147483647, 2147483646)]:
            p, a, r = M["ProxyLeaderInbound"](), M["AcceptorInbound"](), M["ReplicaInbound"]()
            p.phase2a.slot, p.phase2a.round = slot, rnd
            p.phase2a.command_batch_or_noop.CopyFrom(val(commands))
            a.phase2a.CopyFrom(p.phase2a)
            r.chosen.slot = slot
            r.chosen.command_batch_or_noop.CopyFrom(val(commands))
            out.append({"msg": "mencius_phase2a", "slot": slot, "round": rnd, "value": vname,
                        "value_hex": val(commands).SerializeToString().hex(),
                        "proxy_leader_inbound": p.SerializeToString().hex(), "acceptor_inbound": a.SerializeToString().hex(),
                        "replica_inbound_chosen": r.SerializeToString().hex()})
    for g, a_, start, end, rnd in [(0, 0, 0, 1, 0), (1, 2, 256, 512, 3), (15, 1, 5, 1 << 22, 300), (0, 2, 2147483000, 2147483647, -1)]:
        p1, p2, p3 = M["ProxyLeaderInbound"](), M["ProxyLeaderInbound"](), M["ProxyLeaderInbound"]()
        p1.phase2a_noop_range.slot_start_inclusive, p1.phase2a_noop_range.slot_end_exclusive, p1.phase2a_noop_range.round = start, end, rnd
        q = p2.phase2b_noop_range
        q.acceptor_group_index, q.acceptor_index, q.slot_start_inclusive, q.slot_end_exclusive, q.round = g, a_, start, end, rnd
        p3.phase2b.acceptor_index, p3.phase2b.slot, p3.phase2b.round = a_, start, rnd
        a = M["AcceptorInbound"]()
        a.phase2a_noop_range.CopyFrom(p1.phase2a_noop_range)
        a1 = M["AcceptorInbound"]()
        a1.phase1a.round, a1.phase1a.chosen_watermark = rnd, start
        r = M["ReplicaInbound"]()
        r.chosen_noop_range.slot_start_inclusive, r.chosen_noop_range.slot_end_exclusive = start, end
        l = M["LeaderInbound"]()
        l.nack.round = rnd
        h = M["ProxyLeaderInbound"]()
        h.high_watermark.nextSlot = end
        out.append({"msg": "mencius_ranges", "group": g, "acceptor": a_, "start": start, "end": end, "round": rnd,
                    "pl_phase2a_noop_range": p1.SerializeToString().hex(), "pl_phase2b_noop_range": p2.SerializeToString().hex(),
                    "pl_phase2b": p3.SerializeToString().hex(), "acc_phase2a_noop_range": a.SerializeToString().hex(),
                    "acc_phase1a": a1.SerializeToString().hex(), "rep_chosen_noop_range": r.SerializeToString().hex(),
                    "leader_nack": l.SerializeToString().hex(), "pl_high_watermark": h.SerializeToString().hex()})
    return out


def epaxos_vectors():
    M = build_epaxos()
    out = []

    def cmd(c):
        v = M["CommandOrNoop"]()
        if c is None:
            v.noop.SetInParent()
        else:
            v.command.client_address, v.command.client_pseudonym, v.command.client_id, v.command.command = c
        return v

    def deps(d, n, wm, values):
        d.numReplicas = n
        for l in range(n):
            s = d.int_prefix_set.add()
            s.watermark = wm[l]
            s.value.extend([x for (ll, x) in values if ll == l])

    cases = [
        dict(instance=(0, 0), ballot=(0, 0), replica=1, seq=0, wm=[0, 0, 0], values=[], command=None),
        dict(instance=(2, 7), ballot=(3, 1), replica=4, seq=0, wm=[5, 0, 7, 9, 128], values=[(2, 8), (2, 9)],
             command=(b"\x0a\x00\x00\x02:7000", 1, 44, b"set k 1")),
        dict(instance=(6, 1 << 20), ballot=(1 << 24, 6), replica=0, seq=17, wm=[300, 1 << 20, 0, 1, 2, 3, 1 << 20],
             values=[(6, (1 << 20) + 1), (6, (1 << 20) + 2), (0, 305)], command=(b"c", 0, 2147483647, bytes(range(200)))),
    ]
    for c in cases:
        n = len(c["wm"])
        row = {"msg": "epaxos", **{k: (list(v) if isinstance(v, tuple) else v) for k, v in c.items() if k != "command"},
               "values": [list(v) for v in c["values"]], "is_noop": c["command"] is None,
               "command_hex": cmd(c["command"]).SerializeToString().hex()}
        for kind in ("pre_accept", "pre_accept_ok", "accept", "accept_ok", "commit", "prepare", "prepare_ok", "prepare_ok_bare", "nack"):
            r = M["ReplicaInbound"]()
            field = "prepare_ok" if kind == "prepare_ok_bare" else kind
            x = getattr(r, field)
            x.instance.replica_index, x.instance.instance_number = c["instance"]
            if kind == "nack":
                x.largest_ballot.ordering, x.largest_ballot.replica_index = c["ballot"]
            elif kind != "commit":
                x.ballot.ordering, x.ballot.replica_index = c["ballot"]
            if kind in ("pre_accept", "accept", "commit", "prepare_ok"):
                x.command_or_noop.CopyFrom(cmd(c["command"]))
            if kind in ("pre_accept", "pre_accept_ok", "accept", "commit", "prepare_ok"):
                x.sequence_number = c["seq"]
                deps(x.dependencies, n, c["wm"], c["values"])
            if kind in ("pre_accept_ok", "accept_ok", "prepare_ok", "prepare_ok_bare"):
                x.replica_index = c["replica"]
            if kind in ("prepare_ok", "prepare_ok_bare"):
                x.vote_ballot.ordering, x.vote_ballot.replica_index = (c["ballot"][0] // 2, c["ballot"][1])
                x.status = 0 if kind == "prepare_ok_bare" else 2
            row[kind] = r.SerializeToString().hex()
        cr = M["ReplicaInbound"]()
        cr.client_request.command.client_address, cr.client_request.command.client_pseudonym = b"a", 1
        cr.client_request.command.client_id, cr.client_request.command.command = 2, b"x"
        row["client_request"] = cr.SerializeToString().hex()
        out.append(row)
    return out


def value(M, commands):
    """CommandBatchOrNoop: None -> Noop, else a batch of (client_address, pseudonym, id, command bytes)"""
    v = M["CommandBatchOrNoop"]()
    if commands is None:
        v.noop.SetInParent()
    else:
        v.command_batch.SetInParent()
        for addr, pseud, cid, payload in commands:
            c = v.command_batch.command.add()
            c.command_id.client_address = addr
            c.command_id.client_pseudonym = pseud
            c.command_id.client_id = cid
            c.command = payload
    return v


def main():
    M = build()
    vals = {"noop": None,
            "one": [(b"\x0a\x00\x00\x01:9000", 3, 17, b"set x 1")],
            "batch": [(b"c%d" % i, i, 1000 * i, bytes(range(i, i + 40))) for i in range(5)],
            "empty_batch": []}
    ints = [0, 1, 5, 127, 128, 300, 65535, 1 << 20, 2147483647, -1]
    vectors = []
    for vname, commands in vals.items():
        vbytes = value(M, commands).SerializeToString()
        for slot, rnd in [(0, 0), (5, 1), (127, 128), (1 << 20, 300), (2147483647, 2147483646)]:
            p = M["ProxyLeaderInbound"]()
            p.phase2a.slot, p.phase2a.round = slot, rnd
            p.phase2a.command_batch_or_noop.CopyFrom(value(M, commands))
            a = M["AcceptorInbound"]()
            a.phase2a.CopyFrom(p.phase2a)
            r = M["ReplicaInbound"]()
            r.chosen.slot = slot
            r.chosen.command_batch_or_noop.CopyFrom(value(M, commands))
            vectors.append({"msg": "phase2a", "slot": slot, "round": rnd, "value": vname, "value_hex": vbytes.hex(),
                            "proxy_leader_inbound": p.SerializeToString().hex(),
                            "acceptor_inbound": a.SerializeToString().hex(),
                            "replica_inbound_chosen": r.SerializeToString().hex()})
    for g, a_, slot, rnd in [(0, 0, 0, 0), (0, 2, 5, 1), (15, 3, 1 << 20, 7), (1, 255, 2147483647, 300), (3, 1, 9, -1)]:
        p = M["ProxyLeaderInbound"]()
        p.phase2b.group_index, p.phase2b.acceptor_index, p.phase2b.slot, p.phase2b.round = g, a_, slot, rnd
        vectors.append({"msg": "phase2b", "group_index": g, "acceptor_index": a_, "slot": slot, "round": rnd,
                        "proxy_leader_inbound": p.SerializeToString().hex()})
    for x in ints:
        a = M["AcceptorInbound"]()
        a.phase1a.round, a.phase1a.chosen_watermark = x, (x // 2 if x > 0 else x)
        l = M["LeaderInbound"]()
        l.nack.round = x
        vectors.append({"msg": "phase1a_nack", "round": x, "chosen_watermark": a.phase1a.chosen_watermark,
                        "acceptor_inbound": a.SerializeToString().hex(), "leader_inbound_nack": l.SerializeToString().hex()})
    # Phase1b (MultiPaxos.proto:254-271, LeaderInbound.phase1b = 1 :530): what Acceptor.handlePhase1a answers
    for g, a_, rnd, info in [(0, 0, 0, []), (0, 2, 1, [(0, 0, "one")]), (3, 1, 300, [(5, 1, "noop"), (7, 0, "batch"), (1 << 20, 299, "one")]),
                             (15, 255, 2147483647, [(s_, s_ % 3, ("noop", "one", "empty_batch")[s_ % 3]) for s_ in range(40)])]:
        l = M["LeaderInbound"]()
        l.phase1b.group_index, l.phase1b.acceptor_index, l.phase1b.round = g, a_, rnd
        l.phase1b.SetInParent()
        for slot, vr, vname in info:
            e = l.phase1b.info.add()
            e.slot, e.vote_round = slot, vr
            e.vote_value.CopyFrom(value(M, vals[vname]))
        vectors.append({"msg": "phase1b", "group_index": g, "acceptor_index": a_, "round": rnd,
                        "info": [[slot, vr, vname, value(M, vals[vname]).SerializeToString().hex()] for slot, vr, vname in info],
                        "leader_inbound": l.SerializeToString().hex()})
    # the acceptor's read path (Acceptor.scala:222-254)
    for addr, pseud, cid, g, a_, slot in [(b"\x0a\x00\x00\x01:9000", 3, 17, 0, 0, -1), (b"", 0, 0, 0, 2, 0), (b"client-7", 2147483647, -5, 15, 255, 1 << 20),
                                          (bytes(range(200)), 128, 300, 1, 1, 2147483647)]:
        q = M["AcceptorInbound"]()
        q.max_slot_request.command_id.client_address = addr
        q.max_slot_request.command_id.client_pseudonym, q.max_slot_request.command_id.client_id = pseud, cid
        r = M["ClientInbound"]()
        r.max_slot_reply.command_id.CopyFrom(q.max_slot_request.command_id)
        r.max_slot_reply.group_index, r.max_slot_reply.acceptor_index, r.max_slot_reply.slot = g, a_, slot
        vectors.append({"msg": "max_slot", "command_id_hex": q.max_slot_request.command_id.SerializeToString().hex(),
                        "group_index": g, "acceptor_index": a_, "slot": slot,
                        "acceptor_inbound": q.SerializeToString().hex(), "client_inbound": r.SerializeToString().hex()})
    for rbi, rbid, a_, slot in [(0, 0, 0, -1), (1, 7, 2, 0), (300, 2147483647, 255, 1 << 20), (5, 128, 1, 2147483647)]:
        q = M["AcceptorInbound"]()
        q.batch_max_slot_request.read_batcher_index, q.batch_max_slot_request.read_batcher_id = rbi, rbid
        r = M["ReadBatcherInbound"]()
        b = r.batch_max_slot_reply
        b.read_batcher_index, b.read_batcher_id, b.acceptor_index, b.slot = rbi, rbid, a_, slot
        vectors.append({"msg": "batch_max_slot", "read_batcher_index": rbi, "read_batcher_id": rbid, "acceptor_index": a_, "slot": slot,
                        "acceptor_inbound": q.SerializeToString().hex(), "read_batcher_inbound": r.SerializeToString().hex()})
    vectors += mencius_vectors() + epaxos_vectors()
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wire_vectors.json")
    json.dump({"generator": "google.protobuf " + __import__("google.protobuf").protobuf.__version__,
               "vectors": vectors}, open(out, "w"), indent=0)
    print(len(vectors), "vectors ->", out)


if __name__ == "__main__":
    main()
