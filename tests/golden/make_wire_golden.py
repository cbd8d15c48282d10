"""Generates tests/golden/wire_vectors.json: the byte strings an INDEPENDENT protobuf runtime (google.protobuf, the
reference C++/Python implementation of the wire format -- ScalaPB's toByteArray emits the same canonical bytes)
produces for the Phase-2 messages of the reference, from descriptors transcribed field by field from
/root/reference/shared/src/main/scala/frankenpaxos/multipaxos/MultiPaxos.proto (Noop :183-186, CommandId
:188-196, Command :198-204, CommandBatch :206-211, CommandBatchOrNoop :213-221, Phase1a :238-253, Phase2a
:273-281, Phase2b :283-291, Chosen :293-299, Nack :455-460, LeaderInbound.nack = 6 :535, ProxyLeaderInbound
:541-549, AcceptorInbound :551-561, ReplicaInbound :563-575).

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
    msg("Phase2a", ("slot", 1, I32, REQ, None, None), ("round", 2, I32, REQ, None, None),
        ("command_batch_or_noop", 3, MSG, REQ, "CommandBatchOrNoop", None))
    msg("Phase2b", ("group_index", 1, I32, REQ, None, None), ("acceptor_index", 2, I32, REQ, None, None),
        ("slot", 3, I32, REQ, None, None), ("round", 4, I32, REQ, None, None))
    msg("Chosen", ("slot", 1, I32, REQ, None, None), ("command_batch_or_noop", 2, MSG, REQ, "CommandBatchOrNoop", None))
    msg("Nack", ("round", 1, I32, REQ, None, None))
    msg("ProxyLeaderInbound", ("phase2a", 1, MSG, OPT, "Phase2a", "request"), ("phase2b", 2, MSG, OPT, "Phase2b", "request"))
    msg("AcceptorInbound", ("phase1a", 1, MSG, OPT, "Phase1a", "request"), ("phase2a", 2, MSG, OPT, "Phase2a", "request"))
    msg("ReplicaInbound", ("chosen", 1, MSG, OPT, "Chosen", "request"))
    msg("LeaderInbound", ("nack", 6, MSG, OPT, "Nack", "request"))
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("frankenpaxos.multipaxos." + n))
    return {n: get(n) for n in ("Noop", "CommandId", "Command", "CommandBatch", "CommandBatchOrNoop", "Phase1a", "Phase2a",
                                "Phase2b", "Chosen", "Nack", "ProxyLeaderInbound", "AcceptorInbound", "ReplicaInbound",
                                "LeaderInbound")}


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
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wire_vectors.json")
    json.dump({"generator": "google.protobuf " + __import__("google.protobuf").protobuf.__version__,
               "vectors": vectors}, open(out, "w"), indent=0)
    print(len(vectors), "vectors ->", out)


if __name__ == "__main__":
    main()
