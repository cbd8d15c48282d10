"""The wire adapter's DEVICE decoders (include/fpx_wire.h, fpx_wire_decode_*_dev): a tick of serialised
ProxyLeaderInbound / AcceptorInbound messages in HBM -> the SoA batch, one thread per message.  Checked against
the google.protobuf vectors of tests/golden/wire_vectors.json, against the host decoders field by field on random
ticks, on every malformed case of tests/test_wire.py, and end to end: bytes -> decode -> fused step == the oracle.

Run on the MI355X box: python -m pytest tests -m gpu
"""
import json
import os

import numpy as np
import pytest

from tests import workloads as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PLI = ["kind", "slot", "round", "is_noop", "value_off", "value_len", "group_index", "acceptor_index"]
ACC = ["kind", "slot", "round", "is_noop", "value_off", "value_len", "chosen_watermark"]


@pytest.fixture(scope="module")
def fa():
    import frankenpaxos_amd

    frankenpaxos_amd.lib()  # raises if libfpx.so is missing: no fallback
    return frankenpaxos_amd


@pytest.fixture(scope="module")
def wire():
    from frankenpaxos_amd import wire as w

    return w


@pytest.fixture(scope="module")
def gpu(fa):
    return fa.Context(fa.make_config(num_slots=1024, num_replicas=3, f=1))


@pytest.fixture(scope="module")
def vectors():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "wire_vectors.json")))["vectors"]


def dev_decode(gpu, wire, which, msgs, offsets=None, base=0):
    import torch

    buf, off = wire.pack(msgs)
    if offsets is not None:
        off = np.ascontiguousarray(offsets, np.int64)
    dev = torch.device("cuda:0")
    d = gpu.wire_decode_dev(which, torch.from_numpy(buf).to(dev), torch.from_numpy(off).to(dev), base,
                            buf_len=sum(len(m) for m in msgs))
    st = gpu.sync()
    return st, {k: v.cpu().numpy() for k, v in d.items()}


def same(host, dev, names, base=0):
    for k in names:
        assert (host[k] == dev[k]).all(), k
    want_id = np.where(host["kind"] == 1, base + np.arange(len(host["kind"])), -1)
    assert (dev["value_id"] == want_id).all()


def test_golden_vectors_decode_on_the_device(gpu, wire, vectors):
    p2a = [v for v in vectors if v["msg"] == "phase2a"]
    p2b = [v for v in vectors if v["msg"] == "phase2b"]
    p1a = [v for v in vectors if v["msg"] == "phase1a"]
    msgs = [bytes.fromhex(v["proxy_leader_inbound"]) for v in p2a + p2b]
    st, d = dev_decode(gpu, wire, "proxy_leader_inbound", msgs, base=1000)
    assert st == 0
    same(wire.decode_proxy_leader_inbound(msgs), d, PLI, 1000)
    for i, v in enumerate(p2a):
        assert d["kind"][i] == wire.PHASE2A and d["slot"][i] == v["slot"] and d["round"][i] == v["round"]
        assert d["is_noop"][i] == (v["value"] == "noop")
        o, n = int(d["value_off"][i]), int(d["value_len"][i])
        assert b"".join(msgs)[o:o + n].hex() == v["value_hex"]
    for j, v in enumerate(p2b):
        i = len(p2a) + j
        assert d["kind"][i] == wire.PHASE2B and d["slot"][i] == v["slot"] and d["round"][i] == v["round"]
        assert d["group_index"][i] == v["group_index"] and d["acceptor_index"][i] == v["acceptor_index"]
    msgs = [bytes.fromhex(v["acceptor_inbound"]) for v in p2a + p1a]
    st, d = dev_decode(gpu, wire, "acceptor_inbound", msgs)
    assert st == 0
    same(wire.decode_acceptor_inbound(msgs), d, ACC)
    for j, v in enumerate(p1a):
        i = len(p2a) + j
        assert d["kind"][i] == wire.PHASE1A and d["round"][i] == v["round"] and d["chosen_watermark"][i] == v["chosen_watermark"]


def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _field(num, wt, payload):
    return _varint(num << 3 | wt) + (_varint(len(payload)) + payload if wt == 2 else payload)


def odd_messages(rng, n, acceptor):
    """valid encodings no encoder of ours writes: fields in any order, unknown fields of every wire type in between,
    two members of the oneof (the last wins), negative int32s as 10-byte varints, non-minimal varints"""
    unknown = lambda: [_field(9, 0, _varint(int(rng.integers(0, 1 << 40)))), _field(10, 2, bytes(rng.integers(0, 256, int(rng.integers(0, 9)), dtype=np.uint8))),
                       _field(11, 1, bytes(8)), _field(12, 5, bytes(4))][int(rng.integers(0, 4))]
    out = []
    for _ in range(n):
        def phase2a():
            val = _field(2, 2, b"") if rng.random() < 0.4 else _field(1, 2, _field(1, 2, bytes(rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8))))
            if rng.random() < 0.2:
                val = _field(2, 2, b"") + val  # both members of CommandBatchOrNoop: the last wins
            parts = [_field(1, 0, _varint(int(rng.integers(-5, 1 << 31)))), _field(2, 0, _varint(int(rng.integers(0, 1 << 20)))),
                     _field(3, 2, val)]
            return parts

        def ints(k):
            return [_field(f + 1, 0, _varint(int(rng.integers(0, 1 << 31)))) for f in range(k)]

        members = []
        for _m in range(int(rng.integers(1, 3))):
            which = int(rng.integers(0, 3))
            if acceptor:
                # (round 5: the read path's members too -- MaxSlotRequest{CommandId{address, pseudonym, id}}, BatchMaxSlotRequest{2 ints})
                which = int(rng.integers(0, 5))
                cid = [_field(1, 2, bytes(rng.integers(0, 256, int(rng.integers(0, 24)), dtype=np.uint8)))] + ints(3)[1:]
                cid = [cid[i] for i in rng.permutation(3)] + [unknown() for _u in range(int(rng.integers(0, 2)))]
                num, parts = [(1, ints(2)), (2, phase2a()), (9, []), (3, [_field(1, 2, b"".join(cid))]), (4, ints(2))][which]
            else:
                num, parts = [(1, phase2a()), (2, ints(4)), (7, ints(1))][which]
            parts = parts + [unknown() for _u in range(int(rng.integers(0, 3)))]
            parts = [parts[i] for i in rng.permutation(len(parts))]
            if parts and rng.random() < 0.1:
                parts.append(parts[0])  # a field twice: the last value wins
            members.append(_field(num, 2, b"".join(parts)))
        if rng.random() < 0.3:
            members.insert(int(rng.integers(0, len(members) + 1)), unknown())
        out.append(b"".join(members))
    return out


@pytest.mark.parametrize("which", ["proxy_leader_inbound", "acceptor_inbound"])
def test_random_ticks_match_the_host_decoder(gpu, wire, which):
    rng = np.random.default_rng(5)
    msgs = odd_messages(rng, 20000, which == "acceptor_inbound") + [b""]
    host = (wire.decode_proxy_leader_inbound if which == "proxy_leader_inbound" else wire.decode_acceptor_inbound)(msgs)
    assert host["status"] == 0
    # Phase2a, the other member(s), OTHER: all present (AcceptorInbound: Phase1a, MaxSlotRequest, BatchMaxSlotRequest)
    assert len(set(host["kind"].tolist())) == (5 if which == "acceptor_inbound" else 3)
    st, d = dev_decode(gpu, wire, which, msgs, base=7)
    assert st == 0
    same(host, d, PLI if which == "proxy_leader_inbound" else ACC, 7)


def test_malformed_messages_and_offsets(fa, gpu, wire):
    good = bytes.fromhex("0a08080510011a021200")
    for bad in (good[:-1], bytes.fromhex("0a0608051a021200"), bytes.fromhex("0a06080510011a00"),
                bytes.fromhex("0a0308" + "ff" * 11), bytes.fromhex("12050800100218")):
        msgs = [good] * 300 + [bad] + [good] * 300 + [bad] + [good] * 50
        st, _ = dev_decode(gpu, wire, "proxy_leader_inbound", msgs)
        assert st == fa.FPX_EINVAL and gpu.error_detail()[0] == 300, bad.hex()
        assert wire.decode_proxy_leader_inbound(msgs)["bad_index"] == 300
    # offsets are judged before messages, like the host decoder: [.., 10^9, ..] outranks the malformed message 1
    msgs = [good, good[:-1], good, good, good]
    off = np.array([0, 10, 19, 10 ** 9, 39, 49], np.int64)
    host = wire.decode_proxy_leader_inbound(msgs, off)
    st, _ = dev_decode(gpu, wire, "proxy_leader_inbound", msgs, off)
    assert host["status"] == 1 and st == fa.FPX_EINVAL and gpu.error_detail()[0] == host["bad_index"] == 3
    for off in ([0, 10, 9, 29, 39, 49], [-1, 10, 19, 29, 39, 49], [0, 10, 19, 29, 39, 50]):
        host = wire.decode_proxy_leader_inbound(msgs, np.array(off, np.int64))
        st, _ = dev_decode(gpu, wire, "proxy_leader_inbound", msgs, np.array(off, np.int64))
        assert host["status"] == 1 and st == fa.FPX_EINVAL and gpu.error_detail()[0] == host["bad_index"], off
    st, d = dev_decode(gpu, wire, "proxy_leader_inbound", [good] * 3)  # and the context is usable again
    assert st == 0 and d["slot"].tolist() == [5, 5, 5]


def test_a_half_decoded_tick_never_reaches_the_acceptors(fa, wire):
    import torch

    S = 512
    gpu = fa.Context(fa.make_config(num_slots=S, num_replicas=3, f=1))
    dev = torch.device("cuda:0")
    msgs = [wire.encode_proxy_leader_phase2a(s, 0, None) for s in range(S)]
    msgs[77] = msgs[77][:-1]
    buf, off = wire.pack(msgs)
    before = gpu.read_state()
    d = gpu.wire_decode_dev("proxy_leader_inbound", torch.from_numpy(buf).to(dev), torch.from_numpy(off).to(dev),
                            buf_len=int(off[-1]))
    ch = torch.zeros(S, dtype=torch.uint8, device=dev)
    gpu.phase2_fused_dev(d["slot"], d["round"], d["value_id"], chosen=ch)
    assert gpu.sync() == fa.FPX_EINVAL and gpu.error_detail()[0] == 77
    assert not bool(ch.any())
    for a, b in zip(before, gpu.read_state()):
        assert (a == b).all()


@pytest.mark.parametrize("R,f", [(3, 1), (256, 127)])
def test_bytes_to_chosen_matches_the_oracle(fa, oracle, wire, R, f):
    """the whole inbound leg on the device: serialised Phase2a's -> k_wire_decode -> the fused step; same Chosen's
    and the same acceptor state as the oracle fed the decoded fields by the host decoder"""
    import torch

    S = 1 << 14
    kw = dict(num_slots=S, num_replicas=R, f=f)
    gpu, ref = fa.Context(fa.make_config(**kw)), oracle.System(oracle.make_config(**kw))
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(9)
    cmds = [None if rng.random() < 0.2 else bytes(rng.integers(0, 256, int(rng.integers(1, 24)), dtype=np.uint8)) for _ in range(S)]
    slots = rng.permutation(S).astype(np.int32)
    # CommandBatchOrNoop{command_batch = 1 {command = 1: <bytes>}} by hand; None = noop
    msgs = [wire.encode_proxy_leader_phase2a(int(s), 0, None if c is None else _field(1, 2, _field(1, 2, c)))
            for s, c in zip(slots, cmds)]
    buf, off = wire.pack(msgs)
    host = wire.decode_proxy_leader_inbound(msgs)
    d = gpu.wire_decode_dev("proxy_leader_inbound", torch.from_numpy(buf).to(dev), torch.from_numpy(off).to(dev),
                            value_id_base=100, buf_len=int(off[-1]))
    ch = torch.zeros(S, dtype=torch.uint8, device=dev)
    cr = torch.zeros(S, dtype=torch.int32, device=dev)
    cv = torch.zeros(S, dtype=torch.int32, device=dev)
    gpu.phase2_fused_dev(d["slot"], d["round"], d["value_id"], None, ch, cr, cv)
    assert gpu.sync() == 0
    val = (100 + np.arange(S)).astype(np.int32)
    out = ref.phase2_fused(host["slot"], host["round"], val)
    assert out[0] == 0
    assert (ch.cpu().numpy() == out[1]).all() and (cr.cpu().numpy() == out[2]).all() and (cv.cpu().numpy() == out[3]).all()
    assert bool(ch.all())
    W.assert_same_state(gpu, ref, range(0, S, 131))
