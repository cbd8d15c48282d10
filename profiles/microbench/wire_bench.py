"""Host throughput of the wire adapter (include/fpx_wire.h): protobuf bytes <-> SoA batches, one
core, no GPU.  A tick of m ProxyLeaderInbound{Phase2a} messages decoded into (slot, round, value reference) arrays;
the Phase2b replies of m slots x 3 acceptors encoded as ProxyLeaderInbound{Phase2b} byte strings and decoded again
into vote rows."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from frankenpaxos_amd import wire  # noqa: E402

L = wire._L()
m = int(os.environ.get("M", str(1 << 20)))


def best(fn, reps=3):
    """the first call pays the page faults of freshly allocated output arrays; report the best of `reps`"""
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    return r, min(ts), ts[0]


sample = [wire.encode_proxy_leader_phase2a(s, 3, b"\x0a\x10" + (b"%016d" % s)) for s in range(0, 4096)]  # CommandBatchOrNoop{command_batch}: 16 opaque bytes
msgs = sample * (m // len(sample))
buf, off = wire.pack(msgs)
n = len(msgs)
names = ["kind", "slot", "round", "is_noop", "value_off", "value_len", "group_index", "acceptor_index"]
out = {k: (np.zeros(n, np.int64) if k == "value_off" else np.zeros(n, np.int32)) for k in names}
bad = C.c_int32(-1)
st, dt, first = best(lambda: L.fpx_wire_decode_proxy_leader_inbound(buf.ctypes.data, len(buf), off.ctypes.data, n, *[out[k].ctypes.data for k in names], C.byref(bad)))
assert st == 0 and out["slot"][4097] == 1 and out["round"][5] == 3
print("decode ProxyLeaderInbound{Phase2a}: %d messages (%.1f MB) in %.1f ms = %.1fe6 messages/s, %.2f GB/s, 1 core (first call, fresh output pages: %.1f ms)"
      % (n, len(buf) / 1e6, dt * 1e3, n / dt / 1e6, len(buf) / dt / 1e9, first * 1e3))

slot = np.arange(n, dtype=np.int32)
rnd = np.full(n, 3, np.int32)
vb = np.zeros((n, 4), np.uint64)
vb[:, 0] = 7                      # three acceptors voted
max_msgs = 3 * n
obuf = np.zeros(max_msgs * 48, np.uint8)
ooff = np.zeros(max_msgs + 1, np.int64)
k, dt, first = best(lambda: L.fpx_wire_encode_phase2b_batch(n, slot.ctypes.data, rnd.ctypes.data, vb.ctypes.data, None, 0, obuf.ctypes.data, len(obuf),
                                                          ooff.ctypes.data, max_msgs))
assert k == max_msgs
print("encode Phase2b replies: %d messages (%.1f MB) in %.1f ms = %.1fe6 messages/s, 1 core (first call %.1f ms)"
      % (k, ooff[k] / 1e6, dt * 1e3, k / dt / 1e6, first * 1e3))
out = {kk: (np.zeros(k, np.int64) if kk == "value_off" else np.zeros(k, np.int32)) for kk in names}
st, dt, first = best(lambda: L.fpx_wire_decode_proxy_leader_inbound(obuf.ctypes.data, int(ooff[k]), ooff.ctypes.data, k, *[out[kk].ctypes.data for kk in names], C.byref(bad)))
assert st == 0
print("decode ProxyLeaderInbound{Phase2b}: %d messages in %.1f ms = %.1fe6 messages/s, 1 core (first call %.1f ms)" % (k, dt * 1e3, k / dt / 1e6, first * 1e3))
rs, rr, rb = np.zeros(k, np.int32), np.zeros(k, np.int32), np.zeros((k, 4), np.uint64)
nrows = C.c_int32()
st, dt, first = best(lambda: L.fpx_wire_phase2b_rows(k, out["kind"].ctypes.data, out["group_index"].ctypes.data, out["acceptor_index"].ctypes.data,
                                                    out["slot"].ctypes.data, out["round"].ctypes.data, 0, C.byref(nrows), rs.ctypes.data, rr.ctypes.data,
                                                    rb.ctypes.data))
assert st == 0 and nrows.value == n and int(rb[5, 0]) == 7
print("fold Phase2bs into vote rows: %d messages -> %d rows in %.1f ms = %.1fe6 messages/s, 1 core (first call %.1f ms)"
      % (k, nrows.value, dt * 1e3, k / dt / 1e6, first * 1e3))
