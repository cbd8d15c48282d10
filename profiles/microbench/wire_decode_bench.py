"""How fast does a tick of serialised ProxyLeaderInbound{Phase2a} messages become a device batch?
(>= 5e8 Phase2a/s into a device batch.)

  host      fpx_wire_decode_proxy_leader_inbound, one host thread (what round 3 had)
  kernel    fpx_wire_decode_proxy_leader_inbound_dev, bytes already in HBM
  h2d+k     page-locked tick buffer + offsets -> hipMemcpyAsync -> the kernel  (the whole inbound leg)
  h2d+k+p2  ... -> fpx_phase2_fused_dev on the decoded batch (R = 3, f = 1): wire bytes to Chosen

    python profiles/microbench/wire_decode_bench.py [log2 n]
"""
import ctypes as C
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import frankenpaxos_amd as fa
from frankenpaxos_amd import wire


def tick(n, cmd_bytes=16, seed=1):
    """n ProxyLeaderInbound{phase2a{slot = a permutation of 0..n-1, round = 0, command_batch{command{16 random bytes}}}},
    canonical (minimal-varint) encoding, built with numpy"""
    rng = np.random.default_rng(seed)
    slot = rng.permutation(n).astype(np.int64)
    klen = 1 + (slot >= 1 << 7) + (slot >= 1 << 14) + (slot >= 1 << 21) + (slot >= 1 << 28)
    vl = cmd_bytes + 4
    body = 1 + klen + 2 + 2 + vl  # 08 slot | 10 00 | 1a VL | value
    assert vl < 128 and int(body.max()) < 128
    length = 2 + body
    off = np.zeros(n + 1, np.int64)
    np.cumsum(length, out=off[1:])
    buf = np.zeros(int(off[-1]), np.uint8)
    cmd = rng.integers(0, 256, (n, cmd_bytes), dtype=np.uint8)
    for kk in range(1, 6):
        idx = np.nonzero(klen == kk)[0]
        if not len(idx):
            continue
        m = np.zeros((len(idx), 2 + 1 + kk + 2 + 2 + vl), np.uint8)
        s = slot[idx]
        m[:, 0], m[:, 1], m[:, 2] = 0x0a, body[idx], 0x08
        for b in range(kk):
            m[:, 3 + b] = ((s >> (7 * b)) & 0x7f) | (0x80 if b < kk - 1 else 0)
        p = 3 + kk
        m[:, p], m[:, p + 1], m[:, p + 2], m[:, p + 3] = 0x10, 0, 0x1a, vl
        m[:, p + 4], m[:, p + 5], m[:, p + 6], m[:, p + 7] = 0x0a, cmd_bytes + 2, 0x0a, cmd_bytes
        m[:, p + 8:] = cmd[idx]
        buf[(off[idx][:, None] + np.arange(m.shape[1])[None, :]).ravel()] = m.ravel()
    return buf, off, slot.astype(np.int32)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = 1 << lg
    buf, off, slot = tick(n)
    # the tick is what the reference's encoder writes: spot-check against the library's own encoder
    for i in (0, 1, n // 2, n - 1):
        want = wire.encode_proxy_leader_phase2a(int(slot[i]), 0, bytes(buf[off[i + 1] - 20:off[i + 1]]))
        assert bytes(buf[off[i]:off[i + 1]]) == want, i
    dev = torch.device("cuda:0")
    gpu = fa.Context(fa.make_config(num_slots=n, num_replicas=3, f=1, flags=fa.FPX_F_TRUSTED))
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    res = {"n": n, "bytes": int(off[-1]), "bytes_per_message": float(off[-1]) / n}

    L = wire._L()
    outs = [np.zeros(n, np.int64 if k == 4 else np.int32) for k in range(8)]
    bad = C.c_int32(-1)
    t0 = time.perf_counter()
    st = L.fpx_wire_decode_proxy_leader_inbound(buf.ctypes.data, len(buf), off.ctypes.data, n, *[o.ctypes.data for o in outs], C.byref(bad))
    res["host_1thread_msgs_per_s"] = n / (time.perf_counter() - t0)
    assert st == 0 and (outs[1] == slot).all()

    d_buf, d_off = torch.from_numpy(buf).to(dev), torch.from_numpy(off).to(dev)
    d = gpu.wire_decode_dev("proxy_leader_inbound", d_buf, d_off)
    assert gpu.sync() == 0
    assert (d["slot"].cpu().numpy() == slot).all() and (d["value_off"].cpu().numpy() == outs[4]).all()
    t = timed(lambda: gpu.wire_decode_dev("proxy_leader_inbound", d_buf, d_off), 20)
    res["kernel_msgs_per_s"], res["kernel_ms"], res["kernel_GBs_in"] = n / t, t * 1e3, len(buf) / t / 1e9

    p_buf, p_off = torch.from_numpy(buf).pin_memory(), torch.from_numpy(off).pin_memory()

    def leg():
        d_buf.copy_(p_buf, non_blocking=True)
        d_off.copy_(p_off, non_blocking=True)
        return gpu.wire_decode_dev("proxy_leader_inbound", d_buf, d_off)

    t = timed(leg, 10)
    res["h2d_kernel_msgs_per_s"], res["h2d_kernel_ms"] = n / t, t * 1e3

    ch = torch.zeros(n, dtype=torch.uint8, device=dev)
    cv = torch.zeros(n, dtype=torch.int32, device=dev)

    def whole():
        gpu.reset()
        dd = leg()
        gpu.phase2_fused_dev(dd["slot"], dd["round"], dd["value_id"], None, ch, None, cv)

    t = timed(whole, 5)
    assert gpu.sync() == 0 and bool(ch.all())
    res["h2d_kernel_phase2_msgs_per_s"], res["h2d_kernel_phase2_ms"] = n / t, t * 1e3
    print(json.dumps(res))


if __name__ == "__main__":
    main()
