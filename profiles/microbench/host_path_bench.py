"""PCIe-inclusive rate of the host-pointer entry point fpx_phase2_fused (measurement aid, DESIGN.md
section 6): 2^20 fresh slots x 256 acceptors per call, inputs and outputs in host memory -- pageable
numpy arrays vs page-locked buffers from fpx_host_alloc.  Never the bench.py `value`.
All batches are built BEFORE the timed calls and the calls run back to back (a server's event loop): the r01
version built the next batch with numpy between calls, so the GPU idled and every call paid its clock ramp."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import frankenpaxos_amd as fa
from tests import workloads as W

B, R, CALLS = 1 << 20, 256, 12
ctx = fa.Context(fa.make_config(num_slots=B * (CALLS + 1), num_replicas=R, f=127, ballot_mode=fa.FPX_BALLOT_PER_SLOT))
ctx.acceptor_phase1a(0, 0)
ctx.flush_promises()
L = fa.lib()
p = lambda a: a.ctypes.data_as(C.c_void_p)


def run(tag, alloc):
    keep, batches = [], []
    for k in range(CALLS):
        objs = [alloc((B,), np.int32), alloc((B,), np.int32), alloc((B,), np.int32), alloc((B,), np.uint8),
                alloc((B,), np.int32), alloc((B,), np.int32)]
        keep.append(objs)
        s, r, v, och, ocr, ocv = [x.array if hasattr(x, "array") else x for x in objs]
        s[:] = np.arange(k * B, (k + 1) * B, dtype=np.int32)
        r[:] = 0
        v[:] = W.steady_values(s)
        batches.append((s, r, v, och, ocr, ocv))
    times = []
    for s, r, v, och, ocr, ocv in batches:
        t0 = time.perf_counter()
        st = L.fpx_phase2_fused(ctx._h, B, p(s), p(r), p(v), None, p(och), p(ocr), p(ocv), None)
        times.append(time.perf_counter() - t0)
        assert st == 0
    for s, r, v, och, ocr, ocv in batches:
        assert int(och.sum()) == B and (ocv == v).all()
    dt = sorted(times[2:])[len(times[2:]) // 2]
    print("%-12s median %.3f ms (min %.3f) per 2^20-slot call  %.3e slots/s end-to-end (21 B/slot over PCIe = %.1f GB/s)"
          % (tag, dt * 1e3, min(times[2:]) * 1e3, B / dt, 21 * B / dt / 1e9), flush=True)
    ctx.reset()
    ctx.acceptor_phase1a(0, 0)
    ctx.flush_promises()


def run_async(tag):
    """fpx_phase2_fused_submit / _wait: up to 3 calls in flight, waited for in order"""
    keep, batches = [], []
    for k in range(CALLS):
        objs = [fa.PinnedArray((B,), dt) for dt in (np.int32, np.int32, np.int32, np.uint8, np.int32, np.int32)]
        keep.append(objs)
        s, r, v, och, ocr, ocv = [x.array for x in objs]
        s[:] = np.arange(k * B, (k + 1) * B, dtype=np.int32)
        r[:] = 0
        v[:] = W.steady_values(s)
        batches.append((s, r, v, och, ocr, ocv))
    tick = C.c_int32()
    inflight, done_at = [], []
    t0 = time.perf_counter()
    for s, r, v, och, ocr, ocv in batches:
        if len(inflight) == 3:
            assert L.fpx_phase2_fused_wait(ctx._h, inflight.pop(0)) == 0
            done_at.append(time.perf_counter())
        assert L.fpx_phase2_fused_submit(ctx._h, B, p(s), p(r), p(v), None, p(och), p(ocr), p(ocv), None, C.byref(tick)) == 0
        inflight.append(tick.value)
    while inflight:
        assert L.fpx_phase2_fused_wait(ctx._h, inflight.pop(0)) == 0
        done_at.append(time.perf_counter())
    for s, r, v, och, ocr, ocv in batches:
        assert int(och.sum()) == B and (ocv == v).all()
    per = (done_at[-1] - done_at[2]) / (len(done_at) - 3)
    print("%-12s %.3f ms per 2^20-slot call sustained (calls 4..%d), %.3e slots/s end-to-end (21 B/slot over PCIe = %.1f GB/s); all %d calls %.3f ms"
          % (tag, per * 1e3, CALLS, B / per, 21 * B / per / 1e9, CALLS, (done_at[-1] - t0) * 1e3), flush=True)
    ctx.reset()
    ctx.acceptor_phase1a(0, 0)
    ctx.flush_promises()


run("pageable", lambda shape, dt: np.zeros(shape, dt))
run_async("page-locked, 3 calls in flight")
run("page-locked (staged by kernels)", lambda shape, dt: fa.PinnedArray(shape, dt))
os.environ["FPX_HOST_NO_STAGE"] = "1"
run("page-locked, copy engines", lambda shape, dt: fa.PinnedArray(shape, dt))
