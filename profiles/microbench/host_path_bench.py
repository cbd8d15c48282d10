"""PCIe-inclusive rate of the host-pointer entry point fpx_phase2_fused (measurement aid, DESIGN.md
section 6): 2^20 fresh slots x 256 acceptors per call, inputs and outputs in host memory -- pageable
numpy arrays vs page-locked buffers from fpx_host_alloc.  Never the bench.py `value`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import frankenpaxos_amd as fa
from tests import workloads as W

B, R, CALLS = 1 << 20, 256, 6
ctx = fa.Context(fa.make_config(num_slots=B * (CALLS + 1), num_replicas=R, f=127, ballot_mode=fa.FPX_BALLOT_PER_SLOT))
L = fa.lib()
import ctypes as C


def run(tag, alloc):
    slot, rnd, val = alloc((B,), np.int32), alloc((B,), np.int32), alloc((B,), np.int32)
    ch, cr, cv = alloc((B,), np.uint8), alloc((B,), np.int32), alloc((B,), np.int32)
    arrs = [x.array if hasattr(x, "array") else x for x in (slot, rnd, val, ch, cr, cv)]
    s, r, v, och, ocr, ocv = arrs
    r[:] = 0
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    times = []
    for k in range(CALLS):
        base = (k + (0 if tag == "pageable" else 0)) * B
        s[:] = np.arange(base, base + B, dtype=np.int32)
        v[:] = W.steady_values(s)
        t0 = time.perf_counter()
        st = L.fpx_phase2_fused(ctx._h, B, p(s), p(r), p(v), None, p(och), p(ocr), p(ocv), None)
        times.append(time.perf_counter() - t0)
        assert st == 0 and int(och.sum()) == B
    dt = min(times[1:])
    print("%-12s %.3f ms per 2^20-slot call  %.3e slots/s end-to-end (21 B/slot over PCIe = %.1f GB/s)"
          % (tag, dt * 1e3, B / dt, 21 * B / dt / 1e9))
    ctx.reset()


run("pageable", lambda shape, dt: np.zeros(shape, dt))
run("page-locked", lambda shape, dt: fa.PinnedArray(shape, dt))
