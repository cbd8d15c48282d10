"""4 staged host calls under rocprofv3 --kernel-trace: where do k_stage and k_phase2 sit on the timeline?"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import frankenpaxos_amd as fa
from tests import workloads as W
B, R, CALLS = 1 << 20, 256, 5
ctx = fa.Context(fa.make_config(num_slots=B * (CALLS + 1), num_replicas=R, f=127, ballot_mode=fa.FPX_BALLOT_PER_SLOT))
ctx.acceptor_phase1a(0, 0); ctx.flush_promises()
L = fa.lib(); p = lambda a: a.ctypes.data_as(C.c_void_p)
bat = []
for k in range(CALLS):
    a = [fa.PinnedArray((B,), dt) for dt in (np.int32, np.int32, np.int32, np.uint8, np.int32, np.int32)]
    a[0].array[:] = np.arange(k * B, (k + 1) * B); a[1].array[:] = 0; a[2].array[:] = W.steady_values(a[0].array)
    bat.append(a)
for a in bat:
    t0 = time.perf_counter()
    st = L.fpx_phase2_fused(ctx._h, B, p(a[0].array), p(a[1].array), p(a[2].array), None, p(a[3].array), p(a[4].array), p(a[5].array), None)
    print("call %.3f ms st %d" % ((time.perf_counter() - t0) * 1e3, st), flush=True)
