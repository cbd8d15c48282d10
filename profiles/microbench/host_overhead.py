"""Host time per device-pointer call (no GPU work to speak of: 64 messages per call, 2000 calls queued back to back,
one synchronise at the end): what the Python binding + the C entry point + the HIP launches cost per call."""
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import frankenpaxos_amd as fa  # noqa: E402

dev = torch.device("cuda:0")
N, n = 2000, 64
for label, kw, prof in (("MultiPaxos 3 acceptors", dict(num_groups=1), False), ("the same, profiling events on", dict(num_groups=1), True),
                        ("Mencius 8 leader groups", dict(num_groups=1, num_leader_groups=8), False)):
    ctx = fa.Context(fa.make_config(num_slots=N * n, num_replicas=3, f=1, tally_ways=4, flags=fa.FPX_F_TRUSTED, **kw))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if prof:
        ctx.profile_enable(True) if hasattr(ctx, "profile_enable") else None
    slot = torch.arange(N * n, dtype=torch.int32, device=dev)
    rnd, val = torch.zeros_like(slot), slot * 3
    ch = torch.zeros(N * n, dtype=torch.uint8, device=dev)
    cv = torch.zeros_like(slot)
    views = [(slot[i * n:(i + 1) * n], rnd[i * n:(i + 1) * n], val[i * n:(i + 1) * n], ch[i * n:(i + 1) * n], cv[i * n:(i + 1) * n]) for i in range(N)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s, r, v, c, w in views:
        ctx.phase2_fused_dev(s, r, v, None, c, None, w)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    assert bool(ch.all())
    print("%-32s phase2_fused_dev: %.2f us of host time per call (queue drained %.2f ms later)" % (label, (t1 - t0) / N * 1e6, (t2 - t1) * 1e3))
    # the same calls with the arguments converted once (what a C caller pays: the entry point and its launches)
    L = ctx.L
    args = [(ctx._h, n, s.data_ptr(), r.data_ptr(), v.data_ptr(), None, c.data_ptr(), None, w.data_ptr(), None) for s, r, v, c, w in views]
    ch.zero_()
    ctx2 = fa.Context(fa.make_config(num_slots=N * n, num_replicas=3, f=1, tally_ways=4, flags=fa.FPX_F_TRUSTED, **kw))
    ctx2.set_stream(torch.cuda.current_stream().cuda_stream)
    args = [(ctx2._h,) + a[1:] for a in args]
    f = L.fpx_phase2_fused_dev
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a in args:
        f(*a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("%-32s   the C entry point alone (arguments converted beforehand): %.2f us per call" % ("", (t1 - t0) / N * 1e6))
