"""What does PCIe traffic cost the vote kernel it runs beside?  (profiles/r06_host_path.md)
The headline step (2^20 fresh slots x 256, ballot per cell) with device-resident inputs, timed by the library's events on the
vote kernel's dispatch packet (fpx_profile_*) and by wall clock over 12 steps, while
  none      nothing else happens
  h2d_sdma  a side stream copies 3 x 4 MB page-locked -> device per step with hipMemcpyAsync (copy engine)
  d2h_copy  a side stream copies 1 + 4 + 4 MB device -> page-locked per step with hipMemcpyAsync
  out_zero  the vote kernel writes its Chosen records straight into page-locked host memory (zero copy), inputs in HBM
  in_zero   the vote kernel reads slot / round / value straight from page-locked host memory, records to HBM
  both_zero both"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import frankenpaxos_amd as fa
from bench import steady_values_torch

B, R, K, Wm = 1 << 20, 256, 12, 3
dev = torch.device("cuda", 0)
L = fa.lib()


def run(mode):
    ctx = fa.Context(fa.make_config(num_slots=B * (K + Wm), num_replicas=R, f=127, ballot_mode=fa.FPX_BALLOT_PER_SLOT, tally_ways=4,
                                    flags=fa.FPX_F_TRUSTED))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    assert ctx.acceptor_phase1a(0, 0)[0] == 0
    ctx.flush_promises()
    side = torch.cuda.Stream()
    hin = [torch.empty(B, dtype=torch.int32).pin_memory() for _ in range(3)]
    din = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(3)]
    hout = [torch.empty(B, dtype=dt).pin_memory() for dt in (torch.uint8, torch.int32, torch.int32)]
    dout = [torch.empty(B, dtype=dt, device=dev) for dt in (torch.uint8, torch.int32, torch.int32)]
    steps = []
    for k in range(K + Wm):
        slot = torch.arange(k * B, (k + 1) * B, dtype=torch.int32, device=dev)
        steps.append((slot, torch.zeros_like(slot), steady_values_torch(slot)))
    hs = [(s.cpu().pin_memory(), r.cpu().pin_memory(), v.cpu().pin_memory()) for s, r, v in steps] if mode in ("in_zero", "both_zero") else None
    ptr = lambda t: C.c_void_p(t.data_ptr())

    def step(k):
        s, r, v = steps[k]
        if mode == "h2d_sdma":
            with torch.cuda.stream(side):
                for a, b in zip(din, hin):
                    a.copy_(b, non_blocking=True)
        if mode == "d2h_copy":
            with torch.cuda.stream(side):
                for a, b in zip(hout, dout):
                    a.copy_(b, non_blocking=True)
        ins = [ptr(x) for x in (hs[k] if hs else (s, r, v))]
        outs = [ptr(x) for x in (hout if mode in ("out_zero", "both_zero") else dout)]
        st = L.fpx_phase2_fused_dev(ctx._h, B, ins[0], ins[1], ins[2], None, outs[0], outs[1], outs[2], None)
        assert st == 0

    for k in range(Wm):
        step(k)
    assert ctx.sync() == 0
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for k in range(Wm, Wm + K):
        step(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    per = ctx.profile_read_launches()
    assert ctx.sync() == 0
    if mode in ("out_zero", "both_zero"):
        assert bool(hout[0].all()) and bool((hout[2] == steps[Wm + K - 1][2].cpu()).all())
    else:
        assert bool(dout[0].all())
    print("%-10s wall %.4f ms per step, vote kernel avg %.4f ms (min %.4f max %.4f)" % (mode, dt * 1e3, sum(per) / len(per), min(per), max(per)), flush=True)
    ctx.close()


for m in (sys.argv[1:] or ["none", "h2d_sdma", "d2h_copy", "out_zero", "in_zero", "both_zero", "none"]):
    run(m)
