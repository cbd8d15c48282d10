"""config 2 (65 536 slots x 3 acceptors per step) eagerly and as ONE captured HIP graph of all steps: is the step bound by the
host's two launches per step or by the GPU's dependent launches?  """
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import frankenpaxos_amd as fa
from bench import steady_values_torch

dev = torch.device("cuda:0")
for n, R, f in ((65536, 3, 1), (16384, 3, 1), (4096, 3, 1), (65536, 7, 3)):
    K = 40
    ctx = fa.Context(fa.make_config(num_slots=(2 * K + 4) * n, num_replicas=R, f=f, tally_ways=4, flags=fa.FPX_F_TRUSTED))
    side = torch.cuda.Stream()
    steps = []
    for w in range(2 * K + 4):
        slot = torch.arange(w * n, (w + 1) * n, dtype=torch.int32, device=dev)
        steps.append((slot, torch.zeros_like(slot), steady_values_torch(slot), torch.zeros(n, dtype=torch.uint8, device=dev),
                      torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)))
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        ctx.set_stream(side.cuda_stream)
        assert ctx.acceptor_phase1a(0, 0)[0] == 0
        for i in range(4):
            ctx.phase2_fused_dev(steps[i][0], steps[i][1], steps[i][2], None, *steps[i][3:])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(4, 4 + K):
            ctx.phase2_fused_dev(steps[i][0], steps[i][1], steps[i][2], None, *steps[i][3:])
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / K
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(4 + K, 4 + 2 * K):
                ctx.phase2_fused_dev(steps[i][0], steps[i][1], steps[i][2], None, *steps[i][3:])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / K
        assert ctx.sync() == 0
    ok = all(bool(s[3].all()) for s in steps[4:])
    print("n = %6d x R = %d: eager %.4f ms per step (%.3e slots/s), one graph of %d steps %.4f ms per step (%.3e slots/s); all chosen: %s"
          % (n, R, eager * 1e3, n / eager, K, graph * 1e3, n / graph, ok), flush=True)
    ctx.set_stream(None)
    ctx.close()
