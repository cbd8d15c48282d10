"""staged probe used to localise a fault seen only under rocprofv3 (measurement aid)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import frankenpaxos_amd as fa
stage = sys.argv[1]
windows = int(sys.argv[2]) if len(sys.argv) > 2 else 2
S = windows << 20
print("stage", stage, "windows", windows, flush=True)
dev = torch.device("cuda:0")
ctx = fa.Context(fa.make_config(num_slots=S, num_replicas=256, f=127, ballot_mode=1, flags=fa.FPX_F_TRUSTED))
print("created", ctx.device_bytes, flush=True)
if stage == "create": sys.exit(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
print(ctx.acceptor_phase1a(0, 0)[0], "phase1a", flush=True)
if stage == "phase1a": sys.exit(0)
slot = torch.arange(0, 1 << 20, dtype=torch.int32, device=dev)
rnd = torch.zeros(1 << 20, dtype=torch.int32, device=dev)
val = slot * 3
ch = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
cr = torch.empty(1 << 20, dtype=torch.int32, device=dev)
cv = torch.empty(1 << 20, dtype=torch.int32, device=dev)
if stage == "prof":
    ctx.profile_enable(True)
ctx.phase2_fused_dev(slot, rnd, val, None, ch, cr, cv)
print("sync", ctx.sync(), int(ch.sum()), flush=True)
if stage == "prof":
    print(ctx.profile_read(), flush=True)
