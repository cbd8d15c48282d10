#!/usr/bin/env python3
"""bench.py -- committed log slots / second of the fused Phase-2 step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json metric; SURVEY.md section 8d "steady"): a "step" is one fused pass
(ProxyLeader.handlePhase2a -> Acceptor.handlePhase2a x 256 -> ProxyLeader.handlePhase2b) over one
batch of 2^20 fresh log slots x 256 acceptors, threshold quorum f+1 = 128, dense delivery, values
splitmix64(0xF9A405 + slot) & 0x7fffffff, after the leader's one-off Phase 1 in round 0.  Inputs are
resident in HBM; every step works on the next 2^20 slots of the log, so every row is cold.

Ballot model: FPX_BALLOT_PER_SLOT (the "generalised ballot[S x R]" model of SURVEY.md 8d, 3088
algorithmic B/slot: read ballot row 1024 + proposal 8, write voteRound 1024 + voteValue 1024 +
chosen record 8).  `--ballot acceptor` runs the faithful per-acceptor scalar model (2064 B/slot).

N > 1: one process per GPU, every rank owns its own acceptor groups (its own 2^20 x 256 grid per
step): slot-partition sharding, no data-path collective; weak scaling.
`--shard replica` instead splits the 256 acceptors of ONE grid across the ranks: K1 on every rank's
acceptor columns, one RCCL reduce-scatter(sum) of the per-slot vote bitmaps (sum == OR, the bit ranges are
disjoint), K2 on each rank's 1/N of the slots; strong scaling of the replica axis.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

# Test hooks (tests/test_bench_distributed.py: many ranks on ONE GPU over gloo, small windows, the RCCL double) are
# environment variables that only count together with the --test-hooks flag (a leaked variable must not
# change the headline workload silently).  Without the flag a set hook is REFUSED, with it the line carries `test_hooks`.
HOOK_VARS = ("FPX_BENCH_SLOTS_LOG2", "FPX_BENCH_SHARE_GPU", "FPX_BENCH_BACKEND", "FPX_BENCH_FPX_COMM", "FPX_BENCH_DRY_SPAWN")
TEST_HOOKS = "--test-hooks" in sys.argv
HOOKS_SET = sorted(v for v in HOOK_VARS if os.environ.get(v))
if HOOKS_SET and not TEST_HOOKS:
    raise SystemExit("bench.py: %s set without --test-hooks: refusing to run a workload other than the one the line names"
                     % ", ".join(HOOKS_SET))


def hook(name, default=None):
    return os.environ.get(name, default) if TEST_HOOKS else default


SLOTS_PER_STEP = 1 << int(hook("FPX_BENCH_SLOTS_LOG2", "20"))
REPLICAS = 256
F = 127
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MAX_WINDOWS = 56       # log windows kept in HBM (3.3 GiB each in PER_SLOT mode: 185 GiB of 288 GB)


def splitmix64_torch(x):
    """one splitmix64 output per int64 state (wrap-around arithmetic), on the tensor's device"""
    def lsr(v, k):  # logical shift right on int64
        return (v >> k) & ((1 << (64 - k)) - 1)
    z = x + (-0x61C8864680B583EB)  # 0x9E3779B97F4A7C15 as int64
    z = (z ^ lsr(z, 30)) * (-0x40A7B892E31B1A47)  # 0xBF58476D1CE4E5B9
    z = (z ^ lsr(z, 27)) * (-0x6B2FB644ECCEEE15)  # 0x94D049BB133111EB
    return z ^ lsr(z, 31)


def steady_values_torch(slot):
    return (splitmix64_torch(slot.to(torch.int64) + 0xF9A405) & 0x7FFFFFFF).to(torch.int32)


# BASELINE.json's metric, verbatim
METRIC = "committed log slots/sec at 1M slots \u00d7 256 replicas; 1/2/4/8-GPU scaling"


# profiles/r01_hbm_mix.txt (1 x MI355X, bare streaming kernels): pure read, pure write, copy, and this path's 1 read : 2 write mix
MEASURED_STREAM_GBS = {"read": 5871.4, "write": 5868.0, "copy": 5140.2, "mix_1r_2w": 5068.0}


def algorithmic_bytes_per_slot(ballot_mode):
    r = REPLICAS
    if ballot_mode == 1:  # SURVEY.md 8d generalised model
        return 4 * r + 8 + 8 * r + 8  # 3088
    return 8 + 8 * r + 8              # faithful scalar model: reads 8 B/slot (+1 KiB/batch)


def read_bytes_per_slot(ballot_mode):
    """the reads of SURVEY.md 8(d)'s byte model alone: ballot row + proposal (round, value)"""
    return (4 * REPLICAS + 8) if ballot_mode == 1 else 8


def cpu_baseline(ballot_mode, light=False):
    """The CPU oracle (a plain-C, single-threaded port of the reference handlers; the JVM reference
    cannot run here) timed on this box on a bounded sample of the same workload.  light: only the
    single-thread flat port (what an N > 1 line carries: the other ranks wait for rank 0 meanwhile)."""
    from oracle import pyoracle
    from tests import workloads as W

    pyoracle.build()
    S = 1 << 19
    ref = pyoracle.System(pyoracle.make_config(num_slots=S, num_replicas=REPLICAS, f=F,
                                               ballot_mode=ballot_mode))
    ref.acceptor_phase1a(0, 0)
    slot, rnd, val = W.steady_stream(S)
    t0 = time.perf_counter()
    st, ch, cr, cv, nr = ref.phase2_fused(slot, rnd, val)
    dt = time.perf_counter() - t0
    assert st == 0 and int(ch.sum()) == S
    if light:
        return {"value": S / dt, "unit": "slots/s", "cores": 1, "kind": "port",
                "sample": "oracle/fpx_oracle.c fpo_phase2_fused (flat arrays), 2^19 slots x 256 acceptors, steady stream, "
                          "1 thread, on rank 0's host cores; the N = 1 line carries the other CPU forms; nproc=%d"
                          % os.cpu_count()}
    # the same handlers behind a strict FIFO message pump (stand-in for the in-process Transport)
    Sp = 1 << 15
    ref2 = pyoracle.System(pyoracle.make_config(num_slots=Sp, num_replicas=REPLICAS, f=F,
                                                ballot_mode=ballot_mode))
    ref2.acceptor_phase1a(0, 0)
    t1 = time.perf_counter()
    ref2.phase2_fifo_pump(slot[:Sp], rnd[:Sp], val[:Sp])
    dtp = time.perf_counter() - t1
    # B1 "faithful shapes" (BASELINE.md section 2): the same handlers over the reference's data structures
    # (SortedMap per acceptor, HashMap of Pending with a vote map, one heap object per message, FIFO pump)
    import ctypes as C
    faithful = None
    fso = os.path.join(ROOT, "oracle", "libfpx_faithful.so")
    if not os.path.exists(fso):
        import subprocess
        subprocess.call(["make", "-C", os.path.join(ROOT, "oracle"), "libfpx_faithful.so"],
                        stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if os.path.exists(fso):
        L = C.CDLL(fso)
        L.fpo_faithful_run.restype = C.c_int64
        L.fpo_faithful_run.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_int64)]
        Sf = 1 << 14
        v = np.ascontiguousarray(val[:Sf])
        cs = C.c_int64()
        t2 = time.perf_counter()
        n = L.fpo_faithful_run(Sf, REPLICAS, F, v.ctypes.data, C.byref(cs))
        dtf = time.perf_counter() - t2
        assert n == Sf and cs.value == int(v.astype(np.int64).sum())
        faithful = Sf / dtf
    # the flat port on all host cores, slots partitioned over threads exactly as acceptor groups partition
    # over GPUs (SURVEY.md 8d "OpenMP over slots"): shared-nothing systems, ctypes releases the GIL
    from concurrent.futures import ThreadPoolExecutor
    T = max(1, min(os.cpu_count() or 1, 64))
    St = 1 << 15
    systems = []
    for _ in range(T):
        sy = pyoracle.System(pyoracle.make_config(num_slots=St, num_replicas=REPLICAS, f=F, ballot_mode=ballot_mode))
        sy.acceptor_phase1a(0, 0)
        systems.append(sy)
    sl, rn, vl = slot[:St].copy(), rnd[:St].copy(), val[:St].copy()

    def one(sy):
        out = sy.phase2_fused(sl, rn, vl)
        return int(out[1].sum())

    with ThreadPoolExecutor(T) as pool:
        t3 = time.perf_counter()
        done = sum(pool.map(one, systems))
        dta = time.perf_counter() - t3
    assert done == T * St
    all_cores = T * St / dta
    del systems
    return {
        "value": S / dt, "unit": "slots/s", "cores": 1, "kind": "port",
        "all_cores_value": all_cores, "all_cores_threads": T,
        "fifo_pump_value": Sp / dtp, "faithful_shapes_value": faithful,
        "sample": "C/C++ restatements of the reference handlers, not the JVM (none in this image), steady stream x 256 acceptors: "
                  "value = flat arrays, 2^19 slots, 1 thread; fifo_pump = the same behind a FIFO message pump, 2^15 slots; "
                  "faithful_shapes = the reference's container shapes (oracle/fpx_faithful.cpp), 2^14 slots; all_cores = flat on "
                  "%d threads, 2^15 slots each; nproc=%d" % (T, os.cpu_count()),
    }


def setup_comm(fa, ctx, dist, backend, dev, rank, world):
    """The RCCL communicator lives behind the C ABI (fpx_comm_create); its 128-byte id travels over
    torch.distributed, which is control plane only here.  Under the gloo test hook the ranks share one GPU,
    which RCCL refuses (duplicate device): no communicator, the caller falls back to a host exchange."""
    if backend != "nccl" and hook("FPX_BENCH_FPX_COMM") != "1":
        return False
    # (FPX_BENCH_FPX_COMM=1, test hook: the ranks share one GPU and rendezvous over gloo, but the data path still goes
    # through fpx_comm_create and the library's collectives -- bound to the test double FPX_RCCL_LIB names, since RCCL
    # itself refuses two ranks on one device: tests/test_bench_distributed.py runs the 8-rank lines that way)
    idt = torch.zeros(fa.FPX_COMM_ID_BYTES, dtype=torch.uint8, device=dev if backend == "nccl" else "cpu")
    if rank == 0:
        idt.copy_(torch.frombuffer(bytearray(fa.comm_unique_id()), dtype=torch.uint8))
    dist.broadcast(idt, 0)
    ctx.comm_create(idt.cpu().numpy().tobytes(), rank, world)
    return True


def replica_axis_row(fa, dist, backend, dev, rank, world, local_rank, ballot_mode, K, all_reduce):
    """ONE 2^20-slot x 256-acceptor grid per step, its acceptor columns split over the
    ranks; per step K1 on every rank, reduce-scatter(sum) of the per-slot vote bitmaps over xGMI, K2 on each
    rank's slice.  Strong scaling of the acceptor axis; returns the row (rank 0) with the collective broken out."""
    from frankenpaxos_amd import sharding
    R_local = REPLICAS // world
    windows = K + 1
    ctx = fa.Context(fa.make_config(
        num_slots=windows * SLOTS_PER_STEP, num_replicas=R_local, f=F, quorum_kind=fa.FPX_Q_THRESHOLD,
        ballot_mode=ballot_mode, tally_ways=4, device=local_rank, flags=fa.FPX_F_TRUSTED,
        replica_base=rank * R_local, replicas_total=REPLICAS))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    assert ctx.acceptor_phase1a(0, 0)[0] == 0
    ctx.flush_promises()
    have_comm = setup_comm(fa, ctx, dist, backend, dev, rank, world)
    lo, hi = sharding.slot_slice(SLOTS_PER_STEP, world, rank)
    per = hi - lo
    steps = []
    for w in range(windows):
        slot = torch.arange(w * SLOTS_PER_STEP, (w + 1) * SLOTS_PER_STEP, dtype=torch.int32, device=dev)
        steps.append((slot, torch.zeros_like(slot), steady_values_torch(slot),
                      torch.zeros(per, dtype=torch.uint8, device=dev),
                      torch.full((per,), -7, dtype=torch.int32, device=dev),
                      torch.full((per,), -7, dtype=torch.int32, device=dev)))
    if not have_comm:
        vb = torch.empty((SLOTS_PER_STEP, 4), dtype=torch.int64, device=dev)

    def step(i):
        slot, rnd, val, ch, cr, cv = steps[i]
        if have_comm:
            ctx.phase2_replica_sharded_dev(slot, rnd, val, None, ch, cr, cv)
        else:
            ctx.acceptor_phase2a_dev(slot, rnd, val, None, vb, None, None)
            all_reduce(vb, dist.ReduceOp.SUM)
            ctx.proxy_open_dev(slot[lo:hi], rnd[lo:hi], val[lo:hi])
            ctx.proxy_phase2b_dev(slot[lo:hi], rnd[lo:hi], vb[lo:hi].contiguous(), ch, cr, cv)

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    step(0)
    assert ctx.sync() == 0
    ctx.profile_enable(True)
    fence()
    t0 = time.perf_counter()
    for i in range(1, windows):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    k1_n, k1_ms = ctx.profile_read()
    c_n, c_ms = ctx.profile_read_collective()
    assert ctx.sync() == 0
    committed = 0
    for i in range(1, windows):
        slot, rnd, val, ch, cr, cv = steps[i]
        assert bool(ch.all()) and bool((cv == val[lo:hi]).all()) and bool((cr == 0).all()), "replica-axis step %d" % i
        committed += int(ch.sum().item())
    t = torch.tensor([elapsed, float(committed), c_ms / max(c_n, 1)], dtype=torch.float64, device=dev)
    tm = t.clone()
    all_reduce(tm, dist.ReduceOp.MAX)
    all_reduce(t, dist.ReduceOp.SUM)
    ctx.close()
    if rank != 0:
        return None
    el = float(tm[0].item())
    return {
        "workload": "ONE 2^20 x 256 grid per step, acceptor columns split over the ranks (%d per GPU): K1 -> "
                    "reduce-scatter(sum) of the vote bitmaps -> K2 on 1/N of the slots" % R_local,
        "scaling": "strong", "steps": K, "value": float(t[1].item()) / el, "unit": "slots/s",
        "ms_per_step": el / K * 1e3, "rccl_ranks": world if have_comm else 0,
        "exchange": "ncclReduceScatter(ncclSum, ncclUint64) inside fpx_phase2_replica_sharded_dev" if have_comm
                    else "all_reduce over gloo (single-GPU test hook)",
        "collective_avg_ms_max_over_ranks": float(tm[2].item()) if c_n else None,
        "k1_avg_ms_rank0": k1_ms / max(k1_n, 1),
        "xgmi_bytes_per_gpu_per_step": (world - 1) * 32 * SLOTS_PER_STEP // world,
    }


def sig(x, digits=5):
    """a float rounded to `digits` significant digits (the line is kept short: the driver keeps 8 KB of it)"""
    if x is None or isinstance(x, (bool, int)):
        return x
    return float("%.*g" % (digits, x))


def round_floats(x, digits=7):
    if isinstance(x, float):
        return sig(x, digits)
    if isinstance(x, dict):
        return {k: round_floats(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [round_floats(v, digits) for v in x]
    return x


def compact_entry(full):
    """One `configs` entry from a bench_configs line: the numbers only.  What the workload is, which kernels run, the byte
    model and what was verified after the timed region are prose in bench_configs.md, keyed by the entry's name (VERDICT
    r05 weak #9: the driver keeps the last 8 KB of the line, and the prose pushed three entries out of its record).
    `verified: true` = the entry's post-region check ran and passed (a failing check raises: the entry is then an error)."""
    r, c = full["roofline"], full["config"]
    e = {"value": sig(full["value"]), "unit": full["unit"], "steps": full["steps"], "ms_per_step": sig(full["ms_per_step"]),
         "avg_kernel_ms": sig(r["avg_kernel_ms"]), "roofline_frac": sig(r["frac"], 4),
         "algorithmic_bytes_per_unit": r.get("algorithmic_bytes_per_unit"),
         "traffic": sig(r.get("traffic"), 4), "traffic_round": r.get("traffic_round"), "verified": bool(c.get("verified"))}
    for k in ("pcie_GBs", "proposals", "chosen", "nacked"):
        if k in c:
            e[k] = sig(c[k], 4)
    return e


def run_with_deadline(fn, seconds, dev):
    """fn() on a worker thread: (result, False), ({"error": ...}, False) if it raised, or ({"error": ...}, True) if it
    has not returned within `seconds` (the thread is left behind; the caller must end the process with os._exit)"""
    import threading
    box = {}

    def target():
        try:
            if dev is not None and getattr(dev, "type", "") == "cuda":
                torch.cuda.set_device(dev)
            box["value"] = fn()
        except BaseException as e:  # noqa: BLE001 -- reported, not swallowed
            box["error"] = "%s: %s" % (type(e).__name__, e)

    th = threading.Thread(target=target, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        return {"error": "no answer within %d s" % seconds}, True
    if "error" in box:
        return {"error": box["error"]}, False
    return box.get("value"), False


def preflight(args, fa, dist, backend, dev, rank, world, local_rank, all_reduce):
    """what the first real multi-GPU run exercises for the first time, on its own and under
    deadlines -- N ranks creating headline-sized contexts at once (the placement search of every rank probing beside the
    others), ncclCommInitRank behind fpx_comm_create, one replica-sharded step (K1 -> reduce-scatter of the vote bitmaps ->
    all-reduce(max) of the Nack rounds -> K2) and one all-gather of Chosen records.  Every rank reports; rank 0 prints ONE
    line; a rank that does not answer in time makes the line say so and the process leave with status 1."""
    out = {"rank": rank}
    t0 = time.perf_counter()

    def make():
        R_local = REPLICAS // world
        c = fa.Context(fa.make_config(num_slots=2 * SLOTS_PER_STEP, num_replicas=R_local, f=F, quorum_kind=fa.FPX_Q_THRESHOLD,
                                      ballot_mode=fa.FPX_BALLOT_PER_SLOT, tally_ways=4, device=local_rank, flags=fa.FPX_F_TRUSTED,
                                      replica_base=rank * R_local, replicas_total=REPLICAS))
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        return c
    # (a context of the full 256 acceptors first: 6 GiB of cells, the shape whose slab is placed by measurement)
    def make_full():
        c = fa.Context(fa.make_config(num_slots=2 * SLOTS_PER_STEP, num_replicas=REPLICAS, f=F, ballot_mode=fa.FPX_BALLOT_PER_SLOT,
                                      tally_ways=4, device=local_rank, flags=fa.FPX_F_TRUSTED))
        st = c.placement_stats()
        c.close()
        return st
    st, hung = run_with_deadline(make_full, args.comm_deadline, dev)
    out["create_full_s"] = time.perf_counter() - t0
    if hung or (isinstance(st, dict) and "error" in st):
        out["error"] = "fpx_create: %s" % st.get("error")
    else:
        out["placement"] = {"chunks": st["chunks"], "windows": st["windows"], "search_ms": st["search"]["ms"],
                            "probes": st["search"]["probes"], "unprobed_decisions": st["search"]["unprobed_decisions"]}
    ctx = None
    if "error" not in out:
        t1 = time.perf_counter()
        ctx, hung = run_with_deadline(make, args.comm_deadline, dev)
        if hung or isinstance(ctx, dict):
            out["error"] = "fpx_create (sharded): %s" % (ctx.get("error") if isinstance(ctx, dict) else "?")
            ctx = None
        out["create_sharded_s"] = time.perf_counter() - t1
    if ctx is not None:
        t2 = time.perf_counter()
        res, hung = run_with_deadline(lambda: setup_comm(fa, ctx, dist, backend, dev, rank, world), args.comm_deadline, dev)
        out["comm_create_s"] = time.perf_counter() - t2
        if hung or isinstance(res, dict):
            out["error"] = "fpx_comm_create: %s" % (res.get("error") if isinstance(res, dict) else "no answer")
        elif res:
            def one_step():
                from frankenpaxos_amd import sharding
                assert ctx.acceptor_phase1a(0, 0)[0] == 0
                ctx.flush_promises()
                lo, hi = sharding.slot_slice(SLOTS_PER_STEP, world, rank)
                slot = torch.arange(0, SLOTS_PER_STEP, dtype=torch.int32, device=dev)
                val = steady_values_torch(slot)
                ch = torch.zeros(hi - lo, dtype=torch.uint8, device=dev)
                cr = torch.full((hi - lo,), -7, dtype=torch.int32, device=dev)
                cv = torch.full((hi - lo,), -7, dtype=torch.int32, device=dev)
                ctx.profile_enable(True)
                ctx.phase2_replica_sharded_dev(slot, torch.zeros_like(slot), val, None, ch, cr, cv)
                assert ctx.sync() == 0
                n, ms = ctx.profile_read_collective()
                ok = bool(ch.all()) and bool((cv == val[lo:hi]).all()) and bool((cr == 0).all())
                alls = [torch.zeros((SLOTS_PER_STEP,), dtype=t.dtype, device=dev) for t in (ch, cr, cv)]
                equal = hi - lo == SLOTS_PER_STEP // world
                if equal:
                    ctx.comm_allgather_chosen_dev(ch, cr, cv, *alls)
                    assert ctx.sync() == 0
                    ok = ok and bool(alls[0].all()) and bool((alls[2] == val).all())
                return {"step_ok": ok, "collective_ms": ms / max(n, 1), "allgather_checked": equal, "rccl_ranks": ctx.comm_info()[1]}
            res2, hung = run_with_deadline(one_step, args.comm_deadline, dev)
            if hung or "error" in res2:
                out["error"] = "replica-sharded step: %s" % res2.get("error")
            else:
                out.update(res2)
        else:
            out["rccl_ranks"] = 0   # (gloo test hook without FPX_BENCH_FPX_COMM: no communicator to test)
    # the ranks' reports meet on rank 0 over the control plane (gloo objects; under nccl through a byte tensor)
    reports = [None] * world
    def gather():
        if backend == "nccl":
            blob = json.dumps(out).encode()
            buf = torch.zeros(4096, dtype=torch.uint8, device=dev)
            buf[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
            allb = [torch.zeros_like(buf) for _ in range(world)]
            dist.all_gather(allb, buf)
            return [json.loads(bytes(b.cpu().numpy().tobytes()).rstrip(b"\0").decode()) for b in allb]
        objs = [None] * world
        dist.all_gather_object(objs, out)
        return objs
    got, hung = run_with_deadline(gather, 60, dev)
    if hung or isinstance(got, dict):
        reports = [out]
        out.setdefault("error", "the ranks' reports did not meet: %s" % (got.get("error") if isinstance(got, dict) else "?"))
    else:
        reports = got
    ok = all("error" not in r and r.get("step_ok", backend != "nccl") for r in reports)
    if rank == 0:
        print(json.dumps(round_floats({
            "preflight": True, "ok": ok, "n_gpus": world, "backend": backend,
            "rccl_ranks": min((r.get("rccl_ranks", 0) for r in reports), default=0),
            "placement_search_ms_per_rank": [r.get("placement", {}).get("search_ms") for r in reports],
            "placement_unprobed_decisions_per_rank": [r.get("placement", {}).get("unprobed_decisions") for r in reports],
            "create_full_context_s_per_rank": [r.get("create_full_s") for r in reports],
            "comm_create_s_per_rank": [r.get("comm_create_s") for r in reports],
            "collective_ms_per_rank": [r.get("collective_ms") for r in reports],
            "errors": {str(r["rank"]): r["error"] for r in reports if "error" in r},
            "test_hooks": HOOKS_SET if TEST_HOOKS else None}, 5)), flush=True)
    sys.stdout.flush()
    os._exit(0 if ok else 1)   # (no teardown that could wait for a rank that is stuck)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ballot", choices=["per_slot", "acceptor"], default="per_slot")
    ap.add_argument("--shard", choices=["group", "replica"], default="group")
    ap.add_argument("--validate", action="store_true",
                    help="keep the run-contract validation kernel in the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", choices=["headline", "2", "3", "4", "5", "thrifty", "thrifty_random", "acceptor_model", "host_path", "adversarial", "4_execute"], default="headline",
                    help="headline = BASELINE.json's metric grid (2^20 slots x 256 acceptors); 2..5 = the other "
                         "BASELINE.json configs as bench lines of the same schema (bench_configs.py)")
    ap.add_argument("--configs-block-steps", type=int, default=20,
                    help="N = 1 headline run: steps of each of BASELINE.json's other configs (2..5) timed after the "
                         "headline and reported in the line's `configs` block (0 = leave the block out)")
    ap.add_argument("--replica-row-deadline", type=int, default=120,
                    help="seconds the extra replica-axis row may take before the line is printed without it")
    ap.add_argument("--preflight", action="store_true",
                    help="N > 1: do not run the bench; create a headline-sized context and an RCCL communicator on every rank "
                         "under deadlines, run one replica-sharded step, and print per-rank placement / create / communicator "
                         "times and the collective's event time as one JSON line (exit status 1 if any rank failed or hung)")
    ap.add_argument("--comm-deadline", type=int, default=120,
                    help="seconds fpx_comm_create may take on a rank before the run is given up (loudly, not hanging)")
    ap.add_argument("--test-hooks", action="store_true",
                    help="honour the FPX_BENCH_* test hooks of tests/test_bench_distributed.py (refused without this flag)")
    ap.add_argument("--replica-row-steps", type=int, default=5,
                    help="N > 1, --shard group: steps of the extra replica-axis row (0 = skip it)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: become N ranks (one per GPU) under torch.distributed.run.
        # rank 0 of the children prints the one JSON line on the inherited stdout.
        import socket
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        if hook("FPX_BENCH_DRY_SPAWN") == "1":   # test hook: show the launch instead of doing it
            print(json.dumps({"spawn": cmd}))
            return
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d -- refusing to report a %d-GPU number as a "
                         "%d-GPU one" % (args.gpus, world, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libfpx has no CPU path)")
    # test hooks (tests/test_bench_distributed.py): run several ranks on ONE GPU over gloo, so that the
    # N > 1 control flow of this file can be exercised on a 1-GPU box.  Never set by the driver.
    share_gpu = hook("FPX_BENCH_SHARE_GPU") == "1"
    backend = hook("FPX_BENCH_BACKEND", "nccl")
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    def all_reduce(t, op):
        """all-reduce of a device tensor; staged through the host only under the gloo test hook"""
        if backend == "nccl":
            dist.all_reduce(t, op=op)
        else:
            c = t.cpu()
            dist.all_reduce(c, op=op)
            t.copy_(c)

    import frankenpaxos_amd as fa

    if args.preflight:
        if world < 2:
            raise SystemExit("bench.py --preflight is about N > 1 (--gpus N)")
        preflight(args, fa, dist, backend, dev, rank, world, local_rank, all_reduce)

    if args.config != "headline":
        import bench_configs
        line = bench_configs.run(args, fa, dist, dev, rank, world, local_rank, all_reduce)
        if rank == 0:
            print(json.dumps(line))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    ballot_mode = fa.FPX_BALLOT_PER_SLOT if args.ballot == "per_slot" else fa.FPX_BALLOT_ACCEPTOR
    K, Wm = args.steps, args.warmup
    windows = min(K + Wm, MAX_WINDOWS)
    # live (slot, round) tallies per slot: 4 unless the run laps the window ring more than 4 times
    TALLY_WAYS = 4 if K + Wm <= 4 * windows else 8
    S_total = windows * SLOTS_PER_STEP
    replica_shard = args.shard == "replica" and world > 1
    R_local = REPLICAS // world if replica_shard else REPLICAS
    flags = 0 if args.validate else fa.FPX_F_TRUSTED
    ctx = fa.Context(fa.make_config(
        num_slots=S_total, num_replicas=R_local, f=F, quorum_kind=fa.FPX_Q_THRESHOLD,
        ballot_mode=ballot_mode, tally_ways=TALLY_WAYS, device=local_rank, flags=flags,
        replica_base=(rank * R_local if replica_shard else 0),
        replicas_total=(REPLICAS if replica_shard else 0)))
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    # the leader's Phase 1 in round 0 (once, before any Phase 2): every acceptor promises round 0
    st, pb, nb = ctx.acceptor_phase1a(0, 0)
    assert st == 0
    ctx.flush_promises()  # PER_SLOT: the promise is in the cells before the steady stretch starts (setup, untimed)

    # synthetic command stream, resident in HBM
    steps = []
    for i in range(K + Wm):
        w = i % windows
        lap = i // windows  # after a lap over the window ring the slots are re-proposed in round lap
        slot = torch.arange(w * SLOTS_PER_STEP, (w + 1) * SLOTS_PER_STEP, dtype=torch.int32, device=dev)
        rnd = torch.full((SLOTS_PER_STEP,), lap, dtype=torch.int32, device=dev)
        val = steady_values_torch(slot)
        ch = torch.zeros(SLOTS_PER_STEP, dtype=torch.uint8, device=dev)
        # every step keeps its own Chosen records (round, value): all of them are verified after the timed region
        cr = torch.full((SLOTS_PER_STEP,), -7, dtype=torch.int32, device=dev)
        cv = torch.full((SLOTS_PER_STEP,), -7, dtype=torch.int32, device=dev)
        steps.append((slot, rnd, val, ch, cr, cv))
    lo, hi = 0, SLOTS_PER_STEP
    have_comm = False
    if replica_shard:
        from frankenpaxos_amd import sharding
        lo, hi = sharding.slot_slice(SLOTS_PER_STEP, world, rank)
        # the communicator under a deadline: a rank that never comes back from ncclCommInitRank must not hang the job
        res, stuck = run_with_deadline(lambda: setup_comm(fa, ctx, dist, backend, dev, rank, world), args.comm_deadline, dev)
        if stuck or isinstance(res, dict):
            sys.stderr.write("bench.py: rank %d: fpx_comm_create: %s -- giving up\n" % (rank, res.get("error") if isinstance(res, dict) else "?"))
            sys.stderr.flush()
            if rank == 0:
                print(json.dumps({"metric": METRIC, "value": None, "n_gpus": world, "error": "fpx_comm_create on rank %d: %s" %
                                  (rank, res.get("error") if isinstance(res, dict) else "?")}), flush=True)
            os._exit(2)
        have_comm = bool(res)
        if not have_comm:   # gloo test hook only: the exchange through torch.distributed on the host
            vb = torch.empty((SLOTS_PER_STEP, 4), dtype=torch.int64, device=dev)
            vb_mine = torch.empty((hi - lo, 4), dtype=torch.int64, device=dev)

    def step(i):
        slot, rnd, val, ch, cr, cv = steps[i]
        lap = i // windows
        if lap > 0 and lap % TALLY_WAYS == 0:
            # a window of the log is re-proposed once more than the proxy leader keeps tallies for:
            # garbage-collect its (chosen and executed) tallies first -- timed like everything else
            ctx.proxy_forget((i % windows) * SLOTS_PER_STEP, SLOTS_PER_STEP)
        if not replica_shard:
            ctx.phase2_fused_dev(slot, rnd, val, None, ch, cr, cv)
        elif have_comm:
            # ONE C-ABI call: K1 on my acceptors for every slot -> ncclReduceScatter(sum) of the disjoint partial
            # bitmaps over xGMI (each rank receives the full bitmaps of ITS 1/N of the slots) -> open + K2 on that slice
            ctx.phase2_replica_sharded_dev(slot, rnd, val, None, ch[lo:hi], cr[lo:hi], cv[lo:hi])
        else:
            ctx.acceptor_phase2a_dev(slot, rnd, val, None, vb, None, None)
            all_reduce(vb, dist.ReduceOp.SUM)
            vb_mine.copy_(vb[lo:hi])
            ctx.proxy_open_dev(slot[lo:hi], rnd[lo:hi], val[lo:hi])
            ctx.proxy_phase2b_dev(slot[lo:hi], rnd[lo:hi], vb_mine, ch[lo:hi], cr[lo:hi], cv[lo:hi])

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(Wm):
        step(i)
    assert ctx.sync() == 0
    ctx.profile_enable(True)
    fence()
    t0 = time.perf_counter()
    for i in range(Wm, Wm + K):
        step(i)
    fence()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    per_launch = ctx.profile_read_launches()
    launches, kernel_ms = len(per_launch), float(sum(per_launch))
    coll_n, coll_ms = ctx.profile_read_collective()
    assert ctx.sync() == 0
    hbm_bytes = ctx.device_bytes
    placement = ctx.placement_stats()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_reduce(t, dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # EVERY timed step must have committed all of its slots, in its round, with the proposed value
    committed = 0
    for i in range(Wm, Wm + K):
        slot, rnd, val, ch, cr, cv = steps[i]
        committed += int(ch.sum().item())
        assert bool(ch[lo:hi].all()), "step %d: a slot was not chosen" % i    # replica sharding: my slice of the tally
        assert bool((cv[lo:hi] == val[lo:hi]).all()), "step %d: chosen value != proposed value" % i
        assert bool((cr[lo:hi] == rnd[lo:hi]).all()), "step %d: chosen round != proposing round" % i
    assert committed == K * (hi - lo), (committed, K * (hi - lo))
    if dist is not None:
        t = torch.tensor([committed], dtype=torch.int64, device=dev)
        all_reduce(t, dist.ReduceOp.SUM)
        committed = int(t.item())

    # SURVEY.md 8e asks for both sharding rows: with N > 1 the default (group-sharded) run appends the
    # replica-axis row -- one 2^20 x 256 grid split over the N GPUs' acceptor columns, RCCL reduce-scatter of
    # the vote bitmaps behind the C ABI, collective time broken out
    replica_row, hung, replica_row_tried = None, False, False
    if world > 1 and not replica_shard and args.replica_row_steps > 0:
        replica_row_tried = True
        # the headline line must not be lost to the extra row: an exception becomes an "error" field, and a
        # collective that never returns (RCCL between real GPUs runs for the first time in the driver's own
        # multi-GPU job) is given a deadline -- the line is printed without the row and the process leaves
        # without waiting for the stuck thread
        replica_row, hung = run_with_deadline(
            lambda: replica_axis_row(fa, dist, backend, dev, rank, world, local_rank, ballot_mode,
                                     args.replica_row_steps, all_reduce),
            args.replica_row_deadline, dev)
        if rank != 0 and not hung:
            replica_row = None

    # N > 1: does RCCL carry bytes between these GPUs at all?  Asked on its own (a communicator behind the C ABI on a
    # small context + one all-gather of Chosen records), so that `rccl` in the line does not depend on the extra
    # row having finished; deadline-guarded like the row.
    rccl_info = None
    if world > 1 and (backend == "nccl" or hook("FPX_BENCH_FPX_COMM") == "1") and not hung:
        def rccl_probe():
            pctx = fa.Context(fa.make_config(num_slots=4096, num_replicas=4, f=1, device=local_rank))
            pctx.set_stream(torch.cuda.current_stream().cuda_stream)
            setup_comm(fa, pctx, dist, backend, dev, rank, world)
            nloc = 1024
            mine = [torch.full((nloc,), rank + 1, dtype=dt, device=dev) for dt in (torch.uint8, torch.int32, torch.int32)]
            alls = [torch.zeros((world * nloc,), dtype=t.dtype, device=dev) for t in mine]
            pctx.comm_allgather_chosen_dev(*mine, *alls)
            assert pctx.sync() == 0
            want = torch.arange(1, world + 1, device=dev).repeat_interleave(nloc)
            ok = all(bool((a.to(torch.int64) == want).all()) for a in alls)
            r, w2 = pctx.comm_info()
            pctx.close()
            return {"ranks": w2, "allgather_of_chosen_records_ok": ok}
        rccl_info, hung2 = run_with_deadline(rccl_probe, 90, dev)
        hung = hung or hung2

    # N = 1: BASELINE.json's other configs, a few steps each, so that the driver's own run times them too
    configs_block = None
    if world == 1 and args.configs_block_steps > 0 and not replica_shard:
        import types
        import bench_configs
        ctx.close()
        torch.cuda.empty_cache()
        configs_block = {"doc": "bench_configs.md (per entry: workload, kernels, byte model, what `verified` checked)"}
        # BASELINE.json's configs 2-5, the two thrifty deliveries, and what SURVEY.md 8(d) asks for beside the headline: the
        # reference's actual acceptor model, the host-pointer path end to end, the adversarial stream at full size
        for c in ("2", "3", "4", "4_execute", "5", "thrifty", "thrifty_random", "acceptor_model", "host_path", "adversarial"):
            t_c = time.perf_counter()
            try:
                # config 4's ticks are drawn on the host (~0.5 s each): half as many of them
                # configs 2 and 3 get ten times the steps: 20 steps of 12 - 40 us are a timed region of a quarter of a
                # millisecond, which measures the GPU waking up after the fence, not the step -- config 2: 0.018 - 0.032 ms
                # per step over 20 steps, 0.0121 - 0.0129 over 200 (profiles/r05_small_steps.md)
                # (config 5's 85 us bands likewise: twice the steps -- its 4M-slot bands are 2 GB of state each; config 4's ticks are drawn on the host, 0.5 s each)
                # (host_path: the full count -- three calls in flight fill and drain once per region, 5 % of a 10-call region)
                sub_steps = (max(1, args.configs_block_steps // 2) if c in ("4_execute", "thrifty_random") else
                             max(1, 3 * args.configs_block_steps // 4) if c == "4" else
                             10 * args.configs_block_steps if c in ("2", "3") else
                             2 * args.configs_block_steps if c == "5" else args.configs_block_steps)
                sub = types.SimpleNamespace(steps=sub_steps, warmup=2, ballot=args.ballot, config=c, no_cpu_baseline=True)
                full = bench_configs.run(sub, fa, None, dev, 0, 1, local_rank, all_reduce)
                configs_block[c] = compact_entry(full)
                if c == "2" and args.ballot == "per_slot":
                    # BASELINE.json configs[1] is the reference's own f = 1 deployment: its acceptors keep ONE round each
                    # (multipaxos/Acceptor.scala:95) -- the same steps under FPX_BALLOT_ACCEPTOR, beside the
                    # ballot-per-cell figures this block reports for continuity with the headline's model
                    sub2 = types.SimpleNamespace(steps=sub_steps, warmup=2, ballot="acceptor", config="2", no_cpu_baseline=True)
                    configs_block["2_acceptor_model"] = compact_entry(bench_configs.run(sub2, fa, None, dev, 0, 1, local_rank, all_reduce))
            except BaseException as e:  # noqa: BLE001 -- the headline line must survive a failing extra config
                configs_block[c] = {"error": ("%s: %s" % (type(e).__name__, e))[:300]}
            configs_block[c]["wall_s"] = round(time.perf_counter() - t_c, 1)
            torch.cuda.empty_cache()
        # the oracle alone behind a FIFO message pump, on the host (no GPU involved)
        try:
            configs_block["1"] = bench_configs.config1_entry()
        except BaseException as e:  # noqa: BLE001
            configs_block["1"] = {"error": ("%s: %s" % (type(e).__name__, e))[:300]}

    if rank == 0:
        bps = algorithmic_bytes_per_slot(ballot_mode)
        # steps beyond the first lap over the window ring re-propose old slots in a higher round: in the
        # PER_SLOT model that also rewrites the ballot row (+4 R bytes per slot), which is algorithmic
        wrapped = sum(1 for i in range(Wm, Wm + K) if i // windows > 0)
        if ballot_mode == 1 and wrapped:
            bps = (bps * (K - wrapped) + (bps + 4 * REPLICAS) * wrapped) / K
        avg_kernel_s = (kernel_ms / max(launches, 1)) * 1e-3
        slots_per_launch = SLOTS_PER_STEP
        achieved = bps * slots_per_launch / avg_kernel_s / 1e9 if launches else None
        traffic, traffic_round = None, None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                te = json.load(open(tfile)).get(args.ballot) or {}
                traffic, traffic_round = te.get("bytes"), te.get("round")
            except Exception:
                traffic = None
        kernel = "k_phase2<64,vec4,%s,%s>" % ("per_slot" if ballot_mode == 1 else "acceptor",
                                               "K1" if replica_shard else "fused")
        if ballot_mode == 1 and not replica_shard and not os.environ.get("FPX_NO_DEFER_FINALIZE"):
            # (round 6: the fold of the step before rides as the first workgroups of the vote kernel's launch)
            kernel = "k_phase2_fin<64,vec4,per_slot,fused> (k_phase2 + the fold of the step before in its grid)"
        line = {
            "metric": METRIC,
            "value": committed / elapsed,
            "unit": "slots/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if replica_shard else "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": "MultiPaxos Phase-2 fused step (open + 256 acceptor votes + f+1=128 tally), "
                            "steady stream, 2^20 fresh slots x 256 acceptors per step per GPU",
                "slots_per_step": SLOTS_PER_STEP, "replicas": REPLICAS, "quorum": F + 1,
                "ballot_model": args.ballot, "sharding": args.shard if world > 1 else "none",
                "run_contract_validation_in_timed_region": bool(args.validate),
                "log_windows_in_hbm": windows, "hbm_bytes": hbm_bytes,
                # how fpx_create placed the cell arrays (profiles/r05_placement.md): chunks paired by measurement, and the hot
                # access pattern's time on half a window, min / median / max over the windows
                "placement": {"chunks_paired_by_measurement": placement["chunks"], "windows": placement["windows"],
                              "probe_ms_min_median_max": list(placement["probe_ms"]),
                              "search_ms": placement["search"]["ms"], "probes": placement["search"]["probes"],
                              "unprobed_decisions": placement["search"]["unprobed_decisions"]},
                "steps_reproposing_old_slots": sum(1 for i in range(Wm, Wm + K) if i // windows > 0),
            },
            "roofline": {
                "bound": "hbm", "kernel": kernel,
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                # SURVEY.md 8(d) asks for both accountings: `achieved` counts every algorithmic byte (reads + writes,
                # 3088 B per slot in the PER_SLOT model); achieved_read only the reads (ballot row + proposal, 1032 B
                # per slot; 8 B in the ACCEPTOR model) -- two thirds of the bytes are writes, so the read-only figure is
                # bounded at a third of what the kernel moves
                "achieved_total": achieved,
                "achieved_read": (read_bytes_per_slot(ballot_mode) * slots_per_launch / avg_kernel_s / 1e9) if launches else None,
                "frac_read": (read_bytes_per_slot(ballot_mode) * slots_per_launch / avg_kernel_s / 1e9 / HBM_PEAK_GBS) if launches else None,
                "traffic": traffic, "traffic_round": traffic_round,
                "traffic_source": "profiles/traffic.json: rocprofv3 --pmc passes of this command in an earlier profiled run",
                "algorithmic_bytes_per_slot": bps, "slots_per_launch": slots_per_launch,
                "avg_kernel_ms": kernel_ms / max(launches, 1), "launches_timed": launches,
                "kernel_ms_min_max_sigma": ([min(per_launch), max(per_launch), float(np.std(per_launch))] if per_launch else None),
                # SURVEY.md 8(d)'s read-only accounting against north_star's ">= 40 % HBM-read roofline": the byte model has 2056 B
                # written per 1032 B read, so even a kernel that moved its bytes at the full 8 TB/s would read at 1032 / 3088 of
                # it -- frac_read cannot exceed 0.334 on this workload
                "frac_read_cap": read_bytes_per_slot(ballot_mode) / algorithmic_bytes_per_slot(ballot_mode),
                "kernel_time_source": "HIP events on the vote kernel's dispatch packet, every timed launch (fpx_profile_*)",
                # what bare streaming kernels reach on this chip (profiles/microbench/hbm_mix.hip, best of the
                # grid / unroll sweep in profiles/r01_hbm_mix.txt): the practical ceiling beside the spec peak
                "measured_stream_GBs": MEASURED_STREAM_GBS,
                "frac_of_measured_read_stream": (achieved / MEASURED_STREAM_GBS["read"]) if achieved else None,
            },
        }
        # how many ranks an RCCL communicator created in THIS run actually spans (0 at N = 1: none is needed)
        if TEST_HOOKS:
            line["test_hooks"] = HOOKS_SET
        line["rccl_ranks"] = world if (have_comm or (replica_row or {}).get("rccl_ranks") or
                                       (rccl_info or {}).get("ranks") == world) else 0
        if rccl_info is not None:
            line["rccl"] = rccl_info
        if configs_block is not None:
            line["configs"] = configs_block
        if replica_shard:
            line["collective"] = {
                "op": "ncclReduceScatter(ncclSum, ncclUint64) via fpx_phase2_replica_sharded_dev" if have_comm
                      else "all_reduce over gloo (single-GPU test hook)",
                "calls_timed": coll_n, "avg_ms": (coll_ms / coll_n) if coll_n else None,
                "xgmi_bytes_per_gpu_per_step": (world - 1) * 32 * SLOTS_PER_STEP // world,
            }
        if replica_row is not None:
            line["replica_axis"] = replica_row
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ballot_mode, light=world > 1)
        print(json.dumps(round_floats(line)), flush=True)
    if hung:  # a collective of the extra row never returned: no barrier, no teardown that could wait for it
        sys.stderr.write("bench.py: rank %d: the replica-axis row did not finish; exiting without it\n" % rank)
        sys.stderr.flush()
        os._exit(0)
    if dist is not None:
        if replica_row_tried:
            # another rank may have left without the row (see above): the closing barrier gets a deadline too
            _, stuck = run_with_deadline(lambda: (dist.barrier(), dist.destroy_process_group()), 60, dev)
            if stuck:
                os._exit(0)
        else:
            dist.barrier()
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
