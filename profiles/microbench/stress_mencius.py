"""Randomised MultiPaxos / Mencius stress beside the oracle (python profiles/microbench/stress_mencius.py on a GPU box):
random geometries (leader groups, acceptor groups, R, windows that are and are not a multiple of L, both row layouts,
both ballot models), batches of 1 .. 6000 messages in slot order / grouped by leader group / shuffled, with and without
target masks, fused and unfused, noop ranges, Phase1as, leader changes, garbage collection and recycling in between.
Every output of every call, the final state, digests and sampled tallies are compared.  Prints the mismatch count."""
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import frankenpaxos_amd as fa  # noqa: E402
import oracle.pyoracle as oracle  # noqa: E402
from tests import workloads as W  # noqa: E402

oracle.build()
seeds = int(os.environ.get("SEEDS", "40"))
bad = 0
for seed in range(seeds):
    rng = np.random.default_rng(7000 + seed)
    L = int(rng.choice([1, 2, 3, 8, 16, 64, 256]))
    A = int(rng.choice([1, 1, 2, 3]))
    R = int(rng.choice([3, 3, 4, 5, 8, 16]))
    f = (R - 1) // 2
    rows = int(rng.integers(64, 3000))
    S = L * rows + (int(rng.integers(1, 5)) if seed % 5 == 4 else 0)
    ranges = L > 1 and seed % 3 != 2
    mode = 0 if ranges or seed % 2 else 1
    flags = fa.FPX_F_SLOT_MAJOR_ROWS if seed % 4 == 3 else 0
    kw = dict(num_slots=S, num_replicas=R, num_groups=A, num_leader_groups=L, f=f, tally_ways=8, ballot_mode=mode)
    try:
        gpu = fa.Context(fa.make_config(flags=flags, **kw))
    except fa.FpxError:      # a geometry the library refuses (num_groups * num_leader_groups * R > 8192)
        continue
    ref = oracle.System(oracle.make_config(**kw))
    rounds = [0] * L
    desc = "seed %d: L %d A %d R %d S %d mode %d flags %d" % (seed, L, A, R, S, mode, flags)
    try:
        for step in range(14):
            kind = int(rng.integers(0, 8))
            if kind == 0:      # leader changes
                for lg in rng.integers(0, L, 3):
                    rounds[int(lg)] += int(rng.integers(1, 3))
            if kind == 1:      # a competing leader's Phase1a on some acceptors of some groups
                for _ in range(2):
                    lg = int(rng.integers(0, L))
                    g = lg * A + int(rng.integers(0, A))
                    t = W.bits_from_bool(W.random_subsets(rng, 1, R, 1, R))[0]
                    script = [("phase1a", g, rounds[lg] + 1, int(rng.integers(0, S)), t)]
                    W.assert_same_outputs(W.run_script(gpu, script), W.run_script(ref, script))
            n = int(rng.integers(1, min(S, 6000) + 1))
            slots = rng.choice(S, size=n, replace=False).astype(np.int32)
            order = int(rng.integers(0, 3))
            if order == 0:
                slots = np.sort(slots)
            elif order == 1:   # the leader groups' batches back to back, each in slot order
                slots = slots[np.lexsort((slots, slots % L))]
            rr = np.array([rounds[int(s) % L] for s in slots], np.int32)
            val = (slots * 5 + step).astype(np.int32)
            tm = None if rng.random() < 0.4 else W.bits_from_bool(W.random_subsets(rng, n, R, 1, R))
            if kind in (2, 3) and not (kind == 3 and mode == 1):
                dup = rng.random(n) < 0.1
                script = [("k1k2", slots, rr, val, tm, dup)]
            else:
                script = [("fused", slots, rr, val, tm)]
            W.assert_same_outputs(W.run_script(gpu, script), W.run_script(ref, script))
            if ranges and kind in (4, 5, 6):
                # the range table holds 2048 live entries here (fpx_create sizes it by the leader groups): big launches
                # stay below that and the window is garbage-collected right after them
                k = int(rng.integers(1, 300)) if kind != 6 else int(rng.integers(600, 1900))
                if k >= 600:
                    gpu.proxy_forget(0, S), ref.proxy_forget(0, S)
                lgs = rng.integers(0, L, k)
                a = rng.integers(0, rows, k)
                b = np.minimum(rows, a + rng.integers(1, 400, k))
                start = (a * L + lgs).astype(np.int32)
                end = np.minimum(S, (b - 1) * L + lgs + 1).astype(np.int32)
                rnd = np.array([rounds[int(x)] for x in lgs], np.int32)
                tmr = None if kind == 4 else W.bits_from_bool(W.random_subsets(rng, k * A, R, 1, R)).reshape(k, A, 4)
                x, y = gpu.noop_ranges_fused(start, end, rnd, tmr), ref.noop_ranges_fused(start, end, rnd, tmr)
                assert x[0] == y[0], "ranges status"
                for u, v in zip(x[1:], y[1:]):
                    np.testing.assert_array_equal(np.asarray(u), np.asarray(v), err_msg="ranges")
                if k >= 600:
                    gpu.proxy_forget(0, S), ref.proxy_forget(0, S)
            if kind == 7:
                lo = int(rng.integers(0, S // 2))
                cnt = int(rng.integers(1, S // 3 + 2))
                gpu.proxy_forget(lo, cnt), ref.proxy_forget(lo, cnt)
                lo2 = int(rng.integers(0, S // 2))
                gpu.recycle_slots(lo2, cnt), ref.recycle_slots(lo2, cnt)
        W.assert_same_state(gpu, ref, tally_slots=rng.integers(0, S, 40))
        np.testing.assert_array_equal(gpu.state_digest(), ref.state_digest())
    except AssertionError as e:
        bad += 1
        print("MISMATCH", desc, "::", str(e).strip().splitlines()[0][:200])
    gpu.close() if hasattr(gpu, "close") else None
print("stress done: %d geometries, mismatches: %d" % (seeds, bad))
