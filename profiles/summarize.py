#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of profiles/collect.sh (gpurun_out/prof_<tag>/) into the committed
summaries: profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_summary.md, profiles/traffic.json."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(root, "gpurun_out", "prof_" + tag)
out = os.path.join(root, "profiles")


def agg(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


def pick(d, needle):
    # (round 6: with a ballot per cell the steady launches are k_phase2_fin -- the vote kernel with the fold of the step
    # before as its first workgroups; the first launch of a stream, with nothing to fold, is a plain k_phase2)
    for n in ([needle.replace("k_phase2", "k_phase2_fin")] if needle.startswith("k_phase2") else []) + [needle]:
        for (k, c), v in d.items():
            if n in k:
                return v
    return float("nan")


shutil.copy(os.path.join(base, "trace", "bench_kernel_stats.csv"), os.path.join(out, tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(base, "trace_bench.json"), os.path.join(out, tag + "_bench_under_rocprof.json"))
cal_f = agg(os.path.join(base, "pmc_FETCH_SIZE", "calib_counter_collection.csv"))
cal_w = agg(os.path.join(base, "pmc_WRITE_SIZE", "calib_counter_collection.csv"))
GiB_KiB = 1 << 20
f_read = pick(cal_f, "k_mix<1, 0, false, 4>")
w_fill = pick(cal_w, "k_mix<0, 1, false, 4>")
f_mix, w_mix = pick(cal_f, "k_mix<1, 2, true, 4>"), pick(cal_w, "k_mix<1, 2, true, 4>")
fetch_scale = GiB_KiB / f_read  # KiB of HBM read per FETCH_SIZE unit (expected 2.0)
write_scale = GiB_KiB / w_fill  # KiB of HBM written per WRITE_SIZE unit (expected 1.0)
rows, traffic = [], {}
for f, label, alg in (("bench", "per_slot", 3088), ("bench_acceptor", "acceptor", 2064)):
    fe = pick(agg(os.path.join(base, "pmc_FETCH_SIZE", f + "_counter_collection.csv")), "k_phase2")
    wr = pick(agg(os.path.join(base, "pmc_WRITE_SIZE", f + "_counter_collection.csv")), "k_phase2")
    rd_b, wr_b = fe * fetch_scale * 1024, wr * write_scale * 1024
    traffic[label] = {"bytes": rd_b + wr_b, "round": tag, "source": "profiles/%s_pmc_summary.md" % tag}
    rows.append((label, fe, wr, rd_b, wr_b, rd_b + wr_b, alg * (1 << 20)))
try:  # keep the entries other scripts maintain (config4: profiles/microbench/k5_pmc.sh)
    kept = json.load(open(os.path.join(out, "traffic.json")))
except Exception:
    kept = {}
kept.update(traffic)
json.dump(kept, open(os.path.join(out, "traffic.json"), "w"), indent=1)
stats = list(csv.DictReader(open(os.path.join(out, tag + "_kernel_stats.csv"))))
k2 = ([r for r in stats if "k_phase2_fin<64" in r["Name"]] + [r for r in stats if "k_phase2" in r["Name"]])[0]
bench = json.load(open(os.path.join(base, "trace_bench.json")))
with open(os.path.join(out, tag + "_pmc_summary.md"), "w") as f:
    f.write("# %s rocprofv3 summary\n\n" % tag)
    f.write("Collected by `bash profiles/collect.sh %s` on the MI355X box, summarised by "
            "`python profiles/summarize.py %s`.\n\n" % (tag, tag))
    f.write("## Kernel trace (`rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline`)\n\n")
    f.write("`%s`: %s calls, average **%.1f us** (min %.1f, max %.1f). HIP-event average measured live by "
            "bench.py in the same run: **%.1f us** (%d timed launches).\nFull table: `%s_kernel_stats.csv`.\n\n"
            % (k2["Name"].split("(")[0], k2["Calls"], float(k2["AverageNs"]) / 1e3, float(k2["MinNs"]) / 1e3,
               float(k2["MaxNs"]) / 1e3, bench["roofline"]["avg_kernel_ms"] * 1e3,
               bench["roofline"]["launches_timed"], tag))
    # the same kernel instantiation is also launched by the line's `configs` block (configs.host_path: beside staging kernels on
    # other streams, so slower): the headline's launches are the first warmup + steps of the trace, in order
    tr = os.path.join(base, "trace", "bench_kernel_trace.csv")
    if os.path.exists(tr):
        name = k2["Name"].split("(")[0].replace("void ", "")
        rows_ = sorted((r for r in csv.DictReader(open(tr)) if name in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows_]
        W, K = bench["warmup"], bench["steps"]
        t = d[W:W + K]
        if len(t) == K:
            mean = sum(t) / K
            sd = (sum((x - mean) ** 2 for x in t) / K) ** 0.5
            f.write("Of those calls the first %d are the headline's (%d warm-up + %d timed, in trace order); the TIMED %d: average "
                    "**%.1f us** (min %.1f, max %.1f, sigma %.1f = %.2f %%) -- the figure that has to agree with the HIP-event "
                    "average above.  The other %d calls belong to the line's `configs` block (`host_path` runs this kernel beside "
                    "its staging kernels: average %.1f us).\n\n"
                    % (W + K, W, K, K, mean, min(t), max(t), sd, 100 * sd / mean, len(d) - W - K,
                       (sum(d[W + K:]) / max(1, len(d) - W - K))))
    f.write("## Counter calibration on known byte counts (profiles/microbench/hbm_mix.hip, 1 GiB per stream)\n\n")
    f.write("| kernel | KiB read | FETCH_SIZE | KiB written | WRITE_SIZE |\n|---|---|---|---|---|\n")
    f.write("| k_mix<1,0> read | 1,048,576 | %.0f | 0 | %.0f |\n" % (f_read, pick(cal_w, "k_mix<1, 0, false, 4>")))
    f.write("| k_mix<0,1> fill | 0 | %.0f | 1,048,576 | %.0f |\n" % (pick(cal_f, "k_mix<0, 1, false, 4>"), w_fill))
    f.write("| k_mix<1,2> nt  | 1,048,576 | %.0f | 2,097,152 | %.0f |\n\n" % (f_mix, w_mix))
    f.write("=> HBM read bytes = FETCH_SIZE x %.3f x 1024 (the 1/2 factor of MI355X_MICROARCH.md \"HBM\"), "
            "HBM written bytes = WRITE_SIZE x %.3f x 1024.\n\n" % (fetch_scale, write_scale))
    f.write("## k_phase2 (fused K3, G = 64, vec4), per launch of 2^20 slots x 256 acceptors\n\n")
    f.write("| ballot model | FETCH_SIZE | WRITE_SIZE | HBM read B | HBM written B | traffic B | algorithmic B | traffic / algorithmic |\n")
    f.write("|---|---|---|---|---|---|---|---|\n")
    for label, fe, wr, rd_b, wr_b, tot, alg in rows:
        f.write("| %s | %.1f | %.1f | %.0f | %.0f | %.0f | %d | %.4f |\n" % (label, fe, wr, rd_b, wr_b, tot, alg, tot / alg))
    f.write("\nThe ~2 % above the SURVEY.md 8(d) byte model is the proxy leader's tally table (a 16 B key row "
            "read + 4 B key written per slot), the 12 B proposal instead of 8 B (the slot index is read too), "
            "the 1-byte chosen flag (the whole-group maxima are two atomics per workgroup).  No row is read twice.\n")
print(open(os.path.join(out, tag + "_pmc_summary.md")).read())
