"""per-kernel HBM bytes of the K5 tick from the two PMC passes of k5_pmc.sh (gpurun_out/k5pmc/{FETCH,WRITE}_SIZE.csv):
bytes = FETCH_SIZE x 2 x 1024 and WRITE_SIZE x 1024 (the calibration of profiles/r02_pmc_summary.md), averaged per tick"""
import csv, collections, sys
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 5
tab = collections.defaultdict(lambda: [0.0, 0.0, 0])
for which, col, scale in (("FETCH_SIZE", 0, 2 * 1024.0), ("WRITE_SIZE", 1, 1024.0)):
    for r in csv.DictReader(open("gpurun_out/k5pmc/%s.csv" % which)):
        name = r["Kernel_Name"]
        if "k_kp" not in name and "k_epx" not in name and "k_rs" not in name:
            continue
        import re
        short = re.search(r"(k_[a-z0-9_]+(<\d+>)?)", name).group(1)
        tab[short][col] += float(r["Counter_Value"]) * scale
        if col == 0:
            tab[short][2] += 1
tot = [0.0, 0.0]
print("| kernel | calls | HBM read MB / tick | HBM written MB / tick |\n|---|---|---|---|")
for k, (rd, wr, n) in sorted(tab.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
    print("| `%s` | %d | %.1f | %.1f |" % (k, n, rd / ticks / 1e6, wr / ticks / 1e6))
    tot[0] += rd / ticks
    tot[1] += wr / ticks
print("\nper tick: %.1f MB read + %.1f MB written = %.1f MB = %.0f B per command" % (tot[0] / 1e6, tot[1] / 1e6, (tot[0] + tot[1]) / 1e6, (tot[0] + tot[1]) / (1 << 20)))
