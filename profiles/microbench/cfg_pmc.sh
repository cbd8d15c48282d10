# PMC traffic of bench.py --config N ($1): FETCH_SIZE and WRITE_SIZE in separate runs, kernel trace only
N=${1:-5}; mkdir -p gpurun_out/cfg${N}pmc; R=$PWD; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 250 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/cfgpmc_$c -o k -- python $R/bench.py --config $N --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2> $R/gpurun_out/cfg${N}pmc/err_$c.txt
  find /tmp/cfgpmc_$c -name "*counter_collection.csv" -exec cp {} $R/gpurun_out/cfg${N}pmc/$c.csv \;
done
cd $R; python - <<PY
import csv, collections, re
tab = collections.defaultdict(lambda: [0.0, 0.0, 0])
for which, col, scale in (("FETCH_SIZE", 0, 2 * 1024.0), ("WRITE_SIZE", 1, 1024.0)):
    for r in csv.DictReader(open("gpurun_out/cfg${N}pmc/%s.csv" % which)):
        m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", r["Kernel_Name"])
        if not m or "rocclr" in r["Kernel_Name"] or "at::" in r["Kernel_Name"]: continue  # (the EPaxos kernels live in an anonymous namespace)
        tab[m.group(1)][col] += float(r["Counter_Value"]) * scale
        if col == 0: tab[m.group(1)][2] += 1
print("| kernel | calls | HBM read MB / call | HBM written MB / call |\n|---|---|---|---|")
for k, (rd, wr, n) in sorted(tab.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
    print("| \`%s\` | %d | %.2f | %.2f |" % (k, n, rd / n / 1e6, wr / n / 1e6))
PY
