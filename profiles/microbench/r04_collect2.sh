# round 4, second collection: the device dependency graph (timing + kernel stats) and the thrifty config's PMC traffic
R=$PWD; mkdir -p gpurun_out/r04c2
python profiles/microbench/depgraph_dev_bench.py 20 2>&1 | grep "commands" | tee gpurun_out/r04c2/depgraph_dev.txt
python profiles/microbench/depgraph_dev_bench.py 18 2>&1 | grep "commands" | tee -a gpurun_out/r04c2/depgraph_dev.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dgprof -o d -- python $R/profiles/microbench/depgraph_dev_bench.py 20 > /dev/null 2>&1; find /tmp/dgprof -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r04c2/depgraph_dev_kernel_stats.csv \;)
grep -E "k_dg|k_kp|k_epx|radix|sort" gpurun_out/r04c2/depgraph_dev_kernel_stats.csv | cut -d, -f1-4 | cut -c1-140 | head -30
for b in per_slot acceptor; do
  mkdir -p gpurun_out/thr_$b
  (cd /tmp && export TMPDIR=/tmp && for c in FETCH_SIZE WRITE_SIZE; do
    timeout 250 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/thrpmc_${b}_$c -o k -- python $R/bench.py --config thrifty --ballot $b --no-cpu-baseline --steps 4 --warmup 1 > $R/gpurun_out/thr_$b/bench_$c.json 2> $R/gpurun_out/thr_$b/err_$c.txt
    find /tmp/thrpmc_${b}_$c -name "*counter_collection.csv" -exec cp {} $R/gpurun_out/thr_$b/$c.csv \;
  done)
  python - <<PY
import csv, collections, re
tab = collections.defaultdict(lambda: [0.0, 0.0, 0])
for which, col, scale in (("FETCH_SIZE", 0, 2 * 1024.0), ("WRITE_SIZE", 1, 1024.0)):
    for r in csv.DictReader(open("gpurun_out/thr_$b/%s.csv" % which)):
        m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", r["Kernel_Name"])
        if not m or "fpx" not in r["Kernel_Name"]: continue
        tab[m.group(1)][col] += float(r["Counter_Value"]) * scale
        if col == 0: tab[m.group(1)][2] += 1
print("== thrifty $b\n| kernel | calls | HBM read MB / call | HBM written MB / call |\n|---|---|---|---|")
for k, (rd, wr, n) in sorted(tab.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
    print("| \`%s\` | %d | %.2f | %.2f |" % (k, n, rd / max(n, 1) / 1e6, wr / max(n, 1) / 1e6))
PY
done
for b in per_slot acceptor; do python bench.py --config thrifty --ballot $b --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python -c 'import sys,json; l=json.loads(sys.stdin.read()); print("thrifty", sys.argv[1], l["value"], l["ms_per_step"], json.dumps(l["roofline"])[:300])' $b; done
