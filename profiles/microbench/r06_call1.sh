# r06, first GPU call: measurements on the tree as round 5 left it (no code change yet)
#   1. which TCC / TCP counters this rocprofv3 knows
#   2. config 5: write requests by size, L2 hits / misses, TCP stalls per kernel
#   3. host path: kernel + memory-copy timeline of calls in flight
#   4. PMC traffic of the configs whose `traffic` was null
R=$PWD; O=$R/gpurun_out/r06c1; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "(TCC|TCP)_[A-Z0-9_]+" | sort -u > $O/counters_tcc_tcp.txt
wc -l $O/counters_tcc_tcp.txt
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum" "TCC_EA0_WR_UNCACHED_32B_sum TCC_WRITEBACK_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/c5_$i -o k -- python $R/bench.py --config 5 --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2> $O/cfg5_err_$i.txt
  find /tmp/c5_$i -name "*counter_collection.csv" -exec cp {} $O/cfg5_set_$i.csv \;
done
cd $R; python - <<'PY'
import csv, collections, glob
tab = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob("gpurun_out/r06c1/cfg5_set_*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fpx" not in k and "k_" not in k: continue
        k = k.split("(")[0][-60:]
        e = tab[k][r["Counter_Name"]]; e[0] += float(r["Counter_Value"]); e[1] += 1
with open("gpurun_out/r06c1/cfg5_counters.txt", "w") as out:
    for k, d in tab.items():
        print(k, file=out)
        for c, (v, n) in d.items(): print("   %-32s %14.0f per launch (%d launches)" % (c, v / n, n), file=out)
print(open("gpurun_out/r06c1/cfg5_counters.txt").read())
PY
cd /tmp
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/hp_tl -o k -- python $R/bench.py --config host_path --no-cpu-baseline --steps 8 --warmup 2 > $O/host_path_line.json 2> $O/host_path_err.txt
find /tmp/hp_tl -name "*kernel_trace.csv" -exec cp {} $O/host_path_kernel_trace.csv \;
find /tmp/hp_tl -name "*memory_copy_trace.csv" -exec cp {} $O/host_path_memcpy_trace.csv \;
cd $R
python profiles/microbench/timeline.py $O/host_path_kernel_trace.csv 60 > $O/host_path_timeline.txt 2>&1
tail -45 $O/host_path_timeline.txt
for N in thrifty 4_execute 2 3 host_path; do bash profiles/microbench/cfg_pmc.sh $N > $O/pmc_$N.md 2>&1; mkdir -p $O/pmc_raw_$N; cp gpurun_out/cfg${N}pmc/*.csv $O/pmc_raw_$N/ 2>/dev/null; done
head -12 $O/pmc_*.md
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
