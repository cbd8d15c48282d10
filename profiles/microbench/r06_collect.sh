# r06: the round's evidence on the final build -- bash profiles/collect.sh r06 (headline: kernel trace + stats, PMC passes),
# per-kernel stats of configs 4 / 5 / 4_execute / host_path, PMC traffic of 4_execute and host_path in their new forms
R=$PWD; O=$R/gpurun_out/r06col; mkdir -p $O
bash profiles/collect.sh r06 > $O/collect.log 2>&1
for N in 4_execute host_path; do bash profiles/microbench/cfg_pmc.sh $N > $O/pmc_$N.md 2>&1; mkdir -p $O/pmc_raw_$N; cp gpurun_out/cfg${N}pmc/*.csv $O/pmc_raw_$N/ 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
for N in 4 5 4_execute host_path; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$N -o k -- python $R/bench.py --config $N --no-cpu-baseline --steps 10 --warmup 2 > $O/cfg${N}_bench_under_rocprof.json 2> /dev/null
  find /tmp/st_$N -name "*kernel_stats.csv" -exec cp {} $O/cfg${N}_kernel_stats.csv \;
done
cd $R
ls $O gpurun_out/prof_r06 | head -40
