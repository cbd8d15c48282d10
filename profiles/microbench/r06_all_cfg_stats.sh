# r06, the round's last build: rocprofv3 --kernel-trace --stats of every bench.py --config, one CSV each
R=$PWD; O=$R/gpurun_out/r06stats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for N in 2 3 4 4_execute 5 thrifty thrifty_random acceptor_model host_path adversarial; do
  rm -rf /tmp/st4_$N
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st4_$N -o k -- python $R/bench.py --config $N --no-cpu-baseline --steps 20 --warmup 2 > $O/cfg${N}_bench_under_rocprof.json 2> /dev/null
  find /tmp/st4_$N -name "*kernel_stats.csv" -exec cp {} $O/cfg${N}_kernel_stats.csv \;
  echo "$N: $(head -c 300 $O/cfg${N}_bench_under_rocprof.json | python -c "import sys,re; s=sys.stdin.read(); m=re.search(r'\"value\": ([0-9.e+]+)', s); print(m.group(1) if m else 'no line')")"
done
