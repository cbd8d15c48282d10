mkdir -p gpurun_out/epx; R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o epx -- python $R/bench.py --config 4 --no-cpu-baseline --steps 10 --warmup 2 > $R/gpurun_out/epx/bench.json 2> $R/gpurun_out/epx/err.txt
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/epx/ \;
cd $R; timeout 200 python bench.py --config 4 --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/epx/bench_plain.json 2>/dev/null
