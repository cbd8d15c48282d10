# rocprofv3 kernel stats of the default bench command (the headline) + the bench line of that same profiled run
mkdir -p gpurun_out/headline; R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profh -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/headline/bench_under_rocprof.json 2> $R/gpurun_out/headline/err.txt
find /tmp/profh -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/headline/ \;
