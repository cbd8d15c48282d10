# rocprofv3 kernel stats of bench.py --config N (N = $1), summaries under gpurun_out/cfg$1/
N=${1:-5}
mkdir -p gpurun_out/cfg$N; R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$N -o c -- python $R/bench.py --config $N --no-cpu-baseline --steps 10 --warmup 2 > $R/gpurun_out/cfg$N/bench.json 2> $R/gpurun_out/cfg$N/err.txt
find /tmp/prof$N -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/cfg$N/ \;
