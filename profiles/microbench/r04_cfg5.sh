# config 5 after the LDS chain: tests, bench, kernel stats and the timeline of the last steps
mkdir -p gpurun_out/cfg5; R=$PWD
python -m pytest tests/test_mencius_noop_range.py tests/test_mencius_models.py tests/test_mencius_safety.py -x -q -m gpu 2>&1 | tail -4
python bench.py --config 5 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5 -o c -- python $R/bench.py --config 5 --no-cpu-baseline --steps 10 --warmup 2 > $R/gpurun_out/cfg5/bench.json 2> $R/gpurun_out/cfg5/err.txt
find /tmp/prof5 -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/cfg5/ \;
find /tmp/prof5 -name "*kernel_trace.csv" -exec cp {} $R/gpurun_out/cfg5/trace.csv \;
cd $R
grep "fpx::" gpurun_out/cfg5/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
python profiles/microbench/timeline.py gpurun_out/cfg5/trace.csv 400 | grep -v "at::native\|rocprim\|rocclr" | tail -30
python profiles/microbench/timeline.py gpurun_out/cfg5/trace.csv 60 | tail -30 | cut -c1-110
