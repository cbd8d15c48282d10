mkdir -p gpurun_out/r03d
timeout 900 python -m pytest tests/test_epaxos.py -x -q -m gpu > gpurun_out/r03d/gputest.log 2>&1; echo rc=$?; tail -25 gpurun_out/r03d/gputest.log
timeout 300 python bench.py --config 4 --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r03d/bench_cfg4.json 2> gpurun_out/r03d/bench_cfg4.err; echo bench rc=$?; cat gpurun_out/r03d/bench_cfg4.json | cut -c1-600
