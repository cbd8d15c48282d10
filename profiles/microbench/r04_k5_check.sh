set -u
mkdir -p gpurun_out/r4a
( timeout 900 python -m pytest tests/test_epaxos.py tests/test_epaxos_models.py -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r4a/epx_tests.txt
( timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config4 or epaxos or cfg4" 2>&1 | tail -8 ) >> gpurun_out/r4a/epx_tests.txt
K5_MODES=packed timeout 300 python profiles/microbench/k5v2_time.py new 2>&1 | grep -v amdgpu.ids > gpurun_out/r4a/k5_time.txt
K5_MODES=packed K5_READY=1 timeout 300 python profiles/microbench/k5v2_time.py new_ready 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4a/k5_time.txt
K5_N=3 K5_MODES=packed K5_READY=1 timeout 300 python profiles/microbench/k5v2_time.py new_n3_ready 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4a/k5_time.txt
timeout 300 python bench.py --config 4 --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r4a/bench_cfg4_plain.json 2> gpurun_out/r4a/bench_cfg4_plain.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k5prof -o k5 -- python $GRAFT_REPO_ROOT/bench.py --config 4 --no-cpu-baseline --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r4a/bench_cfg4.json 2> $GRAFT_REPO_ROOT/gpurun_out/r4a/bench_cfg4.err
find /tmp/k5prof -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r4a/cfg4_kernel_stats.csv \;
cd $GRAFT_REPO_ROOT; cat gpurun_out/r4a/epx_tests.txt gpurun_out/r4a/k5_time.txt; grep -E "k_epx|k_kp" gpurun_out/r4a/cfg4_kernel_stats.csv | cut -c1-160; cut -c1-330 gpurun_out/r4a/bench_cfg4_plain.json
