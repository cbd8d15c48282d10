set -u; O=gpurun_out/r05c9; mkdir -p $O
timeout 900 python -m pytest tests/test_jni_shim.py tests/test_depgraph_dev.py tests/test_wire_dev.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python profiles/microbench/depgraph_dev_bench.py 20 2>&1 | tee $O/depgraph_dev.txt
FPX_DEBUG=1 timeout 300 python profiles/microbench/r05_windows.py 25 > $O/windows.txt 2> $O/windows.err; grep -h "lap 0:" $O/windows.txt; grep "slab of" $O/windows.err
timeout 1200 python -m pytest tests/test_gpu_sharding_world2.py tests/test_bench_distributed.py -m gpu -x -q -k "world or eight_rank" 2>&1 | tail -8
