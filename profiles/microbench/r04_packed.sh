mkdir -p gpurun_out/r4p
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "thrifty_runs or scattered" 2>&1 | grep -v amdgpu.ids | tail -4 ) | tee gpurun_out/r4p/tests.txt
for b in 0 1; do python profiles/microbench/thrifty_bench.py $b plain all first 2>&1 | grep "slots/s"; done | tee gpurun_out/r4p/thrifty.txt
python profiles/microbench/thrifty_bench.py 0 scattered random first 2>&1 | grep "slots/s" | sed 's/^/hint /' | tee -a gpurun_out/r4p/thrifty.txt
