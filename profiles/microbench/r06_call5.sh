# r06: the host path as shipped -- inputs by the copy engine, records written by the vote kernel into the caller's arrays
R=$PWD; O=$R/gpurun_out/r06c5; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "page_locked or host" 2>&1 | tail -3
for i in 1 2 3; do
  python bench.py --config host_path --no-cpu-baseline --steps 20 --warmup 3 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('host_path ms_per_step', d['ms_per_step'], 'kernel', d['roofline']['avg_kernel_ms'], 'pcie', d['config']['pcie_GBs'])"
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/hp_tl -o k -- python $R/bench.py --config host_path --no-cpu-baseline --steps 8 --warmup 2 > $O/host_path_line.json 2> $O/host_path_err.txt
find /tmp/hp_tl -name "*kernel_trace.csv" -exec cp {} $O/host_path_kernel_trace.csv \;
find /tmp/hp_tl -name "*memory_copy_trace.csv" -exec cp {} $O/host_path_memcpy_trace.csv \;
cd $R
python profiles/microbench/timeline.py $O/host_path_kernel_trace.csv 24 > $O/host_path_timeline.txt 2>&1
tail -20 $O/host_path_timeline.txt
python bench.py --no-cpu-baseline --configs-block-steps 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline ms_per_step', d['ms_per_step'], 'kernel', d['roofline']['avg_kernel_ms'])"
