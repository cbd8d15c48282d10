# r05, last build: the rocprofv3 evidence again with the final bench.py (collect.sh r05) + the device dependency graph's kernels
set -u; R=$PWD
bash profiles/collect.sh r05 > gpurun_out/r05_collect.log 2>&1
mkdir -p gpurun_out/r05dg; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05dg -o dg -- python $R/profiles/microbench/depgraph_dev_bench.py 20 > $R/gpurun_out/r05dg/bench_under_rocprof.txt 2>&1
cd $R; ls gpurun_out/r05dg | head; tail -3 gpurun_out/r05dg/bench_under_rocprof.txt
