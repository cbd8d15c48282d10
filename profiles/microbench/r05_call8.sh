set -u; O=gpurun_out/r05c8; mkdir -p $O; R=$PWD
timeout 600 python -m pytest tests/test_depgraph_dev.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python profiles/microbench/depgraph_dev_bench.py 20 2>&1 | tee $O/depgraph_dev.txt
timeout 300 python profiles/microbench/depgraph_dev_bench.py 18 2>&1 | tee -a $O/depgraph_dev.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/dgprof -o dg -- python $R/profiles/microbench/depgraph_dev_bench.py 20 > $R/$O/dg_under_rocprof.txt 2>&1
cd $R
find $O/dgprof -name "*kernel_stats.csv" | head -2
for t in 3 8; do FPX_PLACEMENT_A_TRIES=$t FPX_DEBUG=1 timeout 300 python profiles/microbench/r05_windows.py 25 > $O/windows_a$t.txt 2> $O/windows_a$t.err; grep -h "lap 0:" $O/windows_a$t.txt; grep "slab of" $O/windows_a$t.err; done
