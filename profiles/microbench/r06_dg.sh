# r06: device dependency-graph execution -- parity, the bench, per-kernel stats
R=$PWD; O=$R/gpurun_out/r06dg; mkdir -p $O
python -m pytest tests/test_depgraph_dev.py -q -x 2>&1 | tail -3
python profiles/microbench/depgraph_dev_bench.py 20 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dg_st -o k -- python $R/bench.py --config 4_execute --no-cpu-baseline --steps 10 > $O/line.json 2> $O/err.txt
find /tmp/dg_st -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
cd $R
python -c "
import json; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); print('4_execute ms_per_step', d['ms_per_step'], 'avg_kernel_ms', d['roofline']['avg_kernel_ms'], d['value'])"
head -22 $O/kernel_stats.csv | cut -c1-150
