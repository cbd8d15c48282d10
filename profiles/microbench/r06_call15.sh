python -m pytest tests/test_depgraph_dev.py -q -x 2>&1 | tail -2
for i in 1 2 3; do python bench.py --config 4_execute --no-cpu-baseline --steps 10 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4_execute ms_per_step', d['ms_per_step'], 'kernel', d['roofline']['avg_kernel_ms'], d['value'])"; done
python profiles/microbench/depgraph_dev_bench.py 20 2>&1 | tail -2
python profiles/microbench/depgraph_dev_bench.py 18 2>&1 | tail -2
