python -m pytest tests/test_next_rows.py tests/test_depgraph_dev.py tests/test_mencius_noop_range.py tests/test_c_example.py -q -x 2>&1 | tail -3
python profiles/microbench/next_rows_bench.py 2>&1 | tail -12
python profiles/microbench/depgraph_dev_bench.py 20 2>&1 | tail -3
FPX_DG_WIDE=1 python profiles/microbench/depgraph_dev_bench.py 20 2>&1 | tail -3
