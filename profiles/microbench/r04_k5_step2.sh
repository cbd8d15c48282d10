( timeout 900 python -m pytest tests/test_epaxos.py -x -q -m gpu 2>&1 | tail -5 )
python profiles/microbench/k5_ablate.py run 2>&1 | tail -8
bash profiles/microbench/k5_pmc.sh > /dev/null 2>&1; python profiles/microbench/k5_pmc_table.py 2>&1 | tail -8
