R=$PWD; O=$R/gpurun_out/r06c8; mkdir -p $O
python -m pytest tests/test_bench_distributed.py -q -x -k "preflight or single_gpu_bench_contract or spawns_its_own" 2>&1 | tail -5
bash profiles/microbench/cfg_pmc.sh 5 > $O/pmc_5.md 2>&1; cat $O/pmc_5.md
python bench.py > $O/bench_default.json 2> $O/bench_default.err; wc -c $O/bench_default.json; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d['configs'].items(): print(k, v)"
