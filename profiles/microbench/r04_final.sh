# round-4 collection of the judged numbers: the default bench line, rocprofv3 kernel stats of the same command + the
# PMC passes (profiles/collect.sh), kernel stats of config 4 (run through gpurun from the repo root)
mkdir -p gpurun_out/final4
( time timeout 400 python bench.py > gpurun_out/final4/bench_default.json 2> gpurun_out/final4/bench_default.err ) 2> gpurun_out/final4/bench_time.txt; echo "bench rc=$?"; grep real gpurun_out/final4/bench_time.txt
python -c "
import json; d=json.load(open('gpurun_out/final4/bench_default.json')); print('main', d['value'], d['ms_per_step'], d['roofline'].get('avg_kernel_ms'), d['roofline']['frac'], d['roofline'].get('frac_read')); print({k:(v.get('ms_per_step'), v.get('value'), v.get('error')) for k,v in d.get('configs',{}).items()})"
bash profiles/collect.sh r04 > gpurun_out/final4/collect.log 2>&1; tail -3 gpurun_out/final4/collect.log
bash profiles/microbench/cfg_prof.sh 4
if [ "${1:-}" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/final4/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/final4/pytest_gpu.txt | tail -2
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final4/smoke.txt 2>&1; tail -1 gpurun_out/final4/smoke.txt
fi
