# after the vmcnt fix: parity, then the thrifty / dense rates and the bench lines, main build and FPX_PREFETCH=1 build
mkdir -p gpurun_out/r4w
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r4w/tests.txt; cat gpurun_out/r4w/tests.txt
for b in 0 1; do python profiles/microbench/run_len_probe.py $b 2>&1 | grep "run of"; done | tee gpurun_out/r4w/run_len.txt
for b in 0 1; do python profiles/microbench/thrifty_bench.py $b 2>&1 | grep "slots/s"; done | tee gpurun_out/r4w/thrifty.txt
python bench.py --no-cpu-baseline > gpurun_out/r4w/bench_main.json 2> gpurun_out/r4w/bench_main.err; python -c "
import json; d=json.load(open('gpurun_out/r4w/bench_main.json')); print('main', d['value'], d['ms_per_step'], d['roofline'].get('avg_kernel_ms'), d['roofline']['frac']); print({k:(v.get('ms_per_step'), v.get('value')) for k,v in d.get('configs',{}).items()})"
python bench.py --no-cpu-baseline --ballot acceptor --configs-block-steps 0 > gpurun_out/r4w/bench_acc.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r4w/bench_acc.json')); print('acceptor', d['value'], d['ms_per_step'], d['roofline']['frac'])"
export FPX_LIB=$PWD/profiles/microbench/build/libfpx_PF1.so
python bench.py --no-cpu-baseline --configs-block-steps 0 > gpurun_out/r4w/bench_pf1.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r4w/bench_pf1.json')); print('pf1', d['value'], d['ms_per_step'], d['roofline'].get('avg_kernel_ms'), d['roofline']['frac'])"
python profiles/microbench/run_len_probe.py 1 2>&1 | grep "run of" | sed 's/^/pf1 /' | tee -a gpurun_out/r4w/run_len.txt
