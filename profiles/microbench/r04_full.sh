mkdir -p gpurun_out/r4f
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -5 ) | tee gpurun_out/r4f/tests.txt
python bench.py > gpurun_out/r4f/bench.json 2> gpurun_out/r4f/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r4f/bench.json')); print('main', d['value'], d['ms_per_step'], d['roofline'].get('avg_kernel_ms'), d['roofline']['frac'], d['roofline'].get('frac_read')); print({k:(v.get('ms_per_step'), v.get('value'), v.get('error')) for k,v in d.get('configs',{}).items()})"
python bench.py --config thrifty --ballot acceptor --no-cpu-baseline > gpurun_out/r4f/bench_thrifty_acc.json 2>gpurun_out/r4f/bench_thrifty_acc.err; tail -c 600 gpurun_out/r4f/bench_thrifty_acc.json | head -c 300; echo
python -c "
import json; d=json.load(open('gpurun_out/r4f/bench_thrifty_acc.json')); print('thrifty acceptor', d['value'], d['ms_per_step'], d['roofline']['frac'])"
