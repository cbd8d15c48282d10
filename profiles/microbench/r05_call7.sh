set -u; O=gpurun_out/r05c7; mkdir -p $O
FPX_DEBUG=1 timeout 300 python profiles/microbench/r05_windows.py 25 > $O/windows_chunks1.txt 2> $O/windows_chunks1.err
grep -h "lap 0\|placement:" $O/windows_chunks1.txt | cut -c1-330; grep "slab of" $O/windows_chunks1.err
timeout 300 python bench.py --no-cpu-baseline --configs-block-steps 0 > $O/bench_a.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_a.json')); print('headline', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])"
for s in 1 0; do FPX_CFG5_SERIAL=$s timeout 300 python bench.py --config 5 --no-cpu-baseline > $O/cfg5_serial$s.json 2>$O/cfg5_serial$s.err; python -c "
import json; d=json.load(open('$O/cfg5_serial$s.json')); print('cfg5 serial=$s', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])"; done
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "band or config5" 2>&1 | tail -5
