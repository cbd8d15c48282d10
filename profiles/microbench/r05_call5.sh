set -u; O=gpurun_out/r05c5; mkdir -p $O
FPX_DEBUG=1 timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
grep -c "placement window" $O/bench.err; grep "slab of" $O/bench.err | head
tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05c5/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'])
for k,v in d['configs'].items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('avg_kernel_ms'), v.get('roofline_frac'), v.get('wall_s'), v.get('error'))
PY
FPX_PLACEMENT_CHUNKS=0 timeout 300 python bench.py --no-cpu-baseline --configs-block-steps 0 > $O/bench_nochunks.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_nochunks.json')); print('nochunks', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])"
timeout 300 python bench.py --no-cpu-baseline --configs-block-steps 0 > $O/bench2.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench2.json')); print('chunks again', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
