set -u; O=gpurun_out/r05final; mkdir -p $O
for i in 1 2; do FPX_DEBUG=1 timeout 900 python bench.py > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench $i rc=$?"; done
python - <<'PY'
import json
for i in (1,2):
    d=json.load(open('gpurun_out/r05final/bench_%d.json'%i))
    print(i, d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'], d['roofline'].get('kernel_ms_min_max_sigma'), d['config'].get('placement'))
    for k,v in d['configs'].items(): print('  ',k, v.get('value'), v.get('ms_per_step'), v.get('avg_kernel_ms'), v.get('roofline_frac'), round(v.get('wall_s') or 0,1), v.get('error'))
PY
timeout 300 python bench.py --ballot acceptor --no-cpu-baseline --configs-block-steps 0 > $O/bench_acceptor.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench_acceptor.json')); print('acceptor', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'])"
