set -u; O=gpurun_out/r06final; mkdir -p $O
for i in 1 2; do timeout 900 python bench.py > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench $i rc=$? bytes=$(wc -c < $O/bench_$i.json)"; done
python - <<'PY'
import json
for i in (1,2):
    d=json.loads(open('gpurun_out/r06final/bench_%d.json'%i).read().strip().splitlines()[-1])
    print(i, d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'], d['roofline'].get('kernel_ms_min_max_sigma'), d['config'].get('placement'))
    for k,v in d['configs'].items():
        if isinstance(v, dict): print('  ',k, v.get('value'), v.get('ms_per_step'), v.get('avg_kernel_ms'), v.get('roofline_frac'), v.get('wall_s'), v.get('error'))
PY
timeout 300 python bench.py --ballot acceptor --no-cpu-baseline --configs-block-steps 0 > $O/bench_acceptor.json 2>/dev/null
python -c "
import json; d=json.loads(open('$O/bench_acceptor.json').read().strip().splitlines()[-1]); print('acceptor', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'])"
python -c "import __graft_entry__ as g; g.smoke()"
