# r06, the round's LAST build (Phase1a in one launch, K5's scatter tail and nontemporal lines, the early ballot row at G = 1):
# the whole -m gpu suite, headline evidence again (kernel trace + PMC), the configs whose kernels changed, the default line twice
R=$PWD; O=$R/gpurun_out/r06col3; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
bash profiles/collect.sh r06 > $O/collect.log 2>&1
for N in 4 adversarial; do timeout 600 bash profiles/microbench/cfg_pmc.sh $N > $O/pmc_$N.md 2>&1; done
cat $O/pmc_4.md $O/pmc_adversarial.md | grep -v k_probe | tail -40
cd /tmp && export TMPDIR=/tmp
for N in 2 4 adversarial; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st3_$N -o k -- python $R/bench.py --config $N --no-cpu-baseline --steps 40 --warmup 2 > $O/cfg${N}_bench_under_rocprof.json 2> /dev/null
  find /tmp/st3_$N -name "*kernel_stats.csv" -exec cp {} $O/cfg${N}_kernel_stats.csv \;
done
cd $R
for i in 1 2; do timeout 900 python bench.py > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench $i rc=$? bytes=$(wc -c < $O/bench_$i.json)"; done
python - <<'PY'
import json
for i in (1,2):
    d=json.loads(open('gpurun_out/r06col3/bench_%d.json'%i).read().strip().splitlines()[-1])
    print(i, d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'], d['roofline'].get('kernel_ms_min_max_sigma'))
    for k,v in d['configs'].items():
        if isinstance(v, dict): print('  ',k, v.get('value'), v.get('ms_per_step'), v.get('avg_kernel_ms'), v.get('roofline_frac'), v.get('error'), v.get('wall_s'))
PY
timeout 300 python bench.py --ballot acceptor --no-cpu-baseline --configs-block-steps 0 > $O/bench_acceptor.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
