# round 4, K5 after the staged scatter / whole-tick buckets: parity, timing, config-4 bench (plain, under rocprofv3), PMC
set -u; R=$PWD; O=gpurun_out/r4k5; mkdir -p $O
( timeout 600 python -m pytest tests/test_epaxos.py tests/test_epaxos_models.py -x -q -m gpu 2>&1 | tail -3 ) > $O/epx_tests.txt
( timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config4 or epaxos or cfg4" 2>&1 | tail -3 ) >> $O/epx_tests.txt
K5_MODES=packed timeout 300 python profiles/microbench/k5v2_time.py new 2>&1 | grep -v amdgpu.ids > $O/k5_time.txt
timeout 300 python bench.py --config 4 --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_cfg4_plain.json 2> $O/bench_cfg4_plain.err
( time timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k5prof -o k5 -- python $R/bench.py --config 4 --no-cpu-baseline --steps 10 --warmup 2 > $R/$O/bench_cfg4.json 2> $R/$O/bench_cfg4.err
find /tmp/k5prof -name "*kernel_stats.csv" -exec cp {} $R/$O/cfg4_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 250 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/k5pmc_$c -o k5 -- python $R/bench.py --config 4 --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2> $R/$O/err_$c.txt
  find /tmp/k5pmc_$c -name "*counter_collection.csv" -exec cp {} $R/$O/$c.csv \;
done
cd $R; cat $O/epx_tests.txt $O/k5_time.txt; grep -E "k_epx|k_kp" $O/cfg4_kernel_stats.csv | cut -c1-200; cut -c1-330 $O/bench_cfg4_plain.json
python - <<PY
import csv, collections, json
for c, scale in (("WRITE_SIZE", 1024.0), ("FETCH_SIZE", 2048.0)):
    tab = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open("$O/%s.csv" % c)):
        n = r["Kernel_Name"]
        for k in ("k_kp_scatter", "k_epx_key2", "k_kp_hist"):
            if k in n: tab[k][0] += float(r["Counter_Value"]) * scale; tab[k][1] += 1
    print(c, {k: (round(v[0] / max(v[1], 1) / 1e6, 1), v[1]) for k, v in tab.items()})
d = json.load(open("$O/bench_default.json")); print('main', d['value'], d['ms_per_step'], d['roofline']['frac']); print({k:(v.get('ms_per_step'), v.get('value'), v.get('error')) for k,v in d.get('configs',{}).items()})
PY
