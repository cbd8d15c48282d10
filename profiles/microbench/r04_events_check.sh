# the library's kernel timing rides on the dispatch packet (hipExtLaunchKernelGGL): HIP-event time vs rocprofv3 of the same run,
# and the default line's configs block
mkdir -p gpurun_out/evchk; R=$PWD
python bench.py --no-cpu-baseline > gpurun_out/evchk/bench.json 2> gpurun_out/evchk/bench.err; tail -3 gpurun_out/evchk/bench.err
python -c "
import json; d=json.load(open('gpurun_out/evchk/bench.json')); print('main', d['value'], d['ms_per_step'], d['roofline'].get('avg_kernel_ms'), d['roofline']['frac']); print({k:(v.get('ms_per_step'), v.get('avg_kernel_ms'), v.get('steps'), round(v.get('wall_s',0),1), v.get('error')) for k,v in d.get('configs',{}).items()})"
cd /tmp && export TMPDIR=/tmp
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profh -o bench -- python $R/bench.py --no-cpu-baseline --configs-block-steps 0 > $R/gpurun_out/evchk/bench_under_rocprof.json 2> $R/gpurun_out/evchk/err.txt
find /tmp/profh -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/evchk/ \;
cd $R; grep k_phase2 gpurun_out/evchk/*kernel_stats.csv | cut -d, -f2-4,6,7; python -c "
import json; d=json.load(open('gpurun_out/evchk/bench_under_rocprof.json')); print('under rocprof: events', d['roofline'].get('avg_kernel_ms'), 'step', d['ms_per_step'])"
for c in 2 thrifty; do timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$c -o c -- python $R/bench.py --config $c --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('cfg', sys.argv[1], 'events', l['roofline']['avg_kernel_ms'], 'step', l['ms_per_step'])" $c; grep k_phase2 /tmp/prof$c/*/*kernel_stats.csv | cut -d, -f2-4 | cut -c1-200; done
