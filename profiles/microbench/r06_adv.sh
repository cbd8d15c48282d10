# r06: the adversarial stream -- messages per wavefront at ~21.5 k proposals per launch, and the per-epoch timeline
R=$PWD; O=$R/gpurun_out/r06adv; mkdir -p $O
for mc in 4 2 1 8; do
  FPX_MIN_CHUNK=$mc python bench.py --config adversarial --no-cpu-baseline --steps 12 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('min_chunk $mc: proposals/s %.4g  ms per stream %.4f' % (d['value'], d['ms_per_step']))"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/adv_tl -o k -- python $R/bench.py --config adversarial --no-cpu-baseline --steps 4 > $O/adv_line.json 2> $O/adv_err.txt
find /tmp/adv_tl -name "*kernel_trace.csv" -exec cp {} $O/adv_kernel_trace.csv \;
find /tmp/adv_tl -name "*kernel_stats.csv" -exec cp {} $O/adv_kernel_stats.csv \;
cd $R
python profiles/microbench/timeline.py $O/adv_kernel_trace.csv 60 > $O/adv_timeline.txt 2>&1
tail -40 $O/adv_timeline.txt
head -12 $O/adv_kernel_stats.csv
