# r06: host path A/B -- copy engines on their own streams (default) vs copies as workgroups of the vote kernel (carry)
R=$PWD; O=$R/gpurun_out/r06c3; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "page_locked" 2>&1 | tail -3
for mode in engines carry engines carry; do
  if [ $mode = carry ]; then export FPX_HOST_STAGE=carry; else unset FPX_HOST_STAGE; fi
  python bench.py --config host_path --no-cpu-baseline --steps 20 --warmup 3 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode host_path ms_per_step', d['ms_per_step'], 'kernel', d['roofline']['avg_kernel_ms'], 'pcie', d['config']['pcie_GBs'])"
done
unset FPX_HOST_STAGE
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/hp_tl -o k -- python $R/bench.py --config host_path --no-cpu-baseline --steps 8 --warmup 2 > $O/host_path_line.json 2> $O/host_path_err.txt
find /tmp/hp_tl -name "*kernel_trace.csv" -exec cp {} $O/host_path_kernel_trace.csv \;
find /tmp/hp_tl -name "*memory_copy_trace.csv" -exec cp {} $O/host_path_memcpy_trace.csv \;
cd $R
python profiles/microbench/timeline.py $O/host_path_kernel_trace.csv 30 > $O/host_path_timeline.txt 2>&1
tail -24 $O/host_path_timeline.txt
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r06c3/host_path_memcpy_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[-40]["Start_Timestamp"]) if len(rows)>40 else int(rows[0]["Start_Timestamp"])
for r in rows[-28:]:
    a,b=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%9.1f dur %7.1f %s %s" % ((a-t0)/1e3,(b-a)/1e3,r.get("Direction",""),r.get("Bytes","")))
PY
