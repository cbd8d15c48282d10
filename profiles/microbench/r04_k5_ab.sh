# same box, three builds of K5's partition: cur = the working tree, old = HEAD (profiles/microbench/build/libfpx_k5old.so)
set -u; R=$PWD; mkdir -p gpurun_out/k5ab
for rep in 1 2; do for v in cur old; do  # (build the other tree into profiles/microbench/build/libfpx_k5old.so first)
  if [ $v = cur ]; then unset FPX_LIB; else export FPX_LIB=$R/profiles/microbench/build/libfpx_k5$v.so; fi
  echo "$v: $(K5_MODES=packed timeout 300 python profiles/microbench/k5v2_time.py $v 2>&1 | grep 'back to back')"
done; done
cd /tmp && export TMPDIR=/tmp
for v in cur old; do  # (build the other tree into profiles/microbench/build/libfpx_k5old.so first)
  if [ $v = cur ]; then unset FPX_LIB; else export FPX_LIB=$R/profiles/microbench/build/libfpx_k5$v.so; fi
  for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/ab_$v$c; timeout 250 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/ab_$v$c -o k5 -- python $R/bench.py --config 4 --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2>&1
  find /tmp/ab_$v$c -name "*counter_collection.csv" -exec cp {} $R/gpurun_out/k5ab/${v}_$c.csv \;
  done
  python - <<PY
import csv, collections
for c, scale in (("WRITE_SIZE", 1024.0), ("FETCH_SIZE", 2048.0)):
    tab = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open("$R/gpurun_out/k5ab/${v}_%s.csv" % c)):
        n = r["Kernel_Name"]
        for k in ("k_kp_scatter", "k_epx_key2", "k_kp_hist"):
            if k in n: tab[k][0] += float(r["Counter_Value"]) * scale; tab[k][1] += 1
    print("$v", c, {k: round(v[0] / max(v[1], 1) / 1e6, 1) for k, v in tab.items()})
PY
done
