# PMC traffic of the K5 tick (bench.py --config 4): FETCH_SIZE and WRITE_SIZE in separate runs, kernel trace only
mkdir -p gpurun_out/k5pmc; R=$PWD; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 250 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/k5pmc_$c -o k5 -- python $R/bench.py --config 4 --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2> $R/gpurun_out/k5pmc/err_$c.txt
  find /tmp/k5pmc_$c -name "*counter_collection.csv" -exec cp {} $R/gpurun_out/k5pmc/$c.csv \;
done
ls -la $R/gpurun_out/k5pmc
