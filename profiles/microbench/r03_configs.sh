mkdir -p gpurun_out/r03b
timeout 400 python bench.py > gpurun_out/r03b/bench_default.json 2> gpurun_out/r03b/bench_default.err; echo "bench rc=$?"
for N in 3 4 5; do bash profiles/microbench/cfg_prof.sh $N; cp gpurun_out/cfg$N/* gpurun_out/r03b/ 2>/dev/null; for f in gpurun_out/cfg$N/*; do cp $f gpurun_out/r03b/cfg${N}_$(basename $f); done; done
ls gpurun_out/r03b
