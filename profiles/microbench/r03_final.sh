# round-3 final build: full GPU suite, smoke, the default bench line, rocprofv3 kernel stats of the headline command and
# of configs 3 / 4 / 5 (run through gpurun from the repo root; results under gpurun_out/final/)
mkdir -p gpurun_out/final
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/final/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/final/pytest_gpu.txt | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.txt 2>&1; tail -1 gpurun_out/final/smoke.txt
( time timeout 400 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err ) 2> gpurun_out/final/bench_time.txt; echo "bench rc=$?"; grep real gpurun_out/final/bench_time.txt
bash profiles/microbench/headline_prof.sh; cp gpurun_out/headline/* gpurun_out/final/ 2>/dev/null
for N in 3 4 5; do bash profiles/microbench/cfg_prof.sh $N; for f in gpurun_out/cfg$N/*; do cp $f gpurun_out/final/cfg${N}_$(basename $f); done; done
ls gpurun_out/final
