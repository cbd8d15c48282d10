# K5 partition in one kernel (k_kp_part): parity, then back-to-back ticks by group size against the two-kernel build
# (profiles/microbench/build/libfpx_k5old.so = the tree before the change)
set -u; R=$PWD; mkdir -p gpurun_out/kpp
( timeout 600 python -m pytest tests/test_epaxos.py tests/test_epaxos_models.py -x -q -m gpu 2>&1 | tail -4 ) | tee gpurun_out/kpp/tests.txt
( timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config4 or epaxos" 2>&1 | tail -3 ) | tee -a gpurun_out/kpp/tests.txt
for rep in 1 2; do
  for pg in 2 4 8; do echo "pg=$pg: $(FPX_KP_PG=$pg K5_MODES=packed timeout 300 python profiles/microbench/k5v2_time.py pg$pg 2>&1 | grep 'back to back')"; done
  echo "old: $(FPX_LIB=$R/profiles/microbench/build/libfpx_k5old.so K5_MODES=packed timeout 300 python profiles/microbench/k5v2_time.py old 2>&1 | grep 'back to back')"
done 2>&1 | tee gpurun_out/kpp/times.txt
cd /tmp && export TMPDIR=/tmp
for pg in 4 8; do
rm -rf /tmp/kpp$pg; FPX_KP_PG=$pg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kpp$pg -o k5 -- python $R/bench.py --config 4 --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | cut -c1-200
find /tmp/kpp$pg -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/kpp/pg${pg}_kernel_stats.csv \;
grep -E "k_epx|k_kp" $R/gpurun_out/kpp/pg${pg}_kernel_stats.csv | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-120
done
