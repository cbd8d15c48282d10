# same box: variants of k_epx_key2 built into profiles/microbench/build/libfpx_k5<name>.so (old = the round's earlier build,
# a = sorting attempt over the tick's [0, m) first, b = index update off the key's tail, ab = both = the working tree)
set -u; R=$PWD; mkdir -p gpurun_out/k5ab2
for rep in 1 2; do for v in ${K5_VARIANTS:-old a b ab abf4}; do
  export FPX_LIB=$R/profiles/microbench/build/libfpx_k5$v.so
  [ -f $FPX_LIB ] || continue
  echo "$v: $(K5_MODES=packed timeout 300 python profiles/microbench/k5v2_time.py $v 2>&1 | grep 'back to back')"
done; done | tee gpurun_out/k5ab2/times.txt
unset FPX_LIB
if [ "${1:-}" = tests ]; then
  timeout 600 python -m pytest tests/test_epaxos.py tests/test_epaxos_models.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/k5ab2/tests.txt
  timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config4 or epaxos" 2>&1 | tail -3 | tee -a gpurun_out/k5ab2/tests.txt
fi
