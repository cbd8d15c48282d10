# round 4, last collection: K5 variants A/B, the whole GPU suite + smoke + the EPaxos stress, then the K5 numbers of the kept build
set -u; R=$PWD; O=gpurun_out/final4b; mkdir -p $O
K5_VARIANTS="vec0 fix1 dsum" bash profiles/microbench/r04_k5_ab2.sh 2>&1 | tee $O/ab.txt
unset FPX_LIB
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 ) > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
( timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/smoke.txt; cat $O/smoke.txt
( timeout 300 python profiles/microbench/stress_epaxos.py 2>&1 | tail -3 ) > $O/stress_epaxos.txt; cat $O/stress_epaxos.txt
bash profiles/microbench/r04_k5_final.sh 2>&1 | tail -12
