# r05 call 1: placement diagnostic (standalone HIP), baseline bench line of this box, K5 base vs fpr2
set -u; R=$PWD; O=gpurun_out/r05c1; mkdir -p $O
timeout 300 profiles/microbench/build/r05_placement 25 1024 > $O/placement_1024.txt 2>&1; echo "placement1024 rc=$?"
timeout 300 profiles/microbench/build/r05_placement 25 256 > $O/placement_256.txt 2>&1; echo "placement256 rc=$?"
timeout 300 profiles/microbench/build/r05_placement 25 32 > $O/placement_32.txt 2>&1; echo "placement32 rc=$?"
FPX_DEBUG=1 timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
grep placement $O/bench.err | head
K5_VARIANTS="base fpr2" bash profiles/microbench/r04_k5_ab2.sh > $O/k5ab.txt 2>&1
cp gpurun_out/k5ab2/times.txt $O/k5_times.txt
tail -5 $O/k5_times.txt
