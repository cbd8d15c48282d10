#!/bin/bash
# rocprofv3 evidence for thrifty delivery (random f+1 of 255 acceptors, first proposals), run through gpurun from
# the repo root:   bash profiles/collect_thrifty.sh r02
# throughput of every case, then kernel trace + stats and FETCH_SIZE / WRITE_SIZE in their own passes (kernel-trace
# only, one kind of k_phase2 launch per process) for both ballot models, plus the counter calibration kernels.
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_thrifty_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
hipcc -O3 --offload-arch=gfx950 -o /tmp/hbm_mix $ROOT/profiles/microbench/hbm_mix.hip
for b in 0 1; do
  timeout 300 python $ROOT/profiles/microbench/thrifty_bench.py $b > $OUT/thrifty_plain_$b.txt 2>&1
  timeout 300 python $ROOT/profiles/microbench/thrifty_bench.py $b scattered > $OUT/thrifty_scattered_$b.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$b -o thrifty -- \
    python $ROOT/profiles/microbench/thrifty_bench.py $b plain random first > /dev/null 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${c}_$b -o thrifty -- \
      python $ROOT/profiles/microbench/thrifty_bench.py $b plain random first > /dev/null 2>&1
  done
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o calib -- /tmp/hbm_mix > $OUT/pmc_${c}_calib.txt 2>&1
done
grep -h "slots/s" $OUT/thrifty_*.txt
find $OUT -name "*.csv" | wc -l
