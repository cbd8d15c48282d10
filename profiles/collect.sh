#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh r01
# 1. kernel trace + stats of the default bench command
# 2. PMC passes (separate runs, kernel-trace only): FETCH_SIZE, WRITE_SIZE for the bench and for the
#    known-byte-count kernels of profiles/microbench/hbm_mix.hip (calibration of the counters,
#    MI355X_MICROARCH.md "HBM": FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950, WRITE_SIZE
#    is uncalibrated)
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
hipcc -O3 --offload-arch=gfx950 -o /tmp/hbm_mix $ROOT/profiles/microbench/hbm_mix.hip
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
  python $ROOT/bench.py --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace_bench.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o bench -- \
    python $ROOT/bench.py --no-cpu-baseline --configs-block-steps 0 --steps 4 --warmup 1 > $OUT/pmc_${c}_bench.json 2> $OUT/pmc_${c}_bench.err
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o calib -- \
    /tmp/hbm_mix > $OUT/pmc_${c}_calib.txt 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o bench_acceptor -- \
    python $ROOT/bench.py --no-cpu-baseline --configs-block-steps 0 --steps 4 --warmup 1 --ballot acceptor > $OUT/pmc_${c}_bench_acceptor.json 2> /dev/null
done
find $OUT -name "*.csv" | head -40
du -sh $OUT
