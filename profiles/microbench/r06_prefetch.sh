#!/bin/bash
# r06: the next step's ballot row requested before this step is decided (-DFPX_PREFETCH=1, measured at the headline's size in
# round 1: no gain) on the SMALL launches, which are one wavefront's chain of round trips long; same box, alternating
cd /root/repo
PF=/root/repo/profiles/microbench/build/libfpx_prefetch.so
one() { timeout 300 python bench.py --config $1 --steps $2 --no-cpu-baseline --configs-block-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$3 $1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
  for c in "adversarial 40" "2 200" "3 200" "thrifty_random 20" "headline 20"; do
    set -- $c
    one $1 $2 plain
    FPX_LIB=$PF one $1 $2 prefetch
  done
done
