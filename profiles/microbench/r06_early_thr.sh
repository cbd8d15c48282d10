#!/bin/bash
# r06: the first step's ballot row requested with the chunk's other loads (FPX_EARLY_THR=1, libfpx.so) against behind the
# wait for them (libfpx_noearly.so = the same sources built with -DFPX_EARLY_THR=0); same box, alternating
cd /root/repo
export TMPDIR=/tmp
NO=/root/repo/frankenpaxos_amd/csrc/libfpx_noearly.so
one() { python bench.py --config $1 --steps $2 --no-cpu-baseline --configs-block-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$3 $1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
  for c in "headline 20" "2 200" "3 200" "5 40" "thrifty 20" "thrifty_random 20" "adversarial 40" "acceptor_model 20"; do
    set -- $c
    one $1 $2 early
    FPX_LIB=$NO one $1 $2 noearly
  done
done
