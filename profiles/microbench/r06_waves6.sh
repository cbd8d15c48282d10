#!/bin/bash
# r06: the adversarial stream's vote kernel (k_phase2<64,2,2,true>: 95 registers, 5 wavefronts per SIMD) compiled for 6 wavefronts
# per SIMD (80 registers + 40 bytes of scratch per lane); same box, alternating
cd /root/repo
W=/root/repo/profiles/microbench/build/libfpx_w6.so
one() { timeout 300 python bench.py --config $1 --steps $2 --no-cpu-baseline --configs-block-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$3 $1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  one adversarial 40 plain
  FPX_LIB=$W one adversarial 40 waves6
done
