#!/bin/bash
# r06: K5 with fresh buffers per tick (bench.py --config 4: every tick its own inputs and its own 64 MB of output lines) against
# the back-to-back timing on one output buffer (k5v2_time.py); variants: ntout = the packed lines as nontemporal stores
cd /root/repo
export TMPDIR=/tmp
B=/root/repo/profiles/microbench/build
for i in 1 2; do for v in dpp ntout; do
  echo "$v: $(FPX_LIB=$B/libfpx_k5$v.so K5_MODES=packed timeout 300 python profiles/microbench/k5v2_time.py $v 2>&1 | grep 'back to back')"
  FPX_LIB=$B/libfpx_k5$v.so python bench.py --config 4 --steps 60 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v bench --config 4, 60 fresh ticks: ms per step', d['ms_per_step'], 'by events', d['roofline']['avg_kernel_ms'])"
done; done
