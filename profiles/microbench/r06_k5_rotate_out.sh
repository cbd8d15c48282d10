#!/bin/bash
# r06: K5 back to back: 8 ticks' inputs in turn into one output buffer / into 8 output buffers in turn; 40 ticks' inputs into
# 40 output buffers (what bench.py --config 4 does: nothing is touched twice)
cd /root/repo
B=/root/repo/profiles/microbench/build
for i in 1 2; do for v in dpp; do
  FPX_LIB=$B/libfpx_k5$v.so K5_ROTATE_OUT=1 K5_MODES=packed timeout 200 python profiles/microbench/k5v2_time.py "$v T=8" 2>&1 | grep 'back to back'
  FPX_LIB=$B/libfpx_k5$v.so K5_T=40 K5_ROTATE_OUT=1 K5_MODES=packed timeout 400 python profiles/microbench/k5v2_time.py "$v T=40" 2>&1 | grep 'back to back'
done; done
