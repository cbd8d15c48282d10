#!/bin/bash
# r06: K5 with nothing touched twice (40 ticks' inputs, 40 output buffers, 200 ticks back to back): the packed lines as
# nontemporal stores (ntout), the tick's inputs as nontemporal loads as well (ntio), against the plain build (dpp)
cd /root/repo
B=/root/repo/profiles/microbench/build
for i in 1 2; do for v in dpp ntout ntio; do
  FPX_LIB=$B/libfpx_k5$v.so K5_T=40 K5_ROTATE_OUT=1 K5_MODES=packed timeout 400 python profiles/microbench/k5v2_time.py "$v T=40" 2>&1 | grep 'in turn'
done; done
