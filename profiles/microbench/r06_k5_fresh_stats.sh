#!/bin/bash
# r06: per-kernel durations of K5 with one output buffer (k5v2_time.py) -- the fresh-buffer figures are bench.py --config 4's
# (profiles/r06_cfg4_kernel_stats.csv)
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/k5fresh
K5_MODES=packed timeout 240 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/k5fresh/same -o s -- python /root/repo/profiles/microbench/k5v2_time.py same 2>&1 | grep "back to back"
f=$(ls gpurun_out/k5fresh/same/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && grep -E "k_kp_hist|k_kp_scatter|k_epx_key2" $f | sed 's/(anonymous namespace):://g' | awk -F, '{print substr($1,1,40), $(NF-6), $(NF-4), $(NF-2), $(NF-1)}'
