#!/bin/bash
# finer sweep of the byte offset between the cell arrays (per_slot model), then repeats of the extremes
P='import sys,json; d=json.loads(sys.stdin.read()); print("kernel %.4f ms  frac %.3f" % (d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"]))'
for st in 0 256 512 1024 2048 3072 4096 4352 5120 6144 8192 12288 16384 20480 32768 65536 69632 131072 262144 524288 1048576 1118208 2097152 4194304; do
  printf "stagger %8d " $st
  FPX_STAGGER=$st python bench.py --no-cpu-baseline 2>/dev/null | python -c "$P"
done
