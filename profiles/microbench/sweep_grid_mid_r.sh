#!/bin/bash
# Round-2 follow-up (not yet run): does the 4-workgroups-per-CU default that doubled R = 3 also help the middle
# group sizes (G = 8 .. 32 lanes per slot)?  Run through gpurun from the repo root:
#   bash profiles/microbench/sweep_grid_mid_r.sh > gpurun_out/sweep_grid_mid_r.txt
for g in 1024 2048 4096 8192; do
  echo "max_grid $g"
  SMALL_R=32,64,128 FPX_MAX_GRID=$g timeout -s KILL 120 python profiles/microbench/small_r_bench.py 2>&1 | grep "ballot model"
done
