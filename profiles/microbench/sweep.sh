#!/bin/bash
# tuning sweep (measurement aid): variants of libfpx x grid sizes, per-slot and acceptor ballot models
P='import sys,json; d=json.loads(sys.stdin.read()); print("%.4e slots/s  step %.4f ms  kernel %.4f ms  frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"]))'
for lib in "" u1 u2 u4plain u2plain; do
  for grid in 2048 4096 8192; do
    for ballot in per_slot acceptor; do
      if [ -n "$lib" ]; then export FPX_LIB=$PWD/frankenpaxos_amd/csrc/variants/libfpx_$lib.so; else unset FPX_LIB; fi
      printf "%-8s grid %5d %-9s " "${lib:-default}" $grid $ballot
      FPX_MAX_GRID=$grid python bench.py --no-cpu-baseline --ballot $ballot 2>/dev/null | python -c "$P"
    done
  done
done
