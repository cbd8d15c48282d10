#!/bin/bash
# A/B: per-workgroup partial rows for the maxima (prev) vs whole-group scalars (default); interleaved reps
P='import sys,json; d=json.loads(sys.stdin.read()); print("%.4e slots/s  step %.4f ms  kernel %.4f ms  frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"]))'
for rep in 1 2 3 4; do
  for lib in "" prev; do
    for ballot in per_slot acceptor; do
      if [ -n "$lib" ]; then export FPX_LIB=$PWD/frankenpaxos_amd/csrc/variants/libfpx_$lib.so; else unset FPX_LIB; fi
      printf "rep %d %-8s %-9s " $rep "${lib:-default}" $ballot
      python bench.py --no-cpu-baseline --ballot $ballot 2>/dev/null | python -c "$P"
    done
  done
done
