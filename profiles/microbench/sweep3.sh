#!/bin/bash
# A/B: nontemporal ballot loads, 32-message chunks (measurement aid); 3 interleaved reps
P='import sys,json; d=json.loads(sys.stdin.read()); print("%.4e slots/s  step %.4f ms  kernel %.4f ms  frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"]))'
for rep in 1 2 3; do
  for lib in "" ntl ch32 ch32ntl; do
    for ballot in per_slot acceptor; do
      if [ -n "$lib" ]; then export FPX_LIB=$PWD/frankenpaxos_amd/csrc/variants/libfpx_$lib.so; else unset FPX_LIB; fi
      case "$lib" in ch32*) export FPX_MAX_GRID=8192;; *) unset FPX_MAX_GRID;; esac
      printf "rep %d %-8s %-9s " $rep "${lib:-default}" $ballot
      python bench.py --no-cpu-baseline --ballot $ballot 2>/dev/null | python -c "$P"
    done
  done
done
