#!/bin/bash
# A/B: byte offset between the cell arrays (DRAM channel / bank alignment of the lockstep streams)
P='import sys,json; d=json.loads(sys.stdin.read()); print("%.4e slots/s  step %.4f ms  kernel %.4f ms  frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"]))'
for rep in 1 2 3; do
  for st in 0 1024 4352 69632 1118208 8454144; do
    for ballot in per_slot acceptor; do
      printf "rep %d stagger %8d %-9s " $rep $st $ballot
      FPX_STAGGER=$st python bench.py --no-cpu-baseline --ballot $ballot 2>/dev/null | python -c "$P"
    done
  done
done
