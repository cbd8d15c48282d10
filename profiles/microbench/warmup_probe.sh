#!/bin/bash
# does a longer warm-up change the measured step time? (clock ramp / DVFS probe; measurement aid)
P='import sys,json; d=json.loads(sys.stdin.read()); print("steps %d warmup %d: %.4e slots/s  step %.4f ms  kernel %.4f ms" % (d["steps"], d["warmup"], d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"]))'
for w in 3 30 300 3 300; do
  python bench.py --no-cpu-baseline --ballot acceptor --steps 30 --warmup $w 2>/dev/null | python -c "$P"
done
for w in 3 35 3 35; do
  python bench.py --no-cpu-baseline --steps 5 --warmup $w 2>/dev/null | python -c "$P"
done
