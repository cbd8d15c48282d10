#!/bin/bash
# r06: bench.py --config 4: how much of a tick's time is the timed region's length and the warm-up in front of it?
cd /root/repo
for a in "20 3" "60 3" "40 40" "20 3" "60 3" "40 40"; do set -- $a
  timeout 300 python bench.py --config 4 --steps $1 --warmup $2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('steps', d['steps'], 'warmup', d['warmup'], 'ms per step', round(d['ms_per_step'],5), 'by events', round(d['roofline']['avg_kernel_ms'],5))"; done
