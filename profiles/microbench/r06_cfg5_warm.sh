#!/bin/bash
# r06: bench.py --config 5: the timed region behind 2 / 20 warm-up bands (are the clocks still rising, as for config 4?)
cd /root/repo
for a in "40 2" "40 20" "40 2" "40 20"; do set -- $a
  timeout 400 python bench.py --config 5 --steps $1 --warmup $2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('steps', d['steps'], 'warmup', d['warmup'], 'ms per step', round(d['ms_per_step'],5), 'by events', round(d['roofline']['avg_kernel_ms'],5))"; done
