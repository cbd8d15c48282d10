#!/bin/bash
# r06: runtime knobs of the HIP runtime against the launch floor (~4.8 us per dependent kernel): HIP_FORCE_DEV_KERNARG (kernel
# arguments in device memory), AMD_OPT_FLUSH (device-scope instead of system-scope fences between kernels)
cd /root/repo
export TMPDIR=/tmp
one() { python bench.py --config $1 --steps $2 --no-cpu-baseline --configs-block-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$3 $1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
  for c in "2 200" "adversarial 40" "4 20" "4_execute 20" "5 40" "headline 20"; do
    set -- $c
    one $1 $2 default
    HIP_FORCE_DEV_KERNARG=1 one $1 $2 devkernarg
    AMD_OPT_FLUSH=0 one $1 $2 optflush0
    AMD_OPT_FLUSH=1 HIP_FORCE_DEV_KERNARG=1 one $1 $2 optflush1_devkernarg
  done
done
