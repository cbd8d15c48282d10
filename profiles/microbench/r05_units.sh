# the small-group vote kernel (G <= 8): FPX_UNITS_BLOCKED bit 0 = a wavefront takes neighbouring chunks, bit 1 = the next
# chunk's messages are requested while this chunk's tally keys travel; FPX_MAX_GRID = workgroups
set -u; O=gpurun_out/r05units; mkdir -p $O
run() {  # config, blocked, grid
  local c=$1 k=$2 g=$3
  export FPX_UNITS_BLOCKED=$k
  if [ $g = 0 ]; then unset FPX_MAX_GRID; else export FPX_MAX_GRID=$g; fi
  timeout 200 python bench.py --config $c --no-cpu-baseline > $O/c${c}_k${k}_g$g.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/c${c}_k${k}_g$g.json')); print('config $c blocked/prefetch $k grid $g:', '%.4e'%d['value'], round(d['ms_per_step'],4), round(d['roofline']['avg_kernel_ms'],4))"
}
for g in 0 1280; do for k in 0 1 2 3; do run 5 $k $g; done; done
for k in 0 2 3; do run 3 $k 0; run 2 $k 0; done
unset FPX_MAX_GRID
FPX_UNITS_BLOCKED=3 timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_mencius_noop_range.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
