# after the VALU diet of the small-group vote kernel (divisions by multiplication, a slot's row and group computed once per
# message, lane k raises acceptor k's maxima): configs 5 / 3 / 2 at three grids, then the parity suites that run it
set -u; O=gpurun_out/r05valu; mkdir -p $O
run() {  # config, grid
  local c=$1 g=$2
  if [ $g = 0 ]; then unset FPX_MAX_GRID; else export FPX_MAX_GRID=$g; fi
  timeout 200 python bench.py --config $c --no-cpu-baseline > $O/c${c}_g$g.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/c${c}_g$g.json')); print('config $c grid $g:', '%.4e'%d['value'], round(d['ms_per_step'],4), round(d['roofline']['avg_kernel_ms'],4))"
}
for g in 0 1280 1536; do run 5 $g; done
for g in 0 1536; do run 3 $g; run 2 $g; done
run 5 0
unset FPX_MAX_GRID
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_mencius_noop_range.py tests/test_gpu_parity.py tests/test_gpu_sharding_world2.py -m gpu -q -x 2>&1 | tail -3
