# config 5 / 3 / 2: does the small-group vote kernel want more workgroups on big batches?  (FPX_MAX_GRID: tuning aid of fpx_create)
set -u; O=gpurun_out/r05grid; mkdir -p $O
for c in 5 3 2; do for g in 0 2048 4096 8192; do
  if [ $g = 0 ]; then unset FPX_MAX_GRID; else export FPX_MAX_GRID=$g; fi
  timeout 200 python bench.py --config $c --no-cpu-baseline > $O/c${c}_g$g.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/c${c}_g$g.json')); print('config $c grid $g:', '%.4e'%d['value'], round(d['ms_per_step'],4), round(d['roofline']['avg_kernel_ms'],4))"
done; done
