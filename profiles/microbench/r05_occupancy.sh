# config 5 (and 3, 2): the small-group vote kernel is bound by dependent round trips (messages -> tally-key row -> rows out, 8
# chunks per wavefront) with 16 wavefronts per CU; its LDS (23.5 KB per workgroup, 12 KB of it the column-quad staging that a
# grouped batch never touches) allows 6 workgroups per CU, its 61 VGPRs 7.  FPX_NO_QUADS=1 drops the staging; FPX_MAX_GRID
# picks the workgroups: does a grid that is RESIDENT AT ONCE (5, 6, 7 per CU) beat 4 per CU?
set -u; O=gpurun_out/r05occ; mkdir -p $O
run() {  # config, quads(0/1), grid
  local c=$1 q=$2 g=$3
  if [ $q = 1 ]; then export FPX_NO_QUADS=1; else unset FPX_NO_QUADS; fi
  if [ $g = 0 ]; then unset FPX_MAX_GRID; else export FPX_MAX_GRID=$g; fi
  timeout 200 python bench.py --config $c --no-cpu-baseline > $O/c${c}_q${q}_g$g.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/c${c}_q${q}_g$g.json')); print('config $c no_quads $q grid $g:', '%.4e'%d['value'], round(d['ms_per_step'],4), round(d['roofline']['avg_kernel_ms'],4))"
}
for q in 0 1; do for g in 0 1280 1536 1792 2048; do run 5 $q $g; done; done
run 5 0 0
for g in 0 1536 1792; do run 3 1 $g; run 2 1 $g; done
