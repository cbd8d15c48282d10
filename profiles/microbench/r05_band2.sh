# the Mencius band in two launches (k_phase2_band + k_ranges_fill_lg_fin): parity suites that run the changed kernels, then
# bench.py --config 5 in the two-launch form (default) and with the halves one after the other (FPX_CFG5_SERIAL=1)
set -u; O=gpurun_out/r05band2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_mencius_noop_range.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4
for rep in 1 2; do
  for m in merged serial; do
    if [ $m = serial ]; then export FPX_CFG5_SERIAL=1; else unset FPX_CFG5_SERIAL; fi
    timeout 200 python bench.py --config 5 --no-cpu-baseline > $O/c5_${m}_$rep.json 2>$O/err_${m}_$rep.txt || tail -3 $O/err_${m}_$rep.txt
    python -c "
import json; d=json.load(open('$O/c5_${m}_$rep.json')); print('config 5 $m:', '%.4e'%d['value'], round(d['ms_per_step'],4), round(d['roofline']['avg_kernel_ms'],4))"
  done
done
