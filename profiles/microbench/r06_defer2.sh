python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_next_rows.py tests/test_mencius_noop_range.py tests/test_stress_scripts.py -q -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for mode in defer nodefer defer nodefer; do
  if [ $mode = nodefer ]; then export FPX_NO_DEFER_FINALIZE=1; else unset FPX_NO_DEFER_FINALIZE; fi
  for c in 2 3; do
    python bench.py --config $c --no-cpu-baseline --steps 200 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode config $c value %.4g ms_per_step %.5f kernel %.5f' % (d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms']))"
  done
done
