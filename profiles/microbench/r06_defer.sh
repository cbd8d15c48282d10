# r06: the fold of a K3 launch riding in the next launch (k_phase2_fin) -- parity, then A/B by FPX_NO_DEFER_FINALIZE
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_next_rows.py tests/test_gpu_sharding.py tests/test_mencius_noop_range.py tests/test_stress_scripts.py -q -x 2>&1 | tail -4
for mode in defer nodefer defer nodefer; do
  if [ $mode = nodefer ]; then export FPX_NO_DEFER_FINALIZE=1; else unset FPX_NO_DEFER_FINALIZE; fi
  python bench.py --no-cpu-baseline --configs-block-steps 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode headline value %.4g ms_per_step %.4f kernel %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms']))"
done
for mode in defer nodefer; do
  if [ $mode = nodefer ]; then export FPX_NO_DEFER_FINALIZE=1; else unset FPX_NO_DEFER_FINALIZE; fi
  for c in 2 3 adversarial host_path thrifty_random; do
    python bench.py --config $c --no-cpu-baseline --steps 20 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode config $c value %.4g ms_per_step %.5f' % (d['value'], d['ms_per_step']))"
  done
done
