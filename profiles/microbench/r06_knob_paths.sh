cd /root/repo
for e in "FPX_P1A_SPLIT=1" "FPX_NO_DEFER_FINALIZE=1" "FPX_P1A_SPLIT=1 FPX_NO_DEFER_FINALIZE=1"; do
  echo "== $e"; env $e timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not fold_that_rides and not phase1a_dev_is_asynchronous" 2>&1 | grep -E "passed|failed|error" | tail -2
done
