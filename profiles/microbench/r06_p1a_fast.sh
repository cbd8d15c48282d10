cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8
one() { python bench.py --config adversarial --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
  one fast
  FPX_P1A_SPLIT=1 one split
done
for i in 1 2; do python bench.py --steps 20 --warmup 5 --configs-block-steps 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
