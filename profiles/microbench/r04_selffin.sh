# the self-finalising dense launches: whole GPU suite, then A/B on the headline and on config 5
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
for v in 0 1; do
  echo "== FPX_NO_SELF_FINALIZE=$v"
  if [ $v = 1 ]; then export FPX_NO_SELF_FINALIZE=1; else unset FPX_NO_SELF_FINALIZE; fi
  python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c 'import sys,json; l=json.loads(sys.stdin.read()); print("headline", l["value"], l["ms_per_step"], l["roofline"].get("avg_kernel_ms"))'
  for c in 2 3 5; do python bench.py --config $c --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c 'import sys,json; l=json.loads(sys.stdin.read()); print("config", sys.argv[1], l["value"], l["ms_per_step"])' $c; done
done
