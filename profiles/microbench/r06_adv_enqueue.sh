cd /root/repo
python bench.py --config adversarial --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['config']['host_enqueue_ms_per_pass'])"
FPX_P1A_SPLIT=1 python bench.py --config adversarial --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('split', d['value'], d['ms_per_step'], d['config']['host_enqueue_ms_per_pass'])"
