# r06: the LSD sort's tile size for the dependency graph's 1 M pairs (RS_ITEMS 16 -> 8: 256 -> 512 tiles)
R=$PWD
for lib in default items8; do
  if [ $lib = items8 ]; then cp frankenpaxos_amd/csrc/libfpx.so /tmp/keep.so; cp profiles/microbench/build/libfpx_items8.so frankenpaxos_amd/csrc/libfpx.so; fi
  echo "== $lib"
  python profiles/microbench/depgraph_dev_bench.py 20 2>&1 | tail -2
  python bench.py --config 4_execute --no-cpu-baseline --steps 10 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4_execute kernel', d['roofline']['avg_kernel_ms'])"
  FPX_EPX_V1=1 python bench.py --config 4 --no-cpu-baseline --steps 6 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 4 first form ms', d['roofline']['avg_kernel_ms'])"
  if [ $lib = items8 ]; then python -m pytest tests/test_depgraph_dev.py tests/test_epaxos.py -q -x 2>&1 | tail -2; cp /tmp/keep.so frankenpaxos_amd/csrc/libfpx.so; fi
done
