# variants of k_ranges_fill_rows with parts switched off (timing only); build here, run on the GPU box: bash cfg5_variants.sh run
set -u
cd "$(dirname "$0")/../.."
V="base"
if [ "${1:-build}" = build ]; then
  mkdir -p profiles/microbench/build
  for v in $V; do
    D=""
    (cd frankenpaxos_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC $D -c -o /tmp/api_$v.o fpx_api.hip &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../profiles/microbench/build/libfpx5_$v.so /tmp/api_$v.o fpx_epaxos.o fpx_wire.o fpx_depgraph.o -ldl) &
  done
  wait; ls profiles/microbench/build
else
  for v in $V; do for il in 0; do echo "$v interleave=$il: $(FPX_INTERLEAVE=$il FPX_LIB=$PWD/profiles/microbench/build/libfpx5_$v.so timeout 200 python bench.py --config 5 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c 'import sys,json; l=json.loads(sys.stdin.read()); print(l["ms_per_step"], l["roofline"]["avg_kernel_ms"])')"; done; done
fi
