# builds variants of libfpx.so with parts of k_epx_key2 switched off (-DKP_X_*: timing only, results are wrong) -- run
# here (hipcc cross-compiles), the .so files travel to the GPU box with the snapshot; then: bash k5v2_variants.sh run
set -u
cd "$(dirname "$0")/../.."
V="base NOOUT NOSORT NOSCAN MB8 MB2"
if [ "${1:-build}" = build ]; then
  mkdir -p profiles/microbench/build
  for v in $V; do
    D=""; [ $v != base ] && D="-DKP_X_$v"; [ $v = MB8 ] && D="-DKP_SCATTER_MB=8"; [ $v = MB2 ] && D="-DKP_SCATTER_MB=2"
    (cd frankenpaxos_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-result $D ${EXTRA:-} -c -o /tmp/epx_$v.o fpx_epaxos.hip &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../profiles/microbench/build/libfpx_$v.so fpx_api.o /tmp/epx_$v.o fpx_wire.o fpx_depgraph.o -ldl) &
  done
  wait; ls -la profiles/microbench/build
else
  mkdir -p gpurun_out/k5v2
  for v in $V; do FPX_LIB=$PWD/profiles/microbench/build/libfpx_$v.so timeout 200 python profiles/microbench/k5v2_time.py $v 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/k5v2/variants.txt
fi
