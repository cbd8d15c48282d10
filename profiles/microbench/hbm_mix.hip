// hbm_mix.hip -- what can HBM3E on this MI355X sustain for the access mixes of the Phase-2 kernel?
// (measurement aid, not part of the product).  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/hbm_mix profiles/microbench/hbm_mix.hip && /tmp/hbm_mix
//
// Patterns (16 B per lane, one wavefront moves one 1 KiB row, like k_phase2 at R = 256):
//   read      1 stream in
//   fill      1 stream out            (plain / nontemporal)
//   copy      1 in, 1 out
//   mix12     1 in, 2 out             (PER_SLOT model: ballot row in, voteRound + voteValue rows out)
//   fill2     2 out                   (ACCEPTOR model: voteRound + voteValue rows out)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int int4v __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                           \
  do {                                                                     \
    hipError_t e = (x);                                                    \
    if (e != hipSuccess) {                                                 \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                             \
    }                                                                      \
  } while (0)

template <int IN, int OUT, bool NT, int UNROLL>
__global__ void __launch_bounds__(256) k_mix(const int4v* __restrict__ a, int4v* __restrict__ o0, int4v* __restrict__ o1,
                                             size_t nvec, int* sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  int acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride * UNROLL) {
    int4v v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const size_t j = i + (size_t)u * stride;
      v[u] = int4v{1, 2, 3, 4};
      if (IN && j < nvec) v[u] = a[j];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const size_t j = i + (size_t)u * stride;
      if (j < nvec) {
        if (OUT >= 1) {
          if (NT) __builtin_nontemporal_store(v[u], o0 + j); else o0[j] = v[u];
        }
        if (OUT >= 2) {
          if (NT) __builtin_nontemporal_store(v[u] + 1, o1 + j); else o1[j] = v[u] + 1;
        }
        if (OUT == 0) acc += v[u][0] ^ v[u][3];
      }
    }
  }
  if (OUT == 0 && acc == 0x12345678) *sink = acc;
}

template <int IN, int OUT, bool NT, int UNROLL>
void run(const char* name, int4v* a, int4v* o0, int4v* o1, size_t bytes, int grid, int* sink) {
  const size_t nvec = bytes / 16;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_mix<IN, OUT, NT, UNROLL>), dim3(grid), dim3(256), 0, 0, a, o0, o1, nvec, sink);
  CHECK(hipEventRecord(e0));
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_mix<IN, OUT, NT, UNROLL>), dim3(grid), dim3(256), 0, 0, a, o0, o1, nvec, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double per = ms / reps * 1e-3;
  const double total = (double)bytes * (IN + OUT);
  printf("%-28s grid %5d unroll %d  %8.1f us  %7.1f GB/s total (%d in, %d out)\n", name, grid, UNROLL, per * 1e6,
         total / per / 1e9, IN, OUT);
}

int main() {
  const size_t bytes = (size_t)1 << 30;  // per stream
  int4v *a, *o0, *o1;
  int* sink;
  CHECK(hipMalloc(&a, bytes));
  CHECK(hipMalloc(&o0, bytes));
  CHECK(hipMalloc(&o1, bytes));
  CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(a, 1, bytes));
  CHECK(hipMemset(o0, 0, bytes));
  CHECK(hipMemset(o1, 0, bytes));
  for (int grid : {2048, 4096, 8192, 65536}) {
    run<1, 0, false, 4>("read", a, o0, o1, bytes, grid, sink);
    run<0, 1, false, 4>("fill plain", a, o0, o1, bytes, grid, sink);
    run<0, 1, true, 4>("fill nt", a, o0, o1, bytes, grid, sink);
    run<1, 1, false, 4>("copy plain", a, o0, o1, bytes, grid, sink);
    run<1, 1, true, 4>("copy nt", a, o0, o1, bytes, grid, sink);
    run<0, 2, false, 4>("fill2 plain", a, o0, o1, bytes, grid, sink);
    run<0, 2, true, 4>("fill2 nt", a, o0, o1, bytes, grid, sink);
    run<1, 2, false, 4>("mix12 plain", a, o0, o1, bytes, grid, sink);
    run<1, 2, true, 4>("mix12 nt", a, o0, o1, bytes, grid, sink);
  }
  run<1, 2, true, 1>("mix12 nt", a, o0, o1, bytes, 2048, sink);
  run<1, 2, true, 2>("mix12 nt", a, o0, o1, bytes, 2048, sink);
  run<1, 2, true, 8>("mix12 nt", a, o0, o1, bytes, 2048, sink);
  run<0, 2, true, 1>("fill2 nt", a, o0, o1, bytes, 2048, sink);
  run<0, 2, true, 8>("fill2 nt", a, o0, o1, bytes, 2048, sink);
  return 0;
}
