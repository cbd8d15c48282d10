// How fast can 5 x 2^20 rows of B bytes be written to / read from pseudo-random row positions of one array?
// (K5's conflict rows: one 32-byte row per (command, replica), written by the scan in (key, delivery order).)
// hipcc --offload-arch=gfx950 -O3 random_rows.hip -o random_rows && ./random_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr long long N = 5ll << 20;

__device__ __forceinline__ long long where(long long i) { return (i * 2654435761ll + 12345ll) % N; }  // a bijection: 2654435761 is odd and not a multiple of 5

template <int W, bool RANDOM>  // W int4 per row
__global__ void k_write(int4* dst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const long long p = RANDOM ? where(i) : i;
  for (int j = 0; j < W; ++j) dst[p * W + j] = make_int4((int)i, j, 2, 3);
}
template <int W, bool RANDOM>
__global__ void k_read(const int4* src, int* sink) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const long long p = RANDOM ? where(i) : i;
  int s = 0;
  for (int j = 0; j < W; ++j) {
    const int4 v = src[p * W + j];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 0x7fffffff) sink[0] = s;
}

template <typename F> float timeit(F f) {
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / 20 * 1000;
}

int main() {
  int4* buf;
  int* sink;
  hipMalloc(&buf, N * 64);
  hipMalloc(&sink, 4);
  hipMemset(buf, 0, N * 64);
  const int grid = (int)((N + 255) / 256);
#define RUN(W, R, K, name) printf("%-8s rows of %2d B %-10s %7.1f us\n", name, W * 16, R ? "random" : "in order", \
  timeit([&] { hipLaunchKernelGGL((K<W, R>), dim3(grid), dim3(256), 0, 0, buf ARGS); }))
#define ARGS
  RUN(1, true, k_write, "write"); RUN(2, true, k_write, "write"); RUN(4, true, k_write, "write");
  RUN(1, false, k_write, "write"); RUN(2, false, k_write, "write"); RUN(4, false, k_write, "write");
#undef ARGS
#define ARGS , sink
  RUN(1, true, k_read, "read"); RUN(2, true, k_read, "read"); RUN(4, true, k_read, "read");
  RUN(1, false, k_read, "read"); RUN(2, false, k_read, "read"); RUN(4, false, k_read, "read");
  return 0;
}
