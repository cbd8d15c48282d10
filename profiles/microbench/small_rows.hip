// What do 2^21 small rows cost on this chip?  One lane per row, int4 stores: rows of 16 or 32 bytes at strides of
// 16 .. 128 bytes, into one or two arrays -- the access patterns of small acceptor groups (R <= 4) when every other
// slot belongs to somebody else.  hipcc -O3 --offload-arch=gfx950 small_rows.hip -o /tmp/small_rows && /tmp/small_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int ROW16, int ARRAYS>
__global__ void __launch_bounds__(256) k_rows(int4* a, int4* b, int stride16, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int4 v = make_int4((int)i, 1, 2, 3);
#pragma unroll
    for (int q = 0; q < ROW16; ++q) {
      a[i * stride16 + q] = v;
      if (ARRAYS == 2) b[i * stride16 + q] = v;
    }
  }
}
// cold = 1: every repetition writes a part of the arena nothing has touched since it was zeroed long ago (what a log
// window that moves on does); cold = 0: the same 2^21 rows again and again (they stay in the 256 MB Infinity Cache)
static int cold = 1;
template <int ROW16, int ARRAYS>
float run(int4* a0, int4* b0, int stride16, long n, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  float best = 1e9f;
  static long at = 0;  // in int4 units
  for (int rep = 0; rep < 5; ++rep) {
    const long span = n * stride16;
    if (at + span > (1l << 30) / 16 * 6) at = 0;  // 6 GiB per array
    int4 *a = a0 + (cold ? at : 0), *b = b0 + (cold ? at : 0);
    at += span;
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_rows<ROW16, ARRAYS>), dim3(grid), dim3(256), 0, 0, a, b, stride16, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  return best;
}
int main() {
  const long n = 1l << 21;
  const size_t bytes = (size_t)6 << 30;
  int4 *a, *b;
  hipMalloc(&a, bytes), hipMalloc(&b, bytes);
  hipMemset(a, 0, bytes), hipMemset(b, 0, bytes);
  hipDeviceSynchronize();
  for (cold = 1; cold >= 0; --cold) {
  printf("%s: 2^21 rows, one lane per row; us and GB/s of USEFUL bytes\n", cold ? "COLD (fresh memory every launch)" : "WARM (the same rows again)");
  for (int grid : {2048}) {
    for (int stride16 : {1, 2, 4, 8}) {
      float t;
      t = run<1, 1>(a, b, stride16, n, grid);
      printf("grid %5d  16-B rows, stride %3d B, 1 array : %7.1f us  %6.0f GB/s\n", grid, stride16 * 16, t * 1e3, n * 16 / t / 1e6);
      t = run<1, 2>(a, b, stride16, n, grid);
      printf("grid %5d  16-B rows, stride %3d B, 2 arrays: %7.1f us  %6.0f GB/s\n", grid, stride16 * 16, t * 1e3, n * 32 / t / 1e6);
      if (stride16 >= 2) {
        t = run<2, 1>(a, b, stride16, n, grid);
        printf("grid %5d  32-B rows, stride %3d B, 1 array : %7.1f us  %6.0f GB/s\n", grid, stride16 * 16, t * 1e3, n * 32 / t / 1e6);
      }
    }
  }
  }
  return 0;
}
