// r05: WHAT is a "slow" placement of the cell slab?  
//
// The hot kernel streams row s of three arrays in lockstep (read ballot, write vote_round, write vote_value; 1 KiB per
// row and array).  One and the same binary runs 0.535 - 0.607 ms per 2^20 rows depending on which physical pages the slab
// got (profiles/r02_placement.txt, r04_kernel_stats.csv).  This program measures, on one box and in one process:
//
//   A. a hipMalloc'ed slab of W windows x 3 arrays: the triple stream per WINDOW and per EIGHTH of a window -- how coarse
//      is the slow / fast pattern inside one allocation?  and the three streams of a window one at a time -- does the
//      slowness belong to one array's pages or to the combination?
//   B. the same slab built from hipMemCreate chunks mapped into one reserved range: every chunk alone (read, write), the
//      triple stream per window; slow windows re-drawn (new physical chunks, the old ones held until the end so that the
//      allocator cannot hand them back) -- does re-drawing help, and how many draws does it take?
//
// build: hipcc -O3 --offload-arch=gfx950 -o r05_placement r05_placement.hip ; run: ./r05_placement [windows=25] [chunk_mib=1024]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

typedef int v4 __attribute__((ext_vector_type(4)));

// one wavefront per 32 consecutive 1 KiB rows, 16 B per lane: the hot kernel's access pattern; any pointer may be null
// rs = row stride in 16-byte units (64 = rows back to back; 128 / 192 = two / three arrays interleaved row by row)
__global__ void __launch_bounds__(256) k_stream(const v4* a_read, v4* b_write, v4* c_write, int rows, int rs, int rs_a) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int first = wave * 32;
  v4 acc = {0, 0, 0, 0};
  for (int i = first; i < first + 32 && i < rows; ++i) {
    const size_t o = (size_t)i * rs + lane;
    v4 v = {i, i, i, i};
    if (a_read) v = __builtin_nontemporal_load(a_read + (size_t)i * rs_a + lane);
    if (b_write) __builtin_nontemporal_store(v, b_write + o);
    if (c_write) __builtin_nontemporal_store(v, c_write + o);
    if (!b_write && !c_write) acc += v;
  }
  if (!b_write && !c_write && acc.x == 0x7fffffff && a_read) *(volatile int*)a_read;  // keep the loads
}

static hipEvent_t e0, e1;
static int g_rs = 64, g_rs_a = 64;
static float time_stream(const void* a, void* b, void* c, int rows, int reps = 4) {
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_stream, dim3((rows + 127) / 128), dim3(256), 0, 0, (const v4*)a, (v4*)b, (v4*)c, rows, g_rs, g_rs_a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0) best = std::min(best, ms);
  }
  return best;
}

static void stats(const char* name, const std::vector<float>& v) {
  std::vector<float> s(v);
  std::sort(s.begin(), s.end());
  double sum = 0;
  for (float x : s) sum += x;
  printf("%s: n=%zu min %.4f med %.4f max %.4f mean %.4f (max/min %.3f)\n", name, s.size(), s.front(), s[s.size() / 2], s.back(),
         sum / s.size(), s.back() / s.front());
}

int main(int argc, char** argv) {
  const int W = argc > 1 ? atoi(argv[1]) : 25;
  const size_t chunk = (size_t)(argc > 2 ? atoi(argv[2]) : 1024) << 20;
  const size_t GiB = (size_t)1 << 30;
  const int ROWS = 1 << 20;  // rows of 1 KiB per window and array
  CK(hipSetDevice(0));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  size_t free_b = 0, total_b = 0;
  CK(hipMemGetInfo(&free_b, &total_b));
  printf("device 0: free %.1f GiB of %.1f; W = %d windows, chunk = %zu MiB\n", free_b / 1073741824.0, total_b / 1073741824.0, W,
         chunk >> 20);

  if (argc > 3 && atoi(argv[3]) == 2) {
    // ---------------------------------------------------------------- M: a map of the triple stream over ONE big allocation
    // positions are GiB offsets into a P-GiB hipMalloc; T(b, c, a) = write vote_round at b, write vote_value at c, read ballot at a
    const int P = W;  // argv[1] = GiB to allocate
    char* slab = nullptr;
    CK(hipMalloc((void**)&slab, (size_t)P * GiB));
    printf("\nM. hipMalloc %d GiB at %p\n", P, (void*)slab);
    auto T = [&](int b, int c, int a) { return time_stream(a < 0 ? nullptr : slab + a * GiB, b < 0 ? nullptr : slab + b * GiB, c < 0 ? nullptr : slab + c * GiB, ROWS, 3); };
    printf("  M1 neighbours (b=i, c=i+1, a=i+2), i = 0, 3, ..:");
    for (int i = 0; i + 2 < P; i += 3) printf(" %.4f", T(i, i + 1, i + 2));
    printf("\n  M2 b=0, a=2, c=j, j = 3, 5, ..:");
    for (int j = 3; j < P; j += 2) printf(" %.4f", T(0, j, 2));
    printf("\n  M3 b=0, c=1, a=k, k = 3, 5, ..:");
    for (int k = 3; k < P; k += 2) printf(" %.4f", T(0, 1, k));
    printf("\n  M4 c=1, a=2, b=i, i = 3, 5, ..:");
    for (int i = 3; i < P; i += 2) printf(" %.4f", T(i, 1, 2));
    printf("\n  M5 writes only, b=i, c=i+1, i = 0, 2, ..:");
    for (int i = 0; i + 1 < P; i += 2) printf(" %.4f", T(i, i + 1, -1));
    printf("\n  M6 writes only, b=0, c=j, j = 1, 3, ..:");
    for (int j = 1; j < P; j += 2) printf(" %.4f", T(0, j, -1));
    printf("\n  M7 one write + read, b=i, a=i+1, i = 0, 2, ..:");
    for (int i = 0; i + 1 < P; i += 2) printf(" %.4f", T(i, -1, i + 1));
    printf("\n  M8 the slab's shape (arrays D GiB apart: b=w, c=D+w, a=2D+w), w = 0..D-1, for D = 25, 40, 64, 80:\n");
    for (int D : {25, 40, 64, 80}) {
      if (3 * D > P) continue;
      printf("    D=%d:", D);
      for (int w = 0; w < D; w += (D > 40 ? 3 : 1)) printf(" %.4f", T(w, D + w, 2 * D + w));
      printf("\n");
    }
    if (argc > 4) {
      const int r1 = atoi(argv[4]);  // a chunk of the OTHER class than chunk 0 (48 on the boxes seen so far)
      printf("  M10 b=0, c=%d, a=k, k = 1, 3, ..:", r1);
      for (int k = 1; k < P; k += 2) if (k != r1) printf(" %.4f", T(0, r1, k));
      printf("\n  M11 writes only, b=%d, c=j, j = 0, 2, ..:", r1);
      for (int j = 0; j < P; j += 2) if (j != r1) printf(" %.4f", T(r1, j, -1));
      printf("\n  M12 writes only, b=20, c=j, j = 1, 3, ..:");
      for (int j = 1; j < P; j += 2) printf(" %.4f", T(20, j, -1));
      printf("\n  M13 writes only, b=100, c=j, j = 1, 3, ..:");
      for (int j = 1; j < P; j += 2) printf(" %.4f", T(100, j, -1));
      printf("\n  M14 read + write, a=0, b=j, j = 1, 3, ..:");
      for (int j = 1; j < P; j += 2) printf(" %.4f", T(j, -1, 0));
      printf("\n  M15 writes only at 256 MiB granularity, b=0, c = 40 GiB + q x 256 MiB, q = 0..63:");
      for (int q = 0; q < 64; ++q) printf(" %.4f", time_stream(nullptr, slab, slab + 40 * GiB + ((size_t)q << 28), ROWS / 4, 4));
      printf("\n");
    }
    printf("  M9 sub-GiB offsets of c against b (b=10, a=12, c=11 GiB + d MiB), d = 0, 2, 4, .. 62:");
    for (int d = 0; d < 64; d += 2) printf(" %.4f", time_stream(slab + 12 * GiB, slab + 10 * GiB, slab + 11 * GiB + ((size_t)d << 20), ROWS - 65536, 3));
    printf("\n");
    CK(hipFree(slab));
    return 0;
  }
  // ---------------------------------------------------------------- A: hipMalloc slab
  for (int round = 0; round < 2; ++round) {
    char* slab = nullptr;
    const size_t stride = (size_t)W * GiB;
    CK(hipMalloc((void**)&slab, stride * 3));
    printf("\nA%d. hipMalloc slab %p (%.1f GiB), arrays %zu GiB apart\n", round, (void*)slab, stride * 3 / 1073741824.0, stride >> 30);
    std::vector<float> tw(W);
    printf("  triple stream per window [ms]:");
    for (int w = 0; w < W; ++w) {
      tw[w] = time_stream(slab + 2 * stride + w * GiB, slab + w * GiB, slab + stride + w * GiB, ROWS);
      printf(" %.4f", tw[w]);
    }
    printf("\n");
    stats("  A windows", tw);
    const int slow = (int)(std::max_element(tw.begin(), tw.end()) - tw.begin()), fast = (int)(std::min_element(tw.begin(), tw.end()) - tw.begin());
    for (int w : {slow, fast}) {
      printf("  window %d (%s, %.4f ms): eighths [ms]", w, w == slow ? "slowest" : "fastest", tw[w]);
      for (int p = 0; p < 8; ++p) {
        const size_t o = w * GiB + p * (GiB / 8);
        printf(" %.4f", time_stream(slab + 2 * stride + o, slab + o, slab + stride + o, ROWS / 8, 6));
      }
      printf("\n    single streams: read ballot %.4f  write vote_round %.4f  write vote_value %.4f | read vr %.4f read vv %.4f write ballot %.4f\n",
             time_stream(slab + 2 * stride + w * GiB, nullptr, nullptr, ROWS), time_stream(nullptr, slab + w * GiB, nullptr, ROWS),
             time_stream(nullptr, nullptr, slab + stride + w * GiB, ROWS), time_stream(slab + w * GiB, nullptr, nullptr, ROWS),
             time_stream(slab + stride + w * GiB, nullptr, nullptr, ROWS), time_stream(nullptr, slab + 2 * stride + w * GiB, nullptr, ROWS));
      // the combination: this window's ballot with ANOTHER window's vote arrays
      const int o2 = w == slow ? fast : slow;
      printf("    crossed: read(w) + writes(other) %.4f ; read(other) + writes(w) %.4f\n",
             time_stream(slab + 2 * stride + w * GiB, slab + o2 * GiB, slab + stride + o2 * GiB, ROWS),
             time_stream(slab + 2 * stride + o2 * GiB, slab + w * GiB, slab + stride + w * GiB, ROWS));
    }
    // the two vote arrays interleaved row by row (2 KiB per slot) in the slab's first 2 W GiB, the ballots apart; and all
    // three interleaved (3 KiB per slot: [vote_round | vote_value | ballot]) -- the same physical pages, another layout
    {
      std::vector<float> t1(W), t2(W), t3(W);
      g_rs = 128;
      printf("  L1 vote rows interleaved, ballots apart:");
      for (int w = 0; w < W; ++w) printf(" %.4f", t1[w] = time_stream(slab + 2 * stride + w * GiB, slab + w * 2 * GiB, slab + w * 2 * GiB + 1024, ROWS));
      printf("\n");
      stats("  L1 windows", t1);
      g_rs = 192, g_rs_a = 192;
      printf("  L2 all three interleaved:");
      for (int w = 0; w < W; ++w) printf(" %.4f", t2[w] = time_stream(slab + w * 3 * GiB + 2048, slab + w * 3 * GiB, slab + w * 3 * GiB + 1024, ROWS));
      printf("\n");
      stats("  L2 windows", t2);
      printf("  L2b [ballot | vote_round | vote_value]:");
      for (int w = 0; w < W; ++w) printf(" %.4f", t3[w] = time_stream(slab + w * 3 * GiB, slab + w * 3 * GiB + 1024, slab + w * 3 * GiB + 2048, ROWS));
      printf("\n");
      stats("  L2b windows", t3);
      g_rs = 64, g_rs_a = 64;
    }
    // separate arrays, vote_value shifted by a stagger (bytes) against vote_round: slowest and fastest window
    for (int w : {slow, fast}) {
      printf("  window %d, vote_value staggered by:", w);
      for (size_t d : {(size_t)0, (size_t)256, (size_t)1024, (size_t)2048, (size_t)4096, (size_t)8192, (size_t)16384, (size_t)32768, (size_t)65536,
                       (size_t)131072, (size_t)262144, (size_t)524288, (size_t)1048576, (size_t)2097152, (size_t)(4352), (size_t)(69632)})
        printf(" %zu:%.4f", d, time_stream(slab + 2 * stride + w * GiB, slab + w * GiB, slab + stride + w * GiB + d, ROWS - 4096));
      printf("\n");
    }
    // repeatability: the same windows again
    printf("  again:");
    std::vector<float> tw2(W);
    for (int w = 0; w < W; ++w) printf(" %.4f", tw2[w] = time_stream(slab + 2 * stride + w * GiB, slab + w * GiB, slab + stride + w * GiB, ROWS));
    printf("\n");
    CK(hipFree(slab));
  }

  if (argc > 3 && atoi(argv[3]) == 0) return 0;
  // ---------------------------------------------------------------- B: chunks
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  size_t gran_min = 0;
  CK(hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum));
  printf("\nB. hipMemCreate chunks: granularity min %zu KiB, recommended %zu KiB\n", gran_min >> 10, gran >> 10);
  const size_t total = (size_t)W * 3 * GiB;
  const int nchunk = (int)(total / chunk), per_win = (int)(GiB / chunk);  // chunks per window and array (chunk <= 1 GiB) -- or
  if (chunk > GiB || GiB % chunk) {
    printf("chunk must divide 1 GiB\n");
    return 1;
  }
  hipDeviceptr_t base = 0;
  CK(hipMemAddressReserve(&base, total, (size_t)1 << 30, 0, 0));
  std::vector<hipMemGenericAllocationHandle_t> h(nchunk);
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  float t_create = 0;
  {
    hipEvent_t a, b;
    (void)a, (void)b;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < nchunk; ++i) {
      CK(hipMemCreate(&h[i], chunk, &prop, 0));
      CK(hipMemMap((hipDeviceptr_t)((char*)base + (size_t)i * chunk), chunk, 0, h[i], 0));
    }
    CK(hipMemSetAccess(base, total, &acc, 1));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    t_create = (t1.tv_sec - t0.tv_sec) * 1e3f + (t1.tv_nsec - t0.tv_nsec) * 1e-6f;
  }
  printf("  created + mapped %d chunks in %.1f ms (%.2f ms per chunk)\n", nchunk, t_create, t_create / nchunk);
  char* slab = (char*)base;
  const size_t stride = (size_t)W * GiB;
  const int crow = (int)(chunk >> 10);
  if (nchunk <= 96) {
    std::vector<float> tr(nchunk), tq(nchunk);
    printf("  per chunk read [ms]:");
    for (int i = 0; i < nchunk; ++i) printf(" %.4f", tr[i] = time_stream(slab + (size_t)i * chunk, nullptr, nullptr, crow));
    printf("\n  per chunk write [ms]:");
    for (int i = 0; i < nchunk; ++i) printf(" %.4f", tq[i] = time_stream(nullptr, slab + (size_t)i * chunk, nullptr, crow));
    printf("\n");
    stats("  B chunk read", tr);
    stats("  B chunk write", tq);
  }
  std::vector<float> tw(W);
  auto win = [&](int w) { return time_stream(slab + 2 * stride + w * GiB, slab + w * GiB, slab + stride + w * GiB, ROWS); };
  printf("  triple stream per window [ms]:");
  for (int w = 0; w < W; ++w) printf(" %.4f", tw[w] = win(w));
  printf("\n");
  stats("  B windows", tw);
  printf("  again:");
  for (int w = 0; w < W; ++w) printf(" %.4f", win(w));
  printf("\n");

  // re-draw: windows slower than 1.02 x the fastest get new physical chunks for all three arrays; the old chunks are
  // unmapped but HELD (not released) so that the allocator cannot return them.  Up to 6 draws per window.
  const float tmin = *std::min_element(tw.begin(), tw.end()), bar = tmin * 1.02f;
  std::vector<hipMemGenericAllocationHandle_t> held;
  int draws = 0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int w = 0; w < W; ++w) {
    int tries = 0;
    while (tw[w] > bar && tries < 6) {
      CK(hipMemGetInfo(&free_b, &total_b));
      if (free_b < 3 * GiB + (8 * GiB)) break;
      for (int arr = 0; arr < 3; ++arr)
        for (int c = 0; c < per_win; ++c) {
          const size_t off = arr * stride + w * GiB + (size_t)c * chunk;
          const int i = (int)(off / chunk);
          CK(hipMemUnmap((hipDeviceptr_t)(slab + off), chunk));
          held.push_back(h[i]);
          CK(hipMemCreate(&h[i], chunk, &prop, 0));
          CK(hipMemMap((hipDeviceptr_t)(slab + off), chunk, 0, h[i], 0));
          CK(hipMemSetAccess((hipDeviceptr_t)(slab + off), chunk, &acc, 1));
        }
      const float t = win(w);
      printf("  window %d: %.4f -> %.4f (draw %d)\n", w, tw[w], t, tries + 1);
      tw[w] = t;
      ++tries, ++draws;
    }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  printf("  %d re-draws in %.1f ms, %zu chunks held\n", draws, (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6, held.size());
  printf("  after re-draws:");
  for (int w = 0; w < W; ++w) printf(" %.4f", tw[w] = win(w));
  printf("\n");
  stats("  B windows after re-draws", tw);
  // does only ONE array's chunk make a window slow?  swap experiments on the slowest window that is left
  {
    const int w = (int)(std::max_element(tw.begin(), tw.end()) - tw.begin());
    printf("  slowest window left %d (%.4f): read %.4f  write vr %.4f  write vv %.4f\n", w, tw[w],
           time_stream(slab + 2 * stride + w * GiB, nullptr, nullptr, ROWS), time_stream(nullptr, slab + w * GiB, nullptr, ROWS),
           time_stream(nullptr, nullptr, slab + stride + w * GiB, ROWS));
  }
  for (auto x : held) CK(hipMemRelease(x));
  CK(hipMemUnmap(base, total));
  for (auto x : h) CK(hipMemRelease(x));
  CK(hipMemAddressFree(base, total));
  printf("done\n");
  return 0;
}
