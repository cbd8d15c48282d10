// fpx_epaxos.hip -- K5: the EPaxos pre-accept fast path on gfx950.
//
// Per replica the conflict scan is a segmented (by key) exclusive prefix-max over the tick's commands
// in that replica's delivery order, on n-wide watermark vectors (util/TopOne.scala): data-parallel as
//   1. k_epx_keys      scatter every command to its position in the replica's delivery order (rank is a
//                      permutation) + the tick's own TopOne contribution per key (atomicMax, once per command)
//   2. rocprim radix sort per replica on the key bits only (stable => (key, delivery order)); a plain
//                      library primitive, everything else is hand-written
//   3. k_epx_segments  [lo, hi) of every (replica, key) segment by binary search
//   4. k_epx_scan<N>   one wavefront per (replica, key): 64 commands per step, wave-level max-scan of the
//                      2N watermark columns with __shfl_up, carry in registers
//   5. k_epx_decide<N> one thread per command: PreAcceptOk = local conflicts U leader's deps; fast path iff
//                      the n-2 answers agree (Util.popularItems), else the union (preAcceptingSlowPath)
//   6. k_epx_commit    every replica's conflict index learns the tick's instances (commit ->
//                      updateConflictIndex): elementwise max with the per-key tick table
// Integer max / compare only: HBM- and latency-bound, no MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <rocprim/rocprim.hpp>
#include <vector>

#include "../../include/fpx.h"

namespace {

struct EpxState {
  int n, num_keys;
  int32_t* gets;  // [n][num_keys][n]
  int32_t* sets;  // [n][num_keys][n]
  int32_t* status;
};

struct EpxBatch {
  int m;
  const int32_t* leader;
  const int32_t* number;
  const int32_t* key;
  const uint8_t* is_set;
  const uint8_t* resp_mask;
  const int32_t* rank;   // [n][m]
  uint32_t* sk;          // [n][m] sort keys: the command's key, in the replica's delivery order
  int32_t* sv;           // [n][m] sort values (message index)
  uint32_t* sk_sorted;   // [n][m]
  int32_t* tick;         // [num_keys][2][n] the tick's own TopOne contribution per key (gets, sets)
  int32_t* sv_sorted;    // [n][m]
  int32_t* seg;          // [n][num_keys][2]
  int32_t* conf;         // [m][n][n] local conflicts of replica r for message i
  uint8_t* fast;
  int32_t* deps;
  int32_t* leader_deps;
};

__device__ __forceinline__ void epx_report(int32_t* status, int code, int index) {
  if (atomicCAS(&status[0], 0, code) == 0) status[1] = index;
}

__global__ void __launch_bounds__(256) k_epx_keys(const EpxState st, const EpxBatch b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.m) return;
  const int n = st.n;
  const int L = b.leader[i], k = b.key[i];
  const unsigned mask = b.resp_mask[i];
  bool ok = L >= 0 && L < n && b.number[i] >= 0 && k >= 0 && k < st.num_keys;
  ok = ok && !((mask >> (ok ? L : 0)) & 1u) && (mask >> n) == 0 && __popc(mask) == n - 2;
  for (int r = 0; r < n; ++r) {
    const int p = b.rank[(size_t)r * b.m + i];
    ok = ok && p >= 0 && p < b.m;
    const bool part = ok && (r == L || ((mask >> r) & 1u));
    // rank is a permutation: scattering to position p lays the tick out in replica r's delivery order;
    // a stable sort by key alone then yields (key, delivery order).  Non-participants sort last.
    if (ok) {
      b.sk[(size_t)r * b.m + p] = part ? (uint32_t)k : (uint32_t)st.num_keys;
      b.sv[(size_t)r * b.m + p] = i;
    }
  }
  if (!ok) {
    epx_report(st.status, FPX_EINVAL, i);
    return;
  }
  // what every replica's conflict index learns from this tick (commit -> updateConflictIndex): once per
  // key, not once per replica
  atomicMax(&b.tick[((size_t)k * 2 + (b.is_set[i] ? 1 : 0)) * n + L], b.number[i] + 1);
}

__global__ void __launch_bounds__(256) k_epx_segments(const EpxState st, const EpxBatch b) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= st.n * st.num_keys) return;
  const int r = t / st.num_keys, k = t % st.num_keys;
  const uint32_t* a = b.sk_sorted + (size_t)r * b.m;
  auto lower = [&](uint32_t x) {
    int lo = 0, hi = b.m;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  b.seg[(size_t)t * 2] = lower((uint32_t)k);
  b.seg[(size_t)t * 2 + 1] = lower((uint32_t)(k + 1));
}

// one wavefront per (replica, key) segment
template <int N>
__global__ void __launch_bounds__(256) k_epx_scan(const EpxState st, const EpxBatch b) {
  if (st.status[0] != 0) return;
  const int lane = threadIdx.x & 63;
  const int seg = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (seg >= N * st.num_keys) return;
  const int r = seg / st.num_keys, k = seg % st.num_keys;
  const int lo = b.seg[(size_t)seg * 2], hi = b.seg[(size_t)seg * 2 + 1];
  const int32_t* sv = b.sv_sorted + (size_t)r * b.m;
  int cg[N], cs[N];  // carry: the replica's TopOne vectors for this key (KeyValueStore.scala:229-230)
  const size_t ib = ((size_t)r * st.num_keys + k) * N;
#pragma unroll
  for (int l = 0; l < N; ++l) cg[l] = st.gets[ib + l], cs[l] = st.sets[ib + l];
  for (int base = lo; base < hi; base += 64) {
    const int p = base + lane;
    const bool valid = p < hi;
    const int i = valid ? sv[p] : 0;
    const int L = b.leader[i];
    const int id1 = b.number[i] + 1;  // TopOne.put: max(.., id + 1), util/TopOne.scala:12-15
    const bool t = b.is_set[i] != 0;
    int ig[N], is[N];  // inclusive prefix maxima of this chunk
#pragma unroll
    for (int l = 0; l < N; ++l) {
      ig[l] = (valid && !t && L == l) ? id1 : 0;
      is[l] = (valid && t && L == l) ? id1 : 0;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
      for (int l = 0; l < N; ++l) {
        const int og = __shfl_up(ig[l], d), os = __shfl_up(is[l], d);
        if (lane >= d) {
          ig[l] = og > ig[l] ? og : ig[l];
          is[l] = os > is[l] ? os : is[l];
        }
      }
    }
#pragma unroll
    for (int l = 0; l < N; ++l) {
      // exclusive prefix (the command's own put comes after its conflict lookup) + carry
      int eg = __shfl_up(ig[l], 1), es = __shfl_up(is[l], 1);
      if (lane == 0) eg = 0, es = 0;
      eg = eg > cg[l] ? eg : cg[l];
      es = es > cs[l] ? es : cs[l];
      // KeyValueStore.scala:259-302: a get conflicts with sets, a set with sets and gets
      const int dep = t ? (es > eg ? es : eg) : es;
      if (valid) b.conf[((size_t)i * N + r) * N + l] = dep;
      const int tg = __shfl(ig[l], 63), ts = __shfl(is[l], 63);
      cg[l] = tg > cg[l] ? tg : cg[l];
      cs[l] = ts > cs[l] ? ts : cs[l];
    }
  }
}

template <int N>
__global__ void __launch_bounds__(256) k_epx_decide(const EpxState st, const EpxBatch b) {
  if (st.status[0] != 0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.m) return;
  const int L = b.leader[i];
  const unsigned mask = b.resp_mask[i];
  const int32_t* c = b.conf + (size_t)i * N * N;
  int D[N], uni[N], first[N];
#pragma unroll
  for (int l = 0; l < N; ++l) D[l] = c[L * N + l], uni[l] = D[l], first[l] = 0;
  bool have_first = false, all_equal = true;
  for (int r = 0; r < N; ++r) {
    if (!((mask >> r) & 1u)) continue;
    bool same = true;
#pragma unroll
    for (int l = 0; l < N; ++l) {
      const int cl = c[r * N + l];
      const int resp = cl > D[l] ? cl : D[l];  // handlePreAccept: local conflicts U preAccept.dependencies
      uni[l] = resp > uni[l] ? resp : uni[l];  // preAcceptingSlowPath: union of all answers
      if (!have_first) first[l] = resp; else same = same && (first[l] == resp);
    }
    if (have_first) all_equal = all_equal && same;
    have_first = true;
  }
  if (b.fast) b.fast[i] = all_equal ? 1 : 0;
#pragma unroll
  for (int l = 0; l < N; ++l) {
    if (b.deps) b.deps[(size_t)i * N + l] = all_equal ? first[l] : uni[l];
    if (b.leader_deps) b.leader_deps[(size_t)i * N + l] = D[l];
  }
}

__global__ void __launch_bounds__(256) k_epx_commit(const EpxState st, const EpxBatch b) {
  if (st.status[0] != 0) return;
  const long long per = (long long)st.num_keys * st.n;            // entries of one replica's gets (or sets)
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per * st.n) return;
  const int r = (int)(t / per);
  const long long e = t % per;                                    // key * n + leader
  const int k = (int)(e / st.n), l = (int)(e % st.n);
  const int tg = b.tick[((size_t)k * 2 + 0) * st.n + l], ts = b.tick[((size_t)k * 2 + 1) * st.n + l];
  int32_t* g = &st.gets[(size_t)r * per + e];
  int32_t* s2 = &st.sets[(size_t)r * per + e];
  if (tg > *g) *g = tg;
  if (ts > *s2) *s2 = ts;
}

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace

struct fpx_epx {
  fpx_epx_config cfg;
  EpxState st;
  hipStream_t stream = nullptr, own_stream = nullptr;
  int last_hip = 0;
  Buf sk, sv, sk2, sv2, seg, conf, tmp, tick, h_leader, h_number, h_key, h_set, h_mask, h_rank, o_fast, o_deps, o_ldeps;
};

namespace {

#define EHIP(e, expr)                                            \
  do {                                                           \
    hipError_t _x = (expr);                                      \
    if (_x != hipSuccess) {                                      \
      (e)->last_hip = (int)_x;                                   \
      return _x == hipErrorOutOfMemory ? FPX_ENOMEM : FPX_EHIP;  \
    }                                                            \
  } while (0)

int grow(fpx_epx* e, Buf* b, size_t bytes) {
  if (bytes <= b->cap) return FPX_OK;
  if (b->p) EHIP(e, hipFree(b->p));
  b->p = nullptr, b->cap = 0;
  EHIP(e, hipMalloc(&b->p, std::max<size_t>(bytes, 256)));
  b->cap = std::max<size_t>(bytes, 256);
  return FPX_OK;
}

template <int N>
void launch_scan_decide(fpx_epx* e, const EpxBatch& b) {
  const int segs = N * e->st.num_keys;
  hipLaunchKernelGGL((k_epx_scan<N>), dim3((segs + 3) / 4), dim3(256), 0, e->stream, e->st, b);
  hipLaunchKernelGGL((k_epx_decide<N>), dim3((b.m + 255) / 256), dim3(256), 0, e->stream, e->st, b);
}

}  // namespace

extern "C" {

int32_t fpx_epx_create(const fpx_epx_config* cfg, fpx_epx** out) {
  if (!cfg || !out) return FPX_EINVAL;
  *out = nullptr;
  const int n = cfg->num_replicas;
  if (!(n == 3 || n == 5 || n == 7) || cfg->num_keys < 1 || cfg->num_keys > (1 << 24)) return FPX_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev)
    return FPX_ENODEVICE;
  fpx_epx* e = new (std::nothrow) fpx_epx();
  if (!e) return FPX_ENOMEM;
  e->cfg = *cfg;
  e->st.n = n;
  e->st.num_keys = cfg->num_keys;
  e->st.gets = e->st.sets = e->st.status = nullptr;
  auto fail = [&](int code) {
    fpx_epx_destroy(e);
    return code;
  };
  if (hipSetDevice(cfg->device) != hipSuccess) return fail(FPX_ENODEVICE);
  if (hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) != hipSuccess) return fail(FPX_EHIP);
  e->stream = e->own_stream;
  const size_t cells = (size_t)n * cfg->num_keys * n;
  if (hipMalloc((void**)&e->st.gets, cells * 4) != hipSuccess) return fail(FPX_ENOMEM);
  if (hipMalloc((void**)&e->st.sets, cells * 4) != hipSuccess) return fail(FPX_ENOMEM);
  if (hipMalloc((void**)&e->st.status, 32) != hipSuccess) return fail(FPX_ENOMEM);
  // TopOne.scala:10: every watermark starts at 0
  if (hipMemsetAsync(e->st.gets, 0, cells * 4, e->stream) != hipSuccess) return fail(FPX_EHIP);
  if (hipMemsetAsync(e->st.sets, 0, cells * 4, e->stream) != hipSuccess) return fail(FPX_EHIP);
  if (hipMemsetAsync(e->st.status, 0, 32, e->stream) != hipSuccess) return fail(FPX_EHIP);
  if (hipStreamSynchronize(e->stream) != hipSuccess) return fail(FPX_EHIP);
  *out = e;
  return FPX_OK;
}

int32_t fpx_epx_destroy(fpx_epx* e) {
  if (!e) return FPX_EINVAL;
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  void* ps[] = {e->st.gets, e->st.sets, e->st.status};
  for (void* p : ps)
    if (p) (void)hipFree(p);
  Buf* bs[] = {&e->sk, &e->sv, &e->sk2, &e->sv2, &e->seg, &e->conf, &e->tmp, &e->tick, &e->h_leader, &e->h_number,
               &e->h_key, &e->h_set, &e->h_mask, &e->h_rank, &e->o_fast, &e->o_deps, &e->o_ldeps};
  for (Buf* b : bs)
    if (b->p) (void)hipFree(b->p);
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  delete e;
  return FPX_OK;
}

int32_t fpx_epx_set_stream(fpx_epx* e, void* hip_stream) {
  if (!e) return FPX_EINVAL;
  EHIP(e, hipStreamSynchronize(e->stream));
  e->stream = hip_stream == FPX_STREAM_OWN ? e->own_stream : (hipStream_t)hip_stream;
  return FPX_OK;
}

int32_t fpx_epx_sync(fpx_epx* e) {
  if (!e) return FPX_EINVAL;
  int32_t h[2] = {0, 0};
  EHIP(e, hipMemcpyAsync(h, e->st.status, sizeof(h), hipMemcpyDeviceToHost, e->stream));
  EHIP(e, hipStreamSynchronize(e->stream));
  if (h[0] != 0) {
    EHIP(e, hipMemsetAsync(e->st.status, 0, 32, e->stream));
    EHIP(e, hipStreamSynchronize(e->stream));
  }
  return h[0];
}

int32_t fpx_epx_preaccept_dev(fpx_epx* e, int32_t m, const int32_t* d_leader, const int32_t* d_number,
                              const int32_t* d_key, const uint8_t* d_is_set, const uint8_t* d_resp_mask,
                              const int32_t* d_rank, uint8_t* d_fast, int32_t* d_deps, int32_t* d_leader_deps) {
  if (!e || m < 0) return FPX_EINVAL;
  if (m == 0) return FPX_OK;
  const int n = e->st.n;
  int rc;
  if ((rc = grow(e, &e->sk, (size_t)n * m * 4))) return rc;
  if ((rc = grow(e, &e->sk2, (size_t)n * m * 4))) return rc;
  if ((rc = grow(e, &e->tick, (size_t)e->st.num_keys * 2 * n * 4))) return rc;
  EHIP(e, hipMemsetAsync(e->tick.p, 0, (size_t)e->st.num_keys * 2 * n * 4, e->stream));
  if ((rc = grow(e, &e->sv, (size_t)n * m * 4))) return rc;
  if ((rc = grow(e, &e->sv2, (size_t)n * m * 4))) return rc;
  if ((rc = grow(e, &e->seg, (size_t)n * e->st.num_keys * 8))) return rc;
  if ((rc = grow(e, &e->conf, (size_t)m * n * n * 4))) return rc;
  EpxBatch b;
  memset(&b, 0, sizeof(b));
  b.m = m, b.leader = d_leader, b.number = d_number, b.key = d_key, b.is_set = d_is_set, b.resp_mask = d_resp_mask;
  b.rank = d_rank;
  b.sk = (uint32_t*)e->sk.p, b.sv = (int32_t*)e->sv.p, b.sk_sorted = (uint32_t*)e->sk2.p, b.sv_sorted = (int32_t*)e->sv2.p;
  b.tick = (int32_t*)e->tick.p;
  b.seg = (int32_t*)e->seg.p, b.conf = (int32_t*)e->conf.p;
  b.fast = d_fast, b.deps = d_deps, b.leader_deps = d_leader_deps;
  hipLaunchKernelGGL(k_epx_keys, dim3((m + 255) / 256), dim3(256), 0, e->stream, e->st, b);
  // stable LSD radix sort on the key bits only (the sequence already is in delivery order)
  unsigned bits = 1;
  while ((1u << bits) <= (unsigned)e->st.num_keys) ++bits;
  size_t tmp_bytes = 0;
  EHIP(e, rocprim::radix_sort_pairs(nullptr, tmp_bytes, b.sk, b.sk_sorted, b.sv, b.sv_sorted, (size_t)m, 0, bits, e->stream));
  if ((rc = grow(e, &e->tmp, tmp_bytes))) return rc;
  for (int r = 0; r < n; ++r) {
    size_t tb = e->tmp.cap;
    EHIP(e, rocprim::radix_sort_pairs(e->tmp.p, tb, b.sk + (size_t)r * m, b.sk_sorted + (size_t)r * m,
                                      b.sv + (size_t)r * m, b.sv_sorted + (size_t)r * m, (size_t)m, 0, bits, e->stream));
  }
  const int segs = n * e->st.num_keys;
  hipLaunchKernelGGL(k_epx_segments, dim3((segs + 255) / 256), dim3(256), 0, e->stream, e->st, b);
  switch (n) {
    case 3: launch_scan_decide<3>(e, b); break;
    case 5: launch_scan_decide<5>(e, b); break;
    default: launch_scan_decide<7>(e, b); break;
  }
  const long long tot = (long long)e->st.num_keys * n * n;
  hipLaunchKernelGGL(k_epx_commit, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, e->stream, e->st, b);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    e->last_hip = (int)le;
    return FPX_EHIP;
  }
  return FPX_OK;
}

int32_t fpx_epx_preaccept(fpx_epx* e, int32_t m, const int32_t* leader, const int32_t* number, const int32_t* key,
                          const uint8_t* is_set, const uint8_t* resp_mask, const int32_t* rank, uint8_t* fast,
                          int32_t* deps, int32_t* leader_deps) {
  if (!e || m < 0 || (m > 0 && (!leader || !number || !key || !is_set || !resp_mask || !rank))) return FPX_EINVAL;
  if (m == 0) return FPX_OK;
  const int n = e->st.n;
  int rc;
  auto up = [&](Buf* b, const void* src, size_t bytes) -> int {
    int r2 = grow(e, b, bytes);
    if (r2) return r2;
    EHIP(e, hipMemcpyAsync(b->p, src, bytes, hipMemcpyHostToDevice, e->stream));
    return FPX_OK;
  };
  if ((rc = up(&e->h_leader, leader, (size_t)m * 4))) return rc;
  if ((rc = up(&e->h_number, number, (size_t)m * 4))) return rc;
  if ((rc = up(&e->h_key, key, (size_t)m * 4))) return rc;
  if ((rc = up(&e->h_set, is_set, (size_t)m))) return rc;
  if ((rc = up(&e->h_mask, resp_mask, (size_t)m))) return rc;
  if ((rc = up(&e->h_rank, rank, (size_t)n * m * 4))) return rc;
  if ((rc = grow(e, &e->o_fast, (size_t)m))) return rc;
  if ((rc = grow(e, &e->o_deps, (size_t)m * n * 4))) return rc;
  if ((rc = grow(e, &e->o_ldeps, (size_t)m * n * 4))) return rc;
  rc = fpx_epx_preaccept_dev(e, m, (int32_t*)e->h_leader.p, (int32_t*)e->h_number.p, (int32_t*)e->h_key.p,
                             (uint8_t*)e->h_set.p, (uint8_t*)e->h_mask.p, (int32_t*)e->h_rank.p, (uint8_t*)e->o_fast.p,
                             (int32_t*)e->o_deps.p, (int32_t*)e->o_ldeps.p);
  if (rc) return rc;
  if (fast) EHIP(e, hipMemcpyAsync(fast, e->o_fast.p, (size_t)m, hipMemcpyDeviceToHost, e->stream));
  if (deps) EHIP(e, hipMemcpyAsync(deps, e->o_deps.p, (size_t)m * n * 4, hipMemcpyDeviceToHost, e->stream));
  if (leader_deps) EHIP(e, hipMemcpyAsync(leader_deps, e->o_ldeps.p, (size_t)m * n * 4, hipMemcpyDeviceToHost, e->stream));
  return fpx_epx_sync(e);
}

int32_t fpx_epx_read_index(fpx_epx* e, int32_t replica, int32_t key, int32_t* gets, int32_t* sets) {
  if (!e || replica < 0 || replica >= e->st.n || key < 0 || key >= e->st.num_keys) return FPX_EINVAL;
  const size_t off = ((size_t)replica * e->st.num_keys + key) * e->st.n;
  EHIP(e, hipStreamSynchronize(e->stream));
  if (gets) EHIP(e, hipMemcpy(gets, e->st.gets + off, (size_t)e->st.n * 4, hipMemcpyDeviceToHost));
  if (sets) EHIP(e, hipMemcpy(sets, e->st.sets + off, (size_t)e->st.n * 4, hipMemcpyDeviceToHost));
  return FPX_OK;
}

}  // extern "C"
