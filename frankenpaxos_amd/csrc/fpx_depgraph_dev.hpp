// fpx_depgraph_dev.hpp -- dependency-graph execution ON THE DEVICE for what the EPaxos kernels commit (SURVEY.md 8f row
// 4).  Included by fpx_epaxos.hip inside its anonymous namespace (uses the radix sort, the DPP
// helpers, Buf, grow).  The host code of fpx_depgraph.cpp stays the general path (explicit dependency sets, sequence
// numbers, small ticks); this is the path for a tick of top-one dependencies at the tick's size.
//
// What is computed (depgraph/DependencyGraph.scala:126-192, TarjanDependencyGraph.scala:225-276): the strongly connected
// components of the committed vertices, components in reverse topological order (a dependency's component never after its
// dependent's), inside a component by (sequenceNumber, key) -- sequence numbers are 0 here (epaxos/Replica.scala:575-578),
// so by (leader, id).  Where the reference leaves the order open (unrelated components) this path takes its own: the
// tests hold it to the SET of components of fpx_depgraph.cpp and to the validity of the order, not to the same sequence.
//
// How.  With top-one dependencies an instance's dependencies are n PREFIXES: vertex v = (L, x) depends on every (l, y)
// with y < d_v[l] (own column: y < x, and the explicit ids x + 1 .. end - 1 when channels reordered).  Nothing is
// materialised as edges:
//   closure  c_v[l] = the largest watermark such that v reaches every (l, y < c_v[l]).  c_v = cover_v  v  max over l' of
//            P[l'][c_v[l']], P[l'][w] = prefix max of the closures of column l' below w: a monotone fixed point; every
//            round doubles the hop distance covered, rounds alternate a prefix-max scan per column with a gather per
//            vertex until nothing moves (5 - 6 rounds on what K5 commits at 2^20 commands).  Round 5: the loop lives on the
//            device -- DG_ROUNDS rounds are enqueued at once, a round whose predecessor moved nothing returns at its first
//            instruction (pre[] then belongs to the final closures), the gather of round r also folds the tile maxima
//            round r + 1 scans from (k_dg_tilemax runs for the first round only), and everything behind the loop reads
//            the number of executables from the device: ONE host read per call (round 4: one per round + two)
//   cycles   v lies on a cycle iff the closure of one of its DIRECT dependencies covers x in column L
//   SCCs     two vertices on cycles are in one component iff their closures are equal (u in reach(v) gives c_u <= c_v, and
//            both ways gives equality; conversely equal closures of two cyclic vertices contain both)
//   order    u in reach(v), different components  =>  sum(c_u) <= sum(c_v), with equality only if u is cyclic and v is
//            not, or u lies inside its own prefix and v does not: sorting by (sum of the closure, kind) is a valid
//            order; the closure's hash as the least significant part of the key makes the members of a component
//            neighbours (a collision of two closures with one sum is detected and sent the host's way)
//   eligible v executes iff everything it reaches is committed: c_v[l] <= first[l] + count[l] for all l
// Columns are dense: column l holds the instances first[l] .. first[l] + count[l] - 1, everything below first[l] executed.
#pragma once

constexpr int DG_TILE = 2048;  // vertices per scan tile
constexpr int DG_ROUNDS = 8;   // closure rounds enqueued per chunk (a round doubles the hops covered)
constexpr int DG_SUB = DG_TILE / 256;  // workgroups of k_dg_relax per tile: each leaves its own row of maxima (no atomics)

template <int N> struct DgRow { static constexpr int NP = N <= 4 ? 4 : 8; };

struct DgArgs {
  int m, n, stride;                 // messages, replicas, ints per packed line
  int32_t first[8], count[8], base[8], tiles[8], tile_base[8];  // per column: first id, instances, first vertex, scan tiles
  int ntiles;
  const int32_t* leader;            // [m]
  const int32_t* number;            // [m]
  const int32_t* packed;            // [m][stride]: deps | leader_deps | own_values_end[2] | fast
  const uint8_t* mask;              // [m] or null: 0 = not committed (blocks what depends on it)
  int32_t* msg_of;                  // [m] vertex -> message (-1: no such instance was handed in)
  int32_t* direct;                  // [m][NP] direct dependency covers (own column: max(watermark, values end))
  int32_t* clo;                     // [m][NP] closure
  int32_t* pre;                     // [m][NP] prefix max of clo within the column
  int32_t* tmax;                    // [ntiles][DG_SUB][NP]: what the next round's scan carries in from the tiles before it (one row
                                    // per workgroup of k_dg_relax; k_dg_tilemax fills row 0 of a tile for the first round)
  int32_t* tstarts;                 // [out tiles] component starts per tile of the sorted order
  int32_t* belig;                   // [ceil(m / 256)] executable vertices per workgroup of k_dg_keys (summed by k_dg_rekey)
  uint2* pairs;                     // [m] (sort key, vertex)
  uint2* pairs2;
  uint32_t* key32;                  // [m] the main sort key of a vertex (pairs carry the closure's hash first)
  int32_t* ctl;                     // [1] malformed, [2] needs the host path, [3] executables, [4] components,
                                    // [8 + k] something moved in round k of the chunk of rounds being enqueued
  volatile int32_t* host;           // page-locked mirror of ctl (8 ints) + [7] = seq
  int32_t seq;
  int32_t count_moved;              // debug: count the vertices a round moves into ctl[24 + k]
  int32_t hash_bits;                // bits of the closure hash the cyclic vertices are grouped by (DG_HASH_BITS; fewer under
                                    // FPX_DG_HASH_BITS, which tests use to force the collisions ctl[2] reports)
  int32_t* order;                   // [m] message indices in execution order
  int32_t* comp;                    // [m] component number of position p (0, 1, ..)
};

// vertex of (L, x), or -1
__device__ __forceinline__ int dg_vertex(const DgArgs& a, int L, int x) {
  const int j = x - a.first[L];
  return (j >= 0 && j < a.count[L]) ? a.base[L] + j : -1;
}

template <int N>
__global__ void __launch_bounds__(256) k_dg_scatter(const DgArgs a) {
  constexpr int NP = DgRow<N>::NP;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.m) return;
  const int L = a.leader[i], x = a.number[i];
  const int v = (L >= 0 && L < N) ? dg_vertex(a, L, x) : -1;
  if (v < 0 || atomicExch(&a.msg_of[v], i) != -1) {  // outside its column, or the instance twice
    a.ctl[1] = 1;
    return;
  }
  const int32_t* line = a.packed + (size_t)i * a.stride;
  const bool committed = !a.mask || a.mask[i];
  int d[NP];
#pragma unroll
  for (int l = 0; l < NP; ++l) d[l] = 0;
#pragma unroll
  for (int l = 0; l < N; ++l) {
    // an instance that is not committed can be waited for only: its cover is beyond every column, so is the closure of
    // whatever reaches it
    d[l] = committed ? line[l] : 0x3fffffff;
    if (d[l] < 0) a.ctl[1] = 1;
  }
  const int end = line[2 * N];  // explicit ids x + 1 .. end - 1 of the own column (dependencies.subtractOne, Replica.scala:582)
  if (committed && end > 0) {
    if (end <= x + 1 || d[L] != x) a.ctl[1] = 1;
    d[L] = end;
  }
  int4* dp = reinterpret_cast<int4*>(a.direct + (size_t)v * NP);
  int4* cp = reinterpret_cast<int4*>(a.clo + (size_t)v * NP);
#pragma unroll
  for (int q = 0; q < NP / 4; ++q) dp[q] = cp[q] = make_int4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
}

// column-wise max of one tile of closures
template <int N>
__global__ void __launch_bounds__(256) k_dg_tilemax(const DgArgs a) {
  constexpr int NP = DgRow<N>::NP;
  __shared__ int sh[4][NP];
  int col = 0;
  while (col + 1 < N && (int)blockIdx.x >= a.tile_base[col + 1]) ++col;
  const int t = blockIdx.x - a.tile_base[col];
  const int lo = a.base[col] + t * DG_TILE, hi = min(a.base[col] + a.count[col], lo + DG_TILE);
  int mx[NP];
#pragma unroll
  for (int l = 0; l < NP; ++l) mx[l] = 0;
  for (int v = lo + threadIdx.x; v < hi; v += 256) {
    const int4* cp = reinterpret_cast<const int4*>(a.clo + (size_t)v * NP);
#pragma unroll
    for (int q = 0; q < NP / 4; ++q) {
      const int4 c = cp[q];
      mx[4 * q] = imax(mx[4 * q], c.x), mx[4 * q + 1] = imax(mx[4 * q + 1], c.y), mx[4 * q + 2] = imax(mx[4 * q + 2], c.z),
      mx[4 * q + 3] = imax(mx[4 * q + 3], c.w);
    }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int l = 0; l < NP; ++l) {
    const int tot = __builtin_amdgcn_readlane(wave_incl_max(mx[l]), 63);
    if (lane == 0) sh[w][l] = tot;
  }
  __syncthreads();
  if (threadIdx.x < NP) a.tmax[(size_t)blockIdx.x * DG_SUB * NP + threadIdx.x] = imax(imax(sh[0][threadIdx.x], sh[1][threadIdx.x]), imax(sh[2][threadIdx.x], sh[3][threadIdx.x]));
  else if (threadIdx.x < DG_SUB * NP) a.tmax[(size_t)blockIdx.x * DG_SUB * NP + threadIdx.x] = 0;
}

// pre[v] = max of the closures of the column's vertices up to and including v
// r = the round (absolute: the tile maxima's half), k = its number within the chunk enqueued at once
template <int N>
__global__ void __launch_bounds__(256) k_dg_prefix(const DgArgs a, int r, int k) {
  constexpr int NP = DgRow<N>::NP;
  __shared__ int carry[NP];
  __shared__ int wtot[4][NP];
  if (k > 1 && a.ctl[8 + k - 1] == 0) return;  // the round before moved nothing: pre[] is final
  (void)r;
  int col = 0;
  while (col + 1 < N && (int)blockIdx.x >= a.tile_base[col + 1]) ++col;
  const int t = blockIdx.x - a.tile_base[col];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // the tiles of the column before this one
  if (w == 0) {
    int mx[NP];
#pragma unroll
    for (int l = 0; l < NP; ++l) mx[l] = 0;
    for (int j = lane; j < t * DG_SUB; j += 64) {
      const int4* tm = reinterpret_cast<const int4*>(a.tmax + ((size_t)a.tile_base[col] * DG_SUB + j) * NP);
#pragma unroll
      for (int q = 0; q < NP / 4; ++q) {
        const int4 x = tm[q];
        mx[4 * q] = imax(mx[4 * q], x.x), mx[4 * q + 1] = imax(mx[4 * q + 1], x.y), mx[4 * q + 2] = imax(mx[4 * q + 2], x.z), mx[4 * q + 3] = imax(mx[4 * q + 3], x.w);
      }
    }
#pragma unroll
    for (int l = 0; l < NP; ++l) {
      const int tot = __builtin_amdgcn_readlane(wave_incl_max(mx[l]), 63);
      if (lane == 0) carry[l] = tot;
    }
  }
  __syncthreads();
  const int lo = a.base[col] + t * DG_TILE, hi = min(a.base[col] + a.count[col], lo + DG_TILE);
  for (int v0 = lo; v0 < hi; v0 += 256) {  // 256 vertices per step, in order
    const int v = v0 + threadIdx.x;
    int c[NP];
#pragma unroll
    for (int l = 0; l < NP; ++l) c[l] = 0;
    if (v < hi) {
      const int4* cp = reinterpret_cast<const int4*>(a.clo + (size_t)v * NP);
#pragma unroll
      for (int q = 0; q < NP / 4; ++q) {
        const int4 x = cp[q];
        c[4 * q] = x.x, c[4 * q + 1] = x.y, c[4 * q + 2] = x.z, c[4 * q + 3] = x.w;
      }
    }
#pragma unroll
    for (int l = 0; l < NP; ++l) {
      c[l] = wave_incl_max(c[l]);
      if (lane == 63) wtot[w][l] = c[l];
    }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < NP; ++l) {
      int before = carry[l];
      for (int w2 = 0; w2 < w; ++w2) before = imax(before, wtot[w2][l]);
      c[l] = imax(c[l], before);
    }
    if (v < hi) {
      int4* pp = reinterpret_cast<int4*>(a.pre + (size_t)v * NP);
#pragma unroll
      for (int q = 0; q < NP / 4; ++q) pp[q] = make_int4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);
    }
    __syncthreads();
    if (threadIdx.x < NP) {
      int nx = carry[threadIdx.x];
      for (int w2 = 0; w2 < 4; ++w2) nx = imax(nx, wtot[w2][threadIdx.x]);
      carry[threadIdx.x] = nx;
    }
    __syncthreads();
  }
}

// c_v = c_v  v  max over l of pre[column l][c_v[l] - 1]; and what the next round's scan carries from tile to tile: the
// grid is laid over the scan tiles (DG_SUB workgroups of 256 vertices per tile), so a workgroup's maxima are ONE row of
// tmax written with plain stores.  (A first version folded them with atomicMax from vertex-order workgroups: 160 atomics
// on each tile's 32-byte line cost 55 us per launch, profiles/r05_depgraph_dev.md.)
template <int N>
__global__ void __launch_bounds__(256) k_dg_relax(const DgArgs a, int r, int k) {
  constexpr int NP = DgRow<N>::NP;
  (void)r;
  __shared__ int sh[4][NP];
  if (k > 1 && a.ctl[8 + k - 1] == 0) return;
  const int tile = blockIdx.x / DG_SUB, sub = blockIdx.x % DG_SUB;
  int col = 0;
  while (col + 1 < N && tile >= a.tile_base[col + 1]) ++col;
  const int j = (tile - a.tile_base[col]) * DG_TILE + sub * 256 + threadIdx.x;
  const bool live = j < a.count[col];
  const int v = a.base[col] + j;
  int c[NP], o[NP];
#pragma unroll
  for (int l = 0; l < NP; ++l) c[l] = o[l] = 0;
  bool moved = false;
  if (live) {
    if (a.msg_of[v] < 0) a.ctl[1] = 1;  // an instance of the column that was not handed in: the columns are not dense
    {
      const int4* cp = reinterpret_cast<const int4*>(a.clo + (size_t)v * NP);
#pragma unroll
      for (int q = 0; q < NP / 4; ++q) {
        const int4 x = cp[q];
        c[4 * q] = o[4 * q] = x.x, c[4 * q + 1] = o[4 * q + 1] = x.y, c[4 * q + 2] = o[4 * q + 2] = x.z, c[4 * q + 3] = o[4 * q + 3] = x.w;
      }
    }
#pragma unroll
    for (int l = 0; l < N; ++l) {
      const int jj = min(o[l], a.first[l] + a.count[l]) - a.first[l] - 1;  // the last vertex of column l below the watermark
      if (jj >= 0) {
        const int4* pp = reinterpret_cast<const int4*>(a.pre + (size_t)(a.base[l] + jj) * NP);
#pragma unroll
        for (int q = 0; q < NP / 4; ++q) {
          const int4 x = pp[q];
          c[4 * q] = imax(c[4 * q], x.x), c[4 * q + 1] = imax(c[4 * q + 1], x.y), c[4 * q + 2] = imax(c[4 * q + 2], x.z), c[4 * q + 3] = imax(c[4 * q + 3], x.w);
        }
      }
    }
#pragma unroll
    for (int l = 0; l < N; ++l) moved = moved || c[l] != o[l];
    if (moved) {
      int4* cp = reinterpret_cast<int4*>(a.clo + (size_t)v * NP);
#pragma unroll
      for (int q = 0; q < NP / 4; ++q) cp[q] = make_int4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);
    }
  }
  if (__any(moved) && (threadIdx.x & 63) == 0 && a.ctl[8 + k] == 0) a.ctl[8 + k] = 1;
  if (a.count_moved) {  // FPX_DG_DEBUG: how many vertices each round moves
    const unsigned long long bal = __ballot(moved);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&a.ctl[24 + k], (int)__popcll(bal));
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int l = 0; l < NP; ++l) {
    const int tot = __builtin_amdgcn_readlane(wave_incl_max(c[l]), 63);
    if (lane == 0) sh[w][l] = tot;
  }
  __syncthreads();
  if (threadIdx.x < NP)
    a.tmax[(size_t)blockIdx.x * NP + threadIdx.x] = imax(imax(sh[0][threadIdx.x], sh[1][threadIdx.x]), imax(sh[2][threadIdx.x], sh[3][threadIdx.x]));
}

// publishes the control words of the round to the host (one workgroup, after the round's kernels)
__global__ void k_dg_publish(const DgArgs a, int round) {
  (void)round;
  if (threadIdx.x < 7) a.host[threadIdx.x] = threadIdx.x == 5 ? a.ctl[8 + DG_ROUNDS] : a.ctl[threadIdx.x];  // [5]: the chunk's last round still moved
  __threadfence_system();
  if (threadIdx.x == 0) a.host[7] = a.seq;  // (call, round): what the host waits for
}

// the sort key of every vertex: 3 x (sum of the closure beyond the columns' executed prefixes) + kind (the sum is at most m,
// so the key fits ceil(log2(3 m + 4)) bits: 22 at 2^20 commands = TWO 11-bit sort passes; `sum << 2 | kind` took three), kind 0 = on a
// cycle, 1 = inside its own prefix (explicit ids reach over it) but on no cycle, 2 = neither; ~0 = cannot execute yet
constexpr int DG_HASH_BITS = 22;
template <int N>
__device__ __forceinline__ uint32_t dg_hash(const int* c, int bits) {
  uint32_t h = 0x9E3779B9u;
#pragma unroll
  for (int l = 0; l < N; ++l) h = kp_mix32(h ^ (uint32_t)c[l]) + 0x7F4A7C15u;
  return h >> (32 - bits);
}
__global__ void __launch_bounds__(256) k_dg_rekey(const DgArgs a) {  // after the sort on the hash: the main key, by vertex
  __shared__ uint32_t sh[8];
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < a.m) a.pairs[p].x = a.key32[a.pairs[p].y];
  if (blockIdx.x == 0) {  // the number of executables (ctl[3]) from the workgroups' counts of k_dg_keys
    const int vblocks = (a.m + 255) >> 8;
    uint32_t s = 0;
    for (int b = (int)threadIdx.x; b < vblocks; b += 256) s += (uint32_t)a.belig[b];
    const uint32_t ex = block_excl_sum(s, sh);
    if (threadIdx.x == 255) a.ctl[3] = (int32_t)(ex + s);
  }
}

template <int N>
__global__ void __launch_bounds__(256) k_dg_keys(const DgArgs a) {
  constexpr int NP = DgRow<N>::NP;
  __shared__ int block_eligible;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (threadIdx.x == 0) block_eligible = 0;
  __syncthreads();
  bool eligible = false;
  if (v < a.m) {
  int c[NP], d[NP];
  {
    const int4* cp = reinterpret_cast<const int4*>(a.clo + (size_t)v * NP);
    const int4* dp = reinterpret_cast<const int4*>(a.direct + (size_t)v * NP);
#pragma unroll
    for (int q = 0; q < NP / 4; ++q) {
      const int4 x = cp[q], y = dp[q];
      c[4 * q] = x.x, c[4 * q + 1] = x.y, c[4 * q + 2] = x.z, c[4 * q + 3] = x.w;
      d[4 * q] = y.x, d[4 * q + 1] = y.y, d[4 * q + 2] = y.z, d[4 * q + 3] = y.w;
    }
  }
  int L = 0;
  while (L + 1 < N && v >= a.base[L + 1]) ++L;
  const int x = a.first[L] + (v - a.base[L]);
  eligible = true;
  uint32_t sum = 0;
#pragma unroll
  for (int l = 0; l < N; ++l) {
    eligible = eligible && c[l] <= a.first[l] + a.count[l];
    sum += (uint32_t)max(0, c[l] - a.first[l]);
  }
  // on a cycle iff the closure of a direct dependency covers x in column L.  Only a vertex that can execute is asked: one
  // that is not committed has 0x3fffffff in every column of d[] and would walk the rest of its column, load by load
  // (O(m^2) per tick with a few percent of the tick uncommitted)
  int back = 0;
  if (eligible) {
#pragma unroll
    for (int l = 0; l < N; ++l) {
      const int bound = l == L ? min(d[l], x) : d[l];  // own column: the prefix below x here, the explicit ids below
      const int j = min(bound, a.first[l] + a.count[l]) - a.first[l] - 1;
      if (j >= 0) back = imax(back, a.pre[(size_t)(a.base[l] + j) * NP + L]);
    }
    for (int y = x + 1; y < min(d[L], a.first[L] + a.count[L]); ++y) back = imax(back, a.clo[(size_t)(a.base[L] + y - a.first[L]) * NP + L]);
  }
  const uint32_t kind = back > x ? 0u : (c[L] > x ? 1u : 2u);
  const uint32_t key = eligible ? (sum * 3u + kind) : 0xffffffffu;  // (all ones: behind every executable in the bits that are sorted)
  if (eligible && sum >= (1u << 29)) a.ctl[2] = 1;
  a.key32[v] = key;
  // vertices on cycles with one closure sum can belong to several components: the closure's hash orders them first, so
  // that the members of a component (equal closures) end up neighbours
  a.pairs[v] = make_uint2((eligible && kind == 0u) ? dg_hash<N>(c, a.hash_bits) : 0u, (uint32_t)v);
  }
  // the number of executables: wave -> workgroup (LDS) -> ONE atomic per workgroup on the counter (an atomic per
  // wavefront was 16 384 of them queueing on one address: most of this kernel's 217 us, profiles/r04_depgraph_dev.md)
  const unsigned long long bal = __ballot(eligible);
  if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&block_eligible, (int)__popcll(bal));
  __syncthreads();
  // (round 6: not even one atomic per workgroup on the counter -- 4096 of them on one word were ~35 us of the packed path's
  // 54 us kernel, profiles/r06_depgraph_dev.md; each workgroup leaves its count, k_dg_rekey's first workgroup adds them up)
  if (threadIdx.x == 0) a.belig[blockIdx.x] = block_eligible;
}

template <int N>
__device__ __forceinline__ int dg_cmp_clo(const DgArgs& a, uint32_t u, uint32_t v) {  // closure vectors, lexicographically
  constexpr int NP = DgRow<N>::NP;
  for (int l = 0; l < N; ++l) {
    const int x = a.clo[(size_t)u * NP + l], y = a.clo[(size_t)v * NP + l];
    if (x != y) return x < y ? -1 : 1;
  }
  return 0;
}

// component starts -> component numbers (inclusive scan of the start flags, one workgroup per DG_TILE positions, the
// tiles' totals through tmax as three kernels would: here the flags of a tile are counted by every later tile again --
// 512 tiles x 8 KB at most), and the message indices in execution order
// does a component start at position p of the sorted order?  Vertices that lie on no cycle are components of their own;
// cyclic neighbours with one key belong together iff their closures are equal.  Two DIFFERENT closures with one key and
// one hash would leave their members interleaved: that is seen here and sends the tick the host's way.
template <int N>
__device__ __forceinline__ uint32_t dg_starts(const DgArgs& a, int p, int executables) {  // executables = ctl[3], read by the caller
  constexpr int NP = DgRow<N>::NP;
  if (p >= executables) return 0u;
  const uint2 e = a.pairs[p];
  if (e.x % 3u != 0u || p == 0) return 1u;
  const uint2 f = a.pairs[p - 1];
  if (f.x != e.x) return 1u;
  if (dg_cmp_clo<N>(a, f.y, e.y) == 0) return 0u;
  if (dg_hash<N>(a.clo + (size_t)f.y * NP, a.hash_bits) == dg_hash<N>(a.clo + (size_t)e.y * NP, a.hash_bits)) a.ctl[2] = 1;
  return 1u;
}

template <int N>
__global__ void __launch_bounds__(256) k_dg_emit(const DgArgs a) {
  __shared__ uint32_t sh[8];
  __shared__ uint32_t before_tile;
  const int executables = a.ctl[3];
  auto starts = [&](int p) -> uint32_t { return dg_starts<N>(a, p, executables); };
  const int t0 = blockIdx.x * DG_TILE;
  // flags of my tile
  uint32_t mine[DG_TILE / 256], total = 0;
#pragma unroll
  for (int j = 0; j < DG_TILE / 256; ++j) mine[j] = starts(t0 + j * 256 + threadIdx.x), total += mine[j];
  // the components that start before my tile were counted into tstarts[tile] by k_dg_count_starts
  if (threadIdx.x == 0) {
    uint32_t s = 0;
    for (int t = 0; t < (int)blockIdx.x; ++t) s += (uint32_t)a.tstarts[t];
    before_tile = s;
  }
  __syncthreads();
  uint32_t run = before_tile;
#pragma unroll
  for (int j = 0; j < DG_TILE / 256; ++j) {
    const uint32_t ex = block_excl_sum(mine[j], sh);
    const int p = t0 + j * 256 + threadIdx.x;
    if (p < executables) {
      a.comp[p] = (int32_t)(run + ex + mine[j]) - 1;
      a.order[p] = a.msg_of[a.pairs[p].y];
    }
    __syncthreads();
    if (threadIdx.x == 255) sh[7] = ex + mine[j];
    __syncthreads();
    run += sh[7];
    __syncthreads();
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) a.ctl[4] = (int32_t)run;
}

template <int N>
__global__ void __launch_bounds__(256) k_dg_count_starts(const DgArgs a) {
  __shared__ uint32_t sh[8];
  const int executables = a.ctl[3];
  const int t0 = blockIdx.x * DG_TILE;
  uint32_t total = 0;
  for (int j = 0; j < DG_TILE / 256; ++j) {
    const int p = t0 + j * 256 + threadIdx.x;
    total += dg_starts<N>(a, p, executables);
  }
  const uint32_t ex = block_excl_sum(total, sh);
  if (threadIdx.x == 255) a.tstarts[blockIdx.x] = (int32_t)(ex + total);
}
