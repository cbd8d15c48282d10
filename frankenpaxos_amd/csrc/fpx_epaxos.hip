// fpx_epaxos.hip -- EPaxos on gfx950: K5 the pre-accept tick (fast path test + slow-path
// union), K6 Prepare / Accept on the command log, K7 handlePreAccept in full (ballots, Nacks, re-sent replies).
//
// K5:
// Per replica the conflict scan is a segmented (by key) exclusive prefix-max over the tick's commands
// in that replica's delivery order, on n-wide watermark vectors (util/TopOne.scala): data-parallel as
//   1. k_epx_keys      scatter every command to its position in the replica's delivery order (rank is a
//                      permutation)
//   2. k_rs_hist / k_rs_scan / k_rs_scatter   stable LSD radix sort (digits of up to 11 bits: 1024 keys are one pass)
//                      of all replicas' sequences at once, on the key bits only (stable => (key, delivery
//                      order)); a workgroup sorts its tile of 4096 elements into LDS (equal digits ranked with
//                      one ballot per digit bit and 64 elements) and writes it out as runs per digit
//   3. k_epx_segments_from_totals  [lo, hi) of every (replica, key) segment from the sort's digit totals (one pass:
//                      digit == key); k_epx_segments (binary search) when the key took several passes
//   4. k_epx_key<N>    one persistent workgroup per CU working through the keys whose commands fit its LDS tables: the scans of all
//                      replicas' segments of the key (steps 4a / 5a below in one kernel, the conflict rows never
//                      leave the chip); the other keys go through
//   4a. k_epx_scan<N>  one wavefront per (replica, key): 64 commands per step, wave-level max-scan of the
//                      2N watermark columns on the DPP network (row_shr / row_bcast), carry in registers;
//                      leader and get/set travel in the sort key's spare bits, the answer row is one aligned
//                      16- / 32-byte store
//   5a. k_epx_decide<N> one thread per command: PreAcceptOk = local conflicts U leader's deps; fast path iff
//                      the n-2 answers agree (Util.popularItems), else the union (preAcceptingSlowPath)
//   6. k_epx_commit    every replica's conflict index learns the tick's instances (commit ->
//                      updateConflictIndex): elementwise max with what the scans saw of this tick (every
//                      command is scanned by its leader's replica at least, so the max over replicas is the
//                      whole tick -- no atomics anywhere)
// Integer max / compare only: HBM- and latency-bound, no MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/fpx.h"

namespace {

// cmdLog entry kinds (epaxos/Replica.scala:303-330)
enum { CL_NONE = 0, CL_NO_COMMAND = 1, CL_PRE_ACCEPTED = 2, CL_ACCEPTED = 3, CL_COMMITTED = 4 };

struct EpxState {
  int n, num_keys;
  int32_t* gets;  // [n][num_keys][n]
  int32_t* sets;  // [n][num_keys][n]
  int32_t* status;
  // the command log of every replica (Replica.scala:440 cmdLog) for the instances (leader, number < num_instances):
  // entry kind, ballot, voteBallot (ballots encoded ordering * 8 + replicaIndex, nullBallot = -1), triple id
  int num_instances;
  uint8_t* cl_status;   // [n][n * num_instances]
  int32_t* cl_ballot;
  int32_t* cl_vote;
  int32_t* cl_triple;
  int32_t* cl_deps;     // [n * n * num_instances][n] the triple's dependencies as watermarks (column 0 = -1: the entry
                        // was written by an Accept, which names its triple by id only)
  int32_t* cl_dend;     // [n * n * num_instances] end of the explicit values number + 1 .. end - 1 of the own-leader column
  int32_t* largest;     // [n]  Replica.largestBallot, encoded
  uint32_t* cl_stamp;   // [n * num_instances]  run id: instances of one batch must be distinct
};

#ifndef FPX_RS_ITEMS
#define FPX_RS_ITEMS 16  // 64-element steps per wavefront (tile = 4 x that); 8 / 16 / 32 measured 0.317 / 0.302 / 0.322 ms per tick
#endif
#define FPX_RS_ITEMS_V FPX_RS_ITEMS

struct EpxBatch {
  int m;
  const int32_t* leader;
  const int32_t* number;
  const int32_t* key;
  const uint8_t* is_set;
  const uint8_t* resp_mask;
  const uint8_t* seen_mask;  // replicas that process the PreAccept (null = resp_mask)
  const int32_t* rank;   // [n][m]
  const int32_t* triple; // [m] or null: the CommandTriple's id, recorded in the command log
  uint2* kv;             // [n][m] in the replica's delivery order: x = key | is_set << 27 | leader << 28
                         // (only the key bits are sorted on), y = message index
  uint2* kv_sorted;      // [n][m]
  int32_t* tick;         // [n][num_keys][2][n] the puts (gets, sets) replica r's scan saw for the key this tick
  int32_t* seg;          // [n][num_keys][2]
  int32_t* conf;         // [m][n][NP] local conflicts of replica r for message i; rows padded to NP = 4 (n = 3)
                         // or 8 ints so a row is one aligned 16- / 32-byte store
  uint8_t* fast;
  int32_t* deps;
  int32_t* leader_deps;
  int32_t* own_values_end;  // [m][2]: explicit values of the own-leader column of deps / leader_deps (0 = none)
  int2* meta;               // [m] or null: (number, resp_mask | seen_mask << 8) of message i, ONE gather for k_epx_key
  uint8_t* fused;           // [num_keys] or null: 1 = k_epx_key did the key's scan and decisions on chip
  int32_t* unfused;         // number of keys left to k_epx_scan / k_epx_decide (null: all of them)
};

constexpr uint32_t EPX_KEY_MASK = (1u << 27) - 1u;
constexpr int EPX_SET_SHIFT = 27, EPX_LEADER_SHIFT = 28;
template <int N> struct ConfRow { static constexpr int NP = N <= 4 ? 4 : 8; };

// wave64 inclusive max-scan / shifts on the DPP network (values are >= 0, so 0 is the identity): 4 row_shr
// steps inside each row of 16, then row_bcast:15 / row_bcast:31 carry the row totals across rows
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp0(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
// The own-leader column of an instance (leader L, number x) is {0 .. cover-1} \ {x} (dependencies.subtractOne(instance),
// Replica.scala:582): an IntPrefixSet with (watermark cover, no values) if cover <= x, (watermark x, no values) if
// cover == x + 1, else (watermark x, values x+1 .. cover-1), reported as values_end = cover.  Unions are max of covers.
__device__ __forceinline__ void own_column(int cover, int x, int* watermark, int* values_end) {
  *watermark = cover <= x ? cover : x;
  *values_end = cover > x + 1 ? cover : 0;
}
__device__ __forceinline__ int wave_incl_max(int v) {
  v = imax(v, dpp0<0x111, 0xF>(v));  // row_shr:1
  v = imax(v, dpp0<0x112, 0xF>(v));  // row_shr:2
  v = imax(v, dpp0<0x114, 0xF>(v));  // row_shr:4
  v = imax(v, dpp0<0x118, 0xF>(v));  // row_shr:8
  // row_bcast:15 -> rows 1, 3, then row_bcast:31 -> rows 2, 3.  Written out: from update_dpp + max the compiler makes a
  // v_mov, a v_mov_dpp and the v_max per step (it does not fold a move under a partial row mask), 6 instructions where 2
  // do -- and this scan is a third of the K5 key kernel's vector instructions.  The s_nops are the two wait states
  // between a VALU write and a DPP read of the same register, which the hazard recogniser does not see through asm.
  asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
      : "+v"(v));
  return v;
}
__device__ __forceinline__ int wave_shr1(int v) { return dpp0<0x138, 0xF>(v); }  // lane 0 gets 0

__device__ __forceinline__ void epx_report(int32_t* status, int code, int index) {
  if (atomicCAS(&status[0], 0, code) == 0) status[1] = index;
}

__global__ void __launch_bounds__(256) k_epx_keys(const EpxState st, const EpxBatch b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.m) return;
  if (i == 0 && b.unfused) *b.unfused = 0;
  const int n = st.n;
  const int L = b.leader[i], k = b.key[i];
  const unsigned mask = b.resp_mask[i];
  bool ok = L >= 0 && L < n && b.number[i] >= 0 && k >= 0 && k < st.num_keys;
  ok = ok && !((mask >> (ok ? L : 0)) & 1u) && (mask >> n) == 0 && (int)__popc(mask) == n - 2;
  const unsigned seen = b.seen_mask ? b.seen_mask[i] : mask;
  ok = ok && (mask & ~seen) == 0 && !((seen >> (ok ? L : 0)) & 1u) && (seen >> n) == 0;
  if (b.meta) b.meta[i] = make_int2(b.number[i], (int)(mask | (seen << 8)));
  for (int r = 0; r < n; ++r) {
    const int p = b.rank[(size_t)r * b.m + i];
    ok = ok && p >= 0 && p < b.m;
    const bool part = ok && (r == L || ((seen >> r) & 1u));
    // rank is a permutation: scattering to position p lays the tick out in replica r's delivery order;
    // a stable sort by key alone then yields (key, delivery order).  Non-participants sort last.
    if (ok) {
      const uint32_t flags = ((uint32_t)(b.is_set[i] ? 1 : 0) << EPX_SET_SHIFT) | ((uint32_t)L << EPX_LEADER_SHIFT);
      b.kv[(size_t)r * b.m + p] = make_uint2((part ? (uint32_t)k : (uint32_t)st.num_keys) | flags, (uint32_t)i);
    }
  }
  if (ok && st.num_instances > 0) {
    // this tick-at-once form covers handlePreAccept's `cmdLog.get(instance) == None` branch only: an instance a
    // participating replica already knows is rejected (nothing of the tick is applied)
    ok = b.number[i] < st.num_instances;
    const unsigned part = (b.seen_mask ? b.seen_mask[i] : mask) | (1u << L);
    for (int r = 0; ok && r < n; ++r)
      if (((part >> r) & 1u) && st.cl_status[((size_t)r * n + L) * st.num_instances + b.number[i]] != CL_NONE) ok = false;
  }
  if (!ok) {
    epx_report(st.status, FPX_EINVAL, i);
    return;
  }
}

// ---- stable LSD radix sort, digits of up to 11 bits, all replicas in one launch (blockIdx.y = replica) --------
// 1024 keys (+ the non-participants' bucket) are ONE pass.  A workgroup owns a tile of RS_TILE consecutive elements,
// wavefront w the w-th quarter of it, walked 64 at a time in lane order: ranking equal digits by (tile, wavefront,
// step, lane) is the input order, so the sort is stable.  The tile is first sorted into LDS and leaves as runs of
// consecutive elements per digit (a scatter straight from the ranking loop wrote 8 bytes per lane and instruction
// to 64 different places: two such passes were 66 us for 5 M pairs, profiles/r02_epaxos_kernel_stats.csv).
constexpr int RS_ITEMS = FPX_RS_ITEMS;      // 64-element steps per wavefront
constexpr int RS_TILE = 256 * RS_ITEMS;     // elements per workgroup tile
constexpr int RS_MAXW = 11;                 // widest digit
constexpr int RS_MAXB = 1 << RS_MAXW;
#ifndef FPX_RS_SW
#define FPX_RS_SW 8
#endif
constexpr int RS_SW = FPX_RS_SW;               // wavefronts of a scatter workgroup: 4 / 8 / 16 measured 0.259 / 0.240 / 0.255 ms per tick
constexpr size_t RS_SCATTER_LDS = (size_t)RS_TILE * 8 + (size_t)RS_SW * (RS_MAXB / 2) * 4 + (size_t)RS_MAXB * 4 + 64;

struct RsArgs {
  int m, tiles, shift;
  int width;       // digit bits of this pass (<= 11): the key bits are split evenly over the passes
  const uint2* src;  // [n][m] (key word, payload)
  uint2* dst;
  uint32_t* hist;  // [n][tiles][B] per-tile digit counts -> exclusive offsets of the tile within the digit
  uint32_t* tot;   // [n][B] digit totals
  // first pass only (else null): rank[n][m] and the status words, to check that every replica's rank really is
  // a permutation -- position p must hold a message i < m of THIS tick with rank[r][i] == p.  A position no
  // message was scattered to still holds an older tick's pair, which fails one of the two tests (if its i had
  // rank p now, i would have written p).
  const int32_t* rank;
  int32_t* status;
};

__global__ void __launch_bounds__(256) k_rs_hist(const RsArgs a) {
  __shared__ uint32_t h[RS_MAXB];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = blockIdx.y, tile = blockIdx.x;
  const int B = 1 << a.width;
  for (int j = threadIdx.x; j < B; j += 256) h[j] = 0;
  __syncthreads();
  const uint2* k = a.src + (size_t)r * a.m;
  const int first = tile * RS_TILE + w * (RS_TILE / 4);
  uint32_t x[RS_ITEMS];
  if (a.rank) {
    uint32_t y[RS_ITEMS];
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
      const int idx = first + it * 64 + lane;
      const uint2 e = idx < a.m ? k[idx] : make_uint2(0xffffffffu, 0u);
      x[it] = e.x, y[it] = e.y;
    }
    int bad = -1;
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
      const int idx = first + it * 64 + lane;
      if (idx < a.m && (y[it] >= (uint32_t)a.m || a.rank[(size_t)r * a.m + y[it]] != idx)) bad = idx;
    }
    if (bad >= 0) epx_report(a.status, FPX_EINVAL, -1);
  } else {
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {  // all the loads in flight before the first LDS atomic
      const int idx = first + it * 64 + lane;
      x[it] = idx < a.m ? k[idx].x : 0xffffffffu;
    }
  }
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it)
    if (first + it * 64 + lane < a.m) atomicAdd(&h[(x[it] >> a.shift) & (B - 1)], 1u);
  __syncthreads();
  uint32_t* out = a.hist + ((size_t)r * a.tiles + tile) * B;
  for (int j = threadIdx.x; j < B; j += 256) out[j] = h[j];
}

// exclusive scan of every digit's per-tile counts, in place: 64 digits x 4 parts of the tiles per workgroup (the 64
// lanes of a wavefront read neighbouring digits of one tile: coalesced); a part sums its tiles, the parts' sums are
// combined through LDS, then each part rewrites its tiles (second read: L2)
__global__ void __launch_bounds__(256) k_rs_scan(const RsArgs a) {
  __shared__ uint32_t part_sum[4][64];
  const int B = 1 << a.width, lane = threadIdx.x & 63, part = threadIdx.x >> 6, r = blockIdx.y;
  const int d = blockIdx.x * 64 + lane;
  const int per = (a.tiles + 3) / 4, t0 = part * per, t1 = min(a.tiles, t0 + per);
  uint32_t* col = a.hist + (size_t)r * a.tiles * B + d;
  uint32_t sum = 0;
  if (d < B) {
    int t = t0;
    for (; t + 8 <= t1; t += 8) {
      uint32_t v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = col[(size_t)(t + j) * B];
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[j];
    }
    for (; t < t1; ++t) sum += col[(size_t)t * B];
  }
  part_sum[part][lane] = sum;
  __syncthreads();
  if (d >= B) return;
  uint32_t run = 0;
  for (int p2 = 0; p2 < part; ++p2) run += part_sum[p2][lane];
  if (part == 3) a.tot[r * B + d] = run + sum;
  int t = t0;
  for (; t + 8 <= t1; t += 8) {
    uint32_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = col[(size_t)(t + j) * B];
#pragma unroll
    for (int j = 0; j < 8; ++j) col[(size_t)(t + j) * B] = run, run += v[j];
  }
  for (; t < t1; ++t) {
    const uint32_t v = col[(size_t)t * B];
    col[(size_t)t * B] = run, run += v;
  }
}

// exclusive prefix of one value per thread over the threads of the workgroup (up to 512; sh: 8 words of LDS)
__device__ __forceinline__ uint32_t block_excl_sum(uint32_t v, uint32_t* sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d);
    if (lane >= d) inc += o;
  }
  __syncthreads();
  if (lane == 63) sh[w] = inc;
  __syncthreads();
  uint32_t before = 0;
  for (int j = 0; j < nw; ++j) before += j < w ? sh[j] : 0u;
  return before + inc - v;
}
__device__ __forceinline__ uint32_t block_excl_sum_256(uint32_t v, uint32_t* sh) { return block_excl_sum(v, sh); }

// 8 wavefronts per tile, each ranks an eighth of it (8 steps of 64): twice the wavefronts per CU of a 4-wavefront
// version at the same LDS, and half the serial chain per wavefront.  The per-wavefront digit counters / cursors are
// 16 bits wide (a tile has 4096 elements), two digits per LDS word, updated with 32-bit LDS atomics on the half.
constexpr int RS_SITEMS = RS_TILE / (64 * RS_SW);
static_assert(RS_TILE <= 65535 && RS_TILE % (64 * RS_SW) == 0, "16-bit cursors; whole steps");

__global__ void __launch_bounds__(64 * RS_SW) k_rs_scatter(const RsArgs a) {
  extern __shared__ __align__(16) unsigned char rs_smem[];
  uint2* sorted = reinterpret_cast<uint2*>(rs_smem);                       // the tile in digit order
  uint32_t* cnt = reinterpret_cast<uint32_t*>(sorted + RS_TILE);           // [RS_SW][B / 2] packed 16-bit count, then cursor
  int32_t* gpos = reinterpret_cast<int32_t*>(cnt + RS_SW * (RS_MAXB / 2)); // digit -> (global position - position in tile)
  uint32_t* sh = reinterpret_cast<uint32_t*>(gpos + RS_MAXB);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = blockIdx.y, tile = blockIdx.x;
  const int B = 1 << a.width, HB = B > 1 ? B / 2 : 1;  // words per wavefront
  constexpr int CW = RS_MAXB / 2;
  for (int j = threadIdx.x; j < RS_SW * HB; j += 64 * RS_SW) cnt[(j / HB) * CW + (j % HB)] = 0;
  const uint2* src = a.src + (size_t)r * a.m;
  uint2* dst = a.dst + (size_t)r * a.m;
  const int first = tile * RS_TILE + w * (RS_TILE / RS_SW);
  uint2 kvs[RS_SITEMS];
#pragma unroll
  for (int it = 0; it < RS_SITEMS; ++it) {
    const int idx = first + it * 64 + lane;
    kvs[it] = idx < a.m ? src[idx] : make_uint2(0u, 0u);
  }
  __syncthreads();
  uint32_t* mycnt = cnt + w * CW;
#pragma unroll
  for (int it = 0; it < RS_SITEMS; ++it) {
    if (first + it * 64 + lane < a.m) {
      const uint32_t d = (kvs[it].x >> a.shift) & (B - 1);
      atomicAdd(&mycnt[d >> 1], 1u << (16 * (d & 1u)));
    }
  }
  __syncthreads();
  // thread j owns the digit pairs [j * pp, (j + 1) * pp): where each digit starts in the tile, where in the whole
  // sequence; the counters become every wavefront's first position for the digit
  {
    const int npairs = HB, pp = npairs >= 64 * RS_SW ? npairs / (64 * RS_SW) : 1;
    const int p0 = threadIdx.x * pp;
    uint32_t mine = 0, all = 0;
    for (int j = 0; j < pp; ++j) {
      const int p = p0 + j;
      if (p < npairs) {
        for (int w2 = 0; w2 < RS_SW; ++w2) {
          const uint32_t c = cnt[w2 * CW + p];
          mine += (c & 0xffffu) + (c >> 16);
        }
        all += a.tot[r * B + 2 * p] + (2 * p + 1 < B ? a.tot[r * B + 2 * p + 1] : 0u);
      }
    }
    uint32_t tstart = block_excl_sum(mine, sh);
    uint32_t gstart = block_excl_sum(all, sh);
    for (int j = 0; j < pp; ++j) {
      const int p = p0 + j;
      if (p < npairs) {
        uint32_t c[RS_SW];
        for (int w2 = 0; w2 < RS_SW; ++w2) c[w2] = cnt[w2 * CW + p];
        // the even digit of the pair, then the odd one
        uint32_t lo_start = tstart, lo_tot = 0;
        for (int w2 = 0; w2 < RS_SW; ++w2) lo_tot += c[w2] & 0xffffu;
        uint32_t hi_start = tstart + lo_tot, hi_tot = 0;
        for (int w2 = 0; w2 < RS_SW; ++w2) hi_tot += c[w2] >> 16;
        const int d = 2 * p;
        const uint32_t* hrow = a.hist + ((size_t)r * a.tiles + tile) * B;
        gpos[d] = (int32_t)(gstart + hrow[d]) - (int32_t)lo_start;
        const uint32_t t_lo = a.tot[r * B + d];
        if (d + 1 < B) gpos[d + 1] = (int32_t)(gstart + t_lo + hrow[d + 1]) - (int32_t)hi_start;
        uint32_t run_lo = lo_start, run_hi = hi_start;
        for (int w2 = 0; w2 < RS_SW; ++w2) {
          cnt[w2 * CW + p] = run_lo | (run_hi << 16);
          run_lo += c[w2] & 0xffffu, run_hi += c[w2] >> 16;
        }
        tstart += lo_tot + hi_tot;
        gstart += t_lo + (d + 1 < B ? a.tot[r * B + d + 1] : 0u);
      }
    }
  }
  __syncthreads();
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int it = 0; it < RS_SITEMS; ++it) {
    const bool valid = first + it * 64 + lane < a.m;
    const uint2 kv = kvs[it];
    const uint32_t dg = (kv.x >> a.shift) & (B - 1);
    unsigned long long peers = __ballot(valid);
    for (int bit = 0; bit < a.width; ++bit) {
      const bool on = (dg >> bit) & 1u;
      const unsigned long long bb = __ballot(on);
      peers &= on ? bb : ~bb;
    }
    const uint32_t before = __popcll(peers & lt);
    const uint32_t sh16 = 16 * (dg & 1u);
    const uint32_t at = (mycnt[dg >> 1] >> sh16) & 0xffffu;
    __builtin_amdgcn_wave_barrier();
    if (valid && before == 0) atomicAdd(&mycnt[dg >> 1], (uint32_t)__popcll(peers) << sh16);  // the word's other half may move too
    __builtin_amdgcn_wave_barrier();
    if (valid) sorted[at + before] = kv;
  }
  __syncthreads();
  const int have = min(RS_TILE, a.m - tile * RS_TILE);
  for (int j = threadIdx.x; j < have; j += 64 * RS_SW) {
    const uint2 kv = sorted[j];
    dst[gpos[(kv.x >> a.shift) & (B - 1)] + j] = kv;
  }
}

__global__ void __launch_bounds__(256) k_epx_segments(const EpxState st, const EpxBatch b) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= st.n * st.num_keys) return;
  const int r = t / st.num_keys, k = t % st.num_keys;
  const uint2* a = b.kv_sorted + (size_t)r * b.m;
  auto lower = [&](uint32_t x) {
    int lo = 0, hi = b.m;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((a[mid].x & EPX_KEY_MASK) < x) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  b.seg[(size_t)t * 2] = lower((uint32_t)k);
  b.seg[(size_t)t * 2 + 1] = lower((uint32_t)(k + 1));
}

// the same from the sort's digit totals when the whole key was ONE digit (digit == key): the segment of key k starts
// where the digits below k end -- an exclusive scan of B totals per replica instead of 2 binary searches per segment
__global__ void __launch_bounds__(256) k_epx_segments_from_totals(const EpxState st, const EpxBatch b, const uint32_t* tot,
                                                                  int B) {
  __shared__ uint32_t sh[4];
  const int r = blockIdx.x, per = B >= 256 ? B / 256 : 1, d0 = threadIdx.x * per;
  uint32_t mine = 0;
  for (int j = 0; j < per; ++j)
    if (d0 + j < B) mine += tot[r * B + d0 + j];
  uint32_t start = block_excl_sum_256(mine, sh);
  for (int j = 0; j < per; ++j) {
    const int k = d0 + j;
    if (k < B) {
      const uint32_t c = tot[r * B + k];
      if (k < st.num_keys) {
        b.seg[((size_t)r * st.num_keys + k) * 2] = (int32_t)start;
        b.seg[((size_t)r * st.num_keys + k) * 2 + 1] = (int32_t)(start + c);
      }
      start += c;
    }
  }
}

// One chunk of 64 consecutive commands of a (replica, key) segment, one per lane: dep[l] = what the command's
// conflict lookup returns in column l (exclusive prefix over the chunk + carry), and the carries cg / cs (the
// replica's TopOne vectors for the key, KeyValueStore.scala:229-230) and ng / ns (this tick's puts alone) move on.
template <int N>
__device__ __forceinline__ void scan_chunk(bool valid, bool t, int L, int id1, int* cg, int* cs, int* ng, int* ns, int* dep) {
#pragma unroll
  for (int l = 0; l < N; ++l) {
    // inclusive prefix maxima of this chunk's puts into column l
    const int ig = wave_incl_max((valid && !t && L == l) ? id1 : 0);
    const int is = wave_incl_max((valid && t && L == l) ? id1 : 0);
    // exclusive prefix (the command's own put comes after its conflict lookup) + carry
    const int eg = imax(wave_shr1(ig), cg[l]);
    const int es = imax(wave_shr1(is), cs[l]);
    // KeyValueStore.scala:259-302: a get conflicts with sets, a set with sets and gets
    dep[l] = t ? imax(es, eg) : es;
    const int tg = __builtin_amdgcn_readlane(ig, 63), ts = __builtin_amdgcn_readlane(is, 63);
    cg[l] = imax(cg[l], tg), cs[l] = imax(cs[l], ts);
    ng[l] = imax(ng[l], tg), ns[l] = imax(ns[l], ts);
  }
}

// one wavefront per (replica, key) segment
template <int N>
__global__ void __launch_bounds__(256) k_epx_scan(const EpxState st, const EpxBatch b) {
  if (st.status[0] != 0) return;
  const int lane = threadIdx.x & 63;
  const int seg = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (seg >= N * st.num_keys) return;
  const int r = seg / st.num_keys, k = seg % st.num_keys;
  if (b.unfused && (*b.unfused == 0 || b.fused[k])) return;  // k_epx_key did this key on chip
  const int lo = b.seg[(size_t)seg * 2], hi = b.seg[(size_t)seg * 2 + 1];
  const uint2* kvs = b.kv_sorted + (size_t)r * b.m;
  constexpr int NP = ConfRow<N>::NP;
  int cg[N], cs[N];  // carry: the replica's TopOne vectors for this key (KeyValueStore.scala:229-230)
  const size_t ib = ((size_t)r * st.num_keys + k) * N;
#pragma unroll
  for (int l = 0; l < N; ++l) cg[l] = st.gets[ib + l], cs[l] = st.sets[ib + l];
  int ng[N], ns[N];  // this tick's puts alone: what the commit teaches the other replicas
#pragma unroll
  for (int l = 0; l < N; ++l) ng[l] = 0, ns[l] = 0;
  // the next chunk's sort pair and instance number are requested before this chunk is scanned: a segment is ~13
  // dependent round trips otherwise (pair -> number gather -> row store)
  uint2 kv_next = (lo + lane < hi) ? kvs[lo + lane] : make_uint2(0u, 0u);
  int num_next = (lo + lane < hi) ? b.number[kv_next.y] : 0;
  for (int base = lo; base < hi; base += 64) {
    const int p = base + lane;
    const bool valid = p < hi;
    const uint2 kv = kv_next;
    const int id1 = num_next + 1;  // TopOne.put: max(.., id + 1), util/TopOne.scala:12-15
    if (base + 64 < hi) {
      const int pn = base + 64 + lane;
      kv_next = pn < hi ? kvs[pn] : make_uint2(0u, 0u);
      num_next = pn < hi ? b.number[kv_next.y] : 0;
    }
    const int i = (int)kv.y;
    const int L = (int)(kv.x >> EPX_LEADER_SHIFT);
    const bool t = (kv.x >> EPX_SET_SHIFT) & 1u;
    int dep[NP];
#pragma unroll
    for (int l = 0; l < NP; ++l) dep[l] = 0;
    scan_chunk<N>(valid, t, L, id1, cg, cs, ng, ns, dep);
    if (valid) {
      int4* row = reinterpret_cast<int4*>(b.conf + ((size_t)i * N + r) * NP);
      row[0] = make_int4(dep[0], dep[1], dep[2], dep[3]);
      if constexpr (NP == 8) row[1] = make_int4(dep[4], dep[5], dep[6], dep[7]);
    }
  }
  if (lane == 0) {
    int32_t* out = b.tick + (size_t)seg * 2 * N;
#pragma unroll
    for (int l = 0; l < N; ++l) out[l] = ng[l], out[N + l] = ns[l];
  }
}

// handlePreAcceptOk for one command (Replica.scala:1291-1419): the leader's own conflicts D become the PreAccept's
// dependencies, every responder answers its conflicts U D (handlePreAccept :1257-1262), the fast path needs the n-2
// answers identical (popularItems), the slow path proposes their union (preAcceptingSlowPath :796-813).
// load_row(r, out[N]) yields replica r's conflicts for the command.  Writes the command log (if kept); returns the
// decision in o_deps / o_ldeps (watermarks, own column re-encoded) and o_end[2].
template <int N, typename Rows>
__device__ __forceinline__ bool epx_decide_core(const EpxState& st, const EpxBatch& b, int i, int L, unsigned mask,
                                                unsigned seen_in, int x, Rows&& load_row, int* o_deps, int* o_ldeps,
                                                int* o_end) {
  int D[N], uni[N], first[N];
  load_row(L, D);
#pragma unroll
  for (int l = 0; l < N; ++l) uni[l] = D[l], first[l] = 0;
  bool have_first = false, all_equal = true;
  for (int r = 0; r < N; ++r) {
    if (!((mask >> r) & 1u)) continue;
    bool same = true;
    int cr[N];
    load_row(r, cr);
#pragma unroll
    for (int l = 0; l < N; ++l) {
      const int cl = cr[l];
      const int resp = cl > D[l] ? cl : D[l];  // handlePreAccept: local conflicts U preAccept.dependencies
      uni[l] = resp > uni[l] ? resp : uni[l];  // preAcceptingSlowPath: union of all answers
      if (!have_first) first[l] = resp; else same = same && (first[l] == resp);
    }
    if (have_first) all_equal = all_equal && same;
    have_first = true;
  }
  // dependencies.subtractOne(instance) (Replica.scala:582, compact/IntPrefixSet.scala:388-398) only touches the
  // column of the instance's own leader: with w = that column's watermark before the subtraction and x the
  // instance number, the set is {0 .. w-1} \ {x}: (watermark w, no values) if x >= w, else (watermark x,
  // values x+1 .. w-1).  Unions (addAll) and the equality test of popularItems commute with that encoding for a
  // fresh instance (w == x + 1 would need the instance itself in the index), so everything above ran on the
  // plain watermarks w and only the own column is re-encoded here.
  if (st.num_instances > 0) {
    // the command log: a fast-path commit is a CommittedEntry at every replica (commit :815-823, Commit to the
    // others); otherwise every replica that processed the PreAccept holds PreAcceptedEntry(Ballot(0, leader),
    // Ballot(0, leader), triple) (:688-696, :1259-1271) for the Accept phase to find
    const unsigned seen = seen_in | (1u << L);
    const int tr = b.triple ? b.triple[i] : -1;
    for (int r = 0; r < N; ++r) {
      const size_t c = ((size_t)r * N + L) * st.num_instances + x;
      if (!all_equal && !((seen >> r) & 1u)) continue;
      // the triple's dependencies: the agreed ones (fast path, committed everywhere), else what THIS replica
      // answered (its conflicts U the PreAccept's, :1257-1271) / what the leader proposed (:688-696)
      int t[N];
      if (all_equal) {
#pragma unroll
        for (int l = 0; l < N; ++l) t[l] = first[l];
      } else if (r == L) {
#pragma unroll
        for (int l = 0; l < N; ++l) t[l] = D[l];
      } else {
        int cr[N];
        load_row(r, cr);
#pragma unroll
        for (int l = 0; l < N; ++l) t[l] = cr[l] > D[l] ? cr[l] : D[l];
      }
      int wm = 0, end = 0;
#pragma unroll
      for (int l = 0; l < N; ++l)
        if (l == L) own_column(t[l], x, &wm, &end);
#pragma unroll
      for (int l = 0; l < N; ++l) st.cl_deps[c * N + l] = l == L ? wm : t[l];
      st.cl_dend[c] = end;
      if (all_equal) {
        st.cl_status[c] = CL_COMMITTED, st.cl_ballot[c] = -1, st.cl_vote[c] = -1, st.cl_triple[c] = tr;
      } else {
        st.cl_status[c] = CL_PRE_ACCEPTED, st.cl_ballot[c] = L, st.cl_vote[c] = L, st.cl_triple[c] = tr;  // 0 * 8 + L
      }
    }
  }
  o_end[0] = 0, o_end[1] = 0;
#pragma unroll
  for (int l = 0; l < N; ++l) {
    const int w = all_equal ? first[l] : uni[l];
    int ow = w, oe = 0, lw = D[l], le = 0;
    if (l == L) {
      own_column(w, x, &ow, &oe), own_column(D[l], x, &lw, &le);
      o_end[0] = oe, o_end[1] = le;
    }
    o_deps[l] = ow;
    o_ldeps[l] = lw;
  }
  return all_equal;
}

// one thread per command; the n-wide dependency rows leave through LDS as contiguous lines (m x n ints written
// by 256 threads with a stride of n ints were 5 strided store instructions per wave)
template <int N>
__global__ void __launch_bounds__(256) k_epx_decide(const EpxState st, const EpxBatch b) {
  __shared__ int out_deps[256 * N], out_ldeps[256 * N];
  __shared__ uint8_t did[256];
  if (st.status[0] != 0) return;
  if (b.unfused && *b.unfused == 0) return;  // k_epx_key decided every key on chip
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool mine = i < b.m && !(b.unfused && b.fused[b.key[i]]);
  did[threadIdx.x] = mine ? 1 : 0;
  if (mine) {
    constexpr int NP = ConfRow<N>::NP;
    const int32_t* c = b.conf + (size_t)i * N * NP;
    auto load_row = [&](int r, int* out) {
      const int4* row = reinterpret_cast<const int4*>(c + r * NP);
      const int4 a = row[0];
      int tmp[8] = {a.x, a.y, a.z, a.w, 0, 0, 0, 0};
      if constexpr (NP == 8) {
        const int4 b2 = row[1];
        tmp[4] = b2.x, tmp[5] = b2.y, tmp[6] = b2.z, tmp[7] = b2.w;
      }
#pragma unroll
      for (int l = 0; l < N; ++l) out[l] = tmp[l];
    };
    int od[N], ol[N], oe[2];
    const unsigned mask = b.resp_mask[i];
    const bool fast = epx_decide_core<N>(st, b, i, b.leader[i], mask, b.seen_mask ? b.seen_mask[i] : mask, b.number[i],
                                         load_row, od, ol, oe);
    if (b.fast) b.fast[i] = fast ? 1 : 0;
    if (b.own_values_end) b.own_values_end[(size_t)i * 2] = oe[0], b.own_values_end[(size_t)i * 2 + 1] = oe[1];
#pragma unroll
    for (int l = 0; l < N; ++l) out_deps[threadIdx.x * N + l] = od[l], out_ldeps[threadIdx.x * N + l] = ol[l];
  }
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * blockDim.x * N;
  for (int k = threadIdx.x; k < 256 * N; k += 256) {
    if (did[k / N]) {
      if (b.deps) b.deps[base + k] = out_deps[k];
      if (b.leader_deps) b.leader_deps[base + k] = out_ldeps[k];
    }
  }
}

#include "fpx_epaxos_kp.hpp"
#include "fpx_depgraph_dev.hpp"
#include "fpx_depgraph_pk.hpp"

// ---- scan + decide of ONE key on chip ------------------------------------------------------------------------
// k_epx_scan hands every (command, replica) conflict row to k_epx_decide through HBM at [command][replica]: n * m
// rows written in (key, delivery order), i.e. at random -- 5 M random row writes are ~70 us on this GPU whatever
// their size (profiles/r02_random_rows.txt).  Here one workgroup owns a key: wavefront r scans replica r's segment
// of the key, the rows stay in LDS ([command of the key][replica][column]), and the same workgroup decides the
// key's commands; only the decisions cross to message-index order.  A command gets its slot in the LDS tables from
// the wavefront of its LEADER's replica (every command is in its leader's segment exactly once); the other
// wavefronts find the slot through an LDS hash table on the message index.  A key with more commands than the
// tables hold (KeyTile<N>::TC) is left to k_epx_scan / k_epx_decide (b.fused[k] = 0), so skewed workloads stay
// correct and merely lose the shortcut on their hot keys.
template <int N> struct KeyTile {
  static constexpr int TC = N <= 3 ? 2048 : N <= 5 ? 1152 : 576;   // commands of one key held on chip
  static constexpr int HC = N <= 3 ? 8192 : N <= 5 ? 4096 : 2048;  // hash slots: a power of two, load <= 0.28 (at 0.56
                                                                   // the probe chains cost 7 us per tick, n = 5)
  static constexpr int W = N <= 3 ? 4 : N <= 5 ? 3 : 2;             // wavefronts per replica segment
  static constexpr int MAXC = (TC + 63) / 64;                       // 64-command chunks of one segment
  static constexpr int CPW = (MAXC + W - 1) / W;                    // chunks per wavefront
  static constexpr int THREADS = 64 * N * W;
  static constexpr int IDX_BITS = 11;                               // TC <= 2048
  static constexpr size_t BYTES =
      (size_t)TC * N * N * 4 + (size_t)HC * 4 + (size_t)TC * 4 * 3 + (size_t)N * TC * 2 + (size_t)N * W * 2 * N * 4 + 64;
};

template <int N>
__global__ void __launch_bounds__(KeyTile<N>::THREADS) k_epx_key(const EpxState st, const EpxBatch b) {
  using T = KeyTile<N>;
  extern __shared__ __align__(16) unsigned char smem[];
  int* rows = reinterpret_cast<int*>(smem);                                   // [TC][N replicas][N columns]
  uint32_t* hk = reinterpret_cast<uint32_t*>(rows + (size_t)T::TC * N * N);   // (message index + 1) << 11 | slot
  int* list_i = reinterpret_cast<int*>(hk + T::HC);                           // slot -> message index
  int* num = list_i + T::TC;                                                  // slot -> instance number
  uint32_t* msk = reinterpret_cast<uint32_t*>(num + T::TC);                   // slot -> resp_mask | seen << 8 | leader << 16
  uint16_t* stage = reinterpret_cast<uint16_t*>(msk + T::TC);                 // [N][TC] slot | is_set << 11 | leader << 12
  int* tot = reinterpret_cast<int*>(stage + (size_t)N * T::TC);               // [N][W][2N] puts of a wavefront's chunks
  int* ctl = tot + N * T::W * 2 * N;                                          // [0] slots handed out, [1] give up
  if (st.status[0] != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = wave / T::W, w = wave - r * T::W;  // replica, part of its segment
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));

  // A workgroup works through the keys blockIdx.x, + gridDim.x, ...  With 149 KB of LDS it is alone on its CU, so
  // nothing else hides its memory round trips: what the NEXT key needs from memory -- segment bounds and the
  // replica's TopOne vectors (A), the sort pairs of this wavefront's part (B), number / masks of the commands it
  // leads (C), each addressed with the one before -- is requested while the current key is worked on.
  struct SegA {
    int lo, len, cpw, first;
    int cg[N], cs[N];
  };
  auto fetch_a = [&](int k, SegA& s) {
    const int seg = r * st.num_keys + k;
    s.lo = b.seg[(size_t)seg * 2];
    s.len = b.seg[(size_t)seg * 2 + 1] - s.lo;
    s.cpw = ((s.len + 63) / 64 + T::W - 1) / T::W;  // chunks per wavefront of THIS segment
    s.first = w * s.cpw * 64;                       // where this wavefront's part starts
    const size_t ib = (size_t)seg * N;
#pragma unroll
    for (int l = 0; l < N; ++l) s.cg[l] = st.gets[ib + l], s.cs[l] = st.sets[ib + l];
  };
  auto fetch_b = [&](const SegA& s, uint2* kv, bool* have) {
    const uint2* kvs = b.kv_sorted + (size_t)r * b.m + s.lo;
#pragma unroll
    for (int c = 0; c < T::CPW; ++c) {
      const int p = s.first + c * 64 + lane;
      have[c] = c < s.cpw && p < s.len && s.len <= T::MAXC * 64;
      kv[c] = have[c] ? kvs[p] : make_uint2(0u, 0u);
    }
  };
  auto fetch_c = [&](const uint2* kv, const bool* have, int* gnum, unsigned* gmask) {
#pragma unroll
    for (int c = 0; c < T::CPW; ++c) {
      gnum[c] = 0, gmask[c] = 0;
      if (have[c] && (int)(kv[c].x >> EPX_LEADER_SHIFT) == r) {
        const int2 mt = b.meta[kv[c].y];  // written by k_epx_keys: three arrays in one 8-byte gather
        gnum[c] = mt.x;
        gmask[c] = (unsigned)mt.y;
      }
    }
  };

  SegA cur, nxt;
  uint2 kvq[T::CPW], kvn[T::CPW];
  bool have[T::CPW], haven[T::CPW];
  int gnum[T::CPW], gnumn[T::CPW];
  unsigned gmask[T::CPW], gmaskn[T::CPW];
  int k = blockIdx.x;
  if (k >= st.num_keys) return;
  fetch_a(k, cur);
  fetch_b(cur, kvq, have);
  fetch_c(kvq, have, gnum, gmask);
  for (; k < st.num_keys; k += gridDim.x) {
    const int kn = k + gridDim.x;
    const bool more = kn < st.num_keys;
    if (more) fetch_a(kn, nxt);
    for (int j = threadIdx.x; j < T::HC; j += T::THREADS) hk[j] = 0;
    for (int j = threadIdx.x; j < N * T::W * 2 * N + 2; j += T::THREADS) tot[j] = 0;  // tot and ctl
    __syncthreads();
    if (cur.len > T::MAXC * 64 && lane == 0) ctl[1] = 1;
    // the wavefronts of the leader's replica hand out the slots of the commands it leads
#pragma unroll
    for (int c = 0; c < T::CPW; ++c) {
      const bool own = have[c] && (int)(kvq[c].x >> EPX_LEADER_SHIFT) == r;
      const unsigned long long bal = __ballot(own);
      int base = 0;
      if (bal) {
        if (lane == 0) base = atomicAdd(&ctl[0], (int)__popcll(bal));
        base = __builtin_amdgcn_readfirstlane(base);
      }
      const int sl = own ? base + (int)__popcll(bal & lt) : -1;
      if (sl >= T::TC) ctl[1] = 1;
      if (sl >= 0 && sl < T::TC) {
        const int i = (int)kvq[c].y;
        list_i[sl] = i;
        num[sl] = gnum[c];
        msk[sl] = gmask[c] | ((unsigned)r << 16);
        const uint32_t word = ((uint32_t)(i + 1) << T::IDX_BITS) | (uint32_t)sl;
        uint32_t h = ((uint32_t)i * 2654435761u >> 12) & (T::HC - 1);
        while (atomicCAS(&hk[h], 0u, word) != 0u) h = (h + 1) & (T::HC - 1);
      }
    }
    __syncthreads();
    const bool give_up = ctl[1] != 0;  // more commands than the tables hold: the key goes the long way
    if (give_up && threadIdx.x == 0) b.fused[k] = 0, atomicAdd(b.unfused, 1);
    uint16_t* mystage = stage + (size_t)r * T::TC + cur.first;
    if (!give_up) {
      // every command finds its slot; the scan below only needs (slot, is_set, leader) per lane.  The puts of this
      // wavefront's part (per column the largest id + 1) are the carry of the parts behind it.
      int* mytot = tot + (r * T::W + w) * 2 * N;
#pragma unroll
      for (int c = 0; c < T::CPW; ++c) {
        if (have[c]) {
          const uint32_t want = kvq[c].y + 1u;
          uint32_t h = (kvq[c].y * 2654435761u >> 12) & (T::HC - 1);
          uint32_t wd = hk[h];
          while ((wd >> T::IDX_BITS) != want) h = (h + 1) & (T::HC - 1), wd = hk[h];
          const uint32_t sl = wd & ((1u << T::IDX_BITS) - 1u);
          const uint32_t flags = kvq[c].x >> EPX_SET_SHIFT;  // bit 0 is_set, bits 1.. leader
          mystage[c * 64 + lane] = (uint16_t)(sl | (flags << T::IDX_BITS));
          atomicMax(&mytot[(flags & 1u) * N + (flags >> 1)], num[sl] + 1);
        }
      }
    }
    if (more) fetch_b(nxt, kvn, haven);  // the current key's pairs are spent
    __syncthreads();
    if (!give_up) {
      int cg[N], cs[N];
#pragma unroll
      for (int l = 0; l < N; ++l) cg[l] = cur.cg[l], cs[l] = cur.cs[l];
      for (int w2 = 0; w2 < w; ++w2) {
        const int* o = tot + (r * T::W + w2) * 2 * N;
#pragma unroll
        for (int l = 0; l < N; ++l) cg[l] = imax(cg[l], o[l]), cs[l] = imax(cs[l], o[N + l]);
      }
      if (w == 0 && lane == 0) {  // what the commit teaches the other replicas: this tick's puts alone
        int32_t* out = b.tick + ((size_t)r * st.num_keys + k) * 2 * N;
        for (int l = 0; l < 2 * N; ++l) {
          int v = 0;
          for (int w2 = 0; w2 < T::W; ++w2) v = imax(v, tot[(r * T::W + w2) * 2 * N + l]);
          out[l] = v;
        }
      }
      int ng[N], ns[N];
#pragma unroll
      for (int l = 0; l < N; ++l) ng[l] = 0, ns[l] = 0;
      for (int base = 0; base < cur.cpw * 64 && cur.first + base < cur.len; base += 64) {
        const bool valid = cur.first + base + lane < cur.len;
        const unsigned code = valid ? mystage[base + lane] : 0u;
        const int sl = (int)(code & ((1u << T::IDX_BITS) - 1u));
        const bool t = (code >> T::IDX_BITS) & 1u;
        const int L = (int)(code >> (T::IDX_BITS + 1));
        const int id1 = valid ? num[sl] + 1 : 0;  // TopOne.put: max(.., id + 1), util/TopOne.scala:12-15
        int dep[N];
        scan_chunk<N>(valid, t, L, id1, cg, cs, ng, ns, dep);
        if (valid) {
#pragma unroll
          for (int l = 0; l < N; ++l) rows[((size_t)sl * N + r) * N + l] = dep[l];
        }
      }
    }
    if (more) fetch_c(kvn, haven, gnumn, gmaskn);
    __syncthreads();
    if (!give_up) {
      const int count = ctl[0];
      for (int sl = threadIdx.x; sl < count; sl += T::THREADS) {
        const int i = list_i[sl];
        const unsigned mw = msk[sl];
        auto load_row = [&](int rr, int* out) {
#pragma unroll
          for (int l = 0; l < N; ++l) out[l] = rows[((size_t)sl * N + rr) * N + l];
        };
        int od[N], ol[N], oe[2];
        const bool fast = epx_decide_core<N>(st, b, i, (int)(mw >> 16), mw & 0xffu, (mw >> 8) & 0xffu, num[sl], load_row,
                                             od, ol, oe);
        if (b.fast) b.fast[i] = fast ? 1 : 0;
        if (b.own_values_end) *reinterpret_cast<int2*>(b.own_values_end + (size_t)i * 2) = make_int2(oe[0], oe[1]);
        // the command's rows are spent: its decision takes their place, to leave below as whole n-int lines (one
        // store instruction per column and thread here would touch 64 different lines each)
#pragma unroll
        for (int l = 0; l < N; ++l) rows[((size_t)sl * N + 0) * N + l] = od[l], rows[((size_t)sl * N + 1) * N + l] = ol[l];
      }
      __syncthreads();
      for (int t = threadIdx.x; t < count * N; t += T::THREADS) {
        const int sl = t / N, l = t - sl * N;
        const size_t o = (size_t)list_i[sl] * N + l;
        if (b.deps) b.deps[o] = rows[((size_t)sl * N + 0) * N + l];
        if (b.leader_deps) b.leader_deps[o] = rows[((size_t)sl * N + 1) * N + l];
      }
      if (threadIdx.x == 0) b.fused[k] = 1;
    }
    __syncthreads();  // the tables are reused by the next key
    cur = nxt;
#pragma unroll
    for (int c = 0; c < T::CPW; ++c) kvq[c] = kvn[c], have[c] = haven[c], gnum[c] = gnumn[c], gmask[c] = gmaskn[c];
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
  int tg = 0, ts = 0;
  for (int q = 0; q < st.n; ++q) {
    const int32_t* seen = b.tick + (((size_t)q * st.num_keys + k) * 2) * st.n;
    tg = imax(tg, seen[l]), ts = imax(ts, seen[st.n + l]);
  }
  int32_t* g = &st.gets[(size_t)r * per + e];
  int32_t* s2 = &st.sets[(size_t)r * per + e];
  if (tg > *g) *g = tg;
  if (ts > *s2) *s2 = ts;
}

// the four output arrays -> one packed line per command (fpx_epx_preaccept_packed_dev through the first form): 256
// commands per workgroup, every array read and the lines written with consecutive lanes on consecutive words
__global__ void __launch_bounds__(256) k_epx_pack(const EpxState st, int m, const uint8_t* fast, const int32_t* deps,
                                                  const int32_t* ldeps, const int32_t* own, int32_t* packed, int stride) {
  __shared__ int32_t line[256 * 20];
  if (st.status[0] != 0) return;
  const int n = st.n, i0 = blockIdx.x * 256, cnt = min(256, m - i0);
  for (int t = threadIdx.x; t < cnt * stride; t += 256) line[t] = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < cnt * n; t += 256) {
    const int c = t / n, l = t - c * n;
    line[c * stride + l] = deps[(size_t)i0 * n + t], line[c * stride + n + l] = ldeps[(size_t)i0 * n + t];
  }
  for (int t = threadIdx.x; t < cnt * 2; t += 256) line[(t >> 1) * stride + 2 * n + (t & 1)] = own[(size_t)i0 * 2 + t];
  if ((int)threadIdx.x < cnt) line[threadIdx.x * stride + 2 * n + 2] = fast[i0 + threadIdx.x] ? 1 : 0;
  __syncthreads();
  for (int t = threadIdx.x; t < cnt * stride; t += 256) packed[(size_t)i0 * stride + t] = line[t];
}

// ---- the per-instance Paxos of EPaxos on the command log: Prepare (phase 1) and Accept (phase 2) ---------------
// Messages are delivered in array order to the replicas of target[i]; the instances of a batch are pairwise
// distinct, so every command-log cell has one writer and the only order-dependent quantity is each replica's
// running largestBallot (Replica.scala:458), which only Nacks report: an inclusive prefix max per replica over the
// ballots the replica took in (k_cl_tilemax / k_cl_tilescan / k_cl_nacks).
struct ClBatch {
  int m, accept;           // accept = 0: Prepare
  const int32_t* leader;
  const int32_t* number;
  const int32_t* b_ord;
  const int32_t* b_rep;
  const int32_t* triple;   // Accept
  const int32_t* key;      // Accept: the command's key, -1 = Noop
  const uint8_t* is_set;   // Accept
  const uint8_t* target;
  uint8_t* ok_bits;
  uint8_t* nack_bits;
  uint8_t* commit_bits;
  int32_t* nack_ballot;
  uint8_t* committed;      // Accept
  int32_t* reply_status;   // Prepare: [m][n]
  int32_t* reply_vote;
  int32_t* reply_triple;
  int32_t* contrib;        // [n][m] scratch: the ballot replica r took in with message i, or -1
  uint8_t* nackflag;       // [n][m] scratch
  int32_t* tilemax;        // [n][tiles] scratch
  uint8_t* skip;           // [m] scratch: Accept whose proposer refused it (logger.fatal / checkLe)
  uint32_t run_id;
};
constexpr int CL_TILE = 1024;

// updateConflictIndex(instance, commandOrNoop) (Replica.scala:602-614) at replica r: conflictIndex.put = a max-merge of
// the TopOne column of the instance's leader (util/TopOne.scala:14-17), so several messages of one batch may land on
// one cell in any order; a Noop leaves the index alone
__device__ __forceinline__ void index_put(const EpxState& st, int r, int key, int is_set, int L, int x) {
  if (key < 0) return;
  int32_t* a = is_set ? st.sets : st.gets;
  atomicMax(&a[((size_t)r * st.num_keys + key) * st.n + L], x + 1);
}

// an Accept names its triple by the caller's id alone: the stored dependencies are marked unknown
__device__ __forceinline__ void deps_by_id(const EpxState& st, size_t cell) {
  for (int l = 0; l < st.n; ++l) st.cl_deps[cell * st.n + l] = -1;
  st.cl_dend[cell] = 0;
}

__global__ void __launch_bounds__(256) k_cl_validate(const EpxState st, const ClBatch b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.m) return;
  const int n = st.n, L = b.leader[i], x = b.number[i], bo = b.b_ord[i], br = b.b_rep[i];
  bool ok = L >= 0 && L < n && x >= 0 && x < st.num_instances && bo >= 0 && bo < (1 << 27) && br >= 0 && br < n &&
            (b.target[i] >> n) == 0;
  if (ok && b.accept) ok = !((b.target[i] >> br) & 1u);  // thriftyOtherReplicas: never the proposer itself (:774)
  if (ok && b.accept) ok = b.key[i] >= -1 && b.key[i] < st.num_keys;
  if (ok) ok = atomicExch(&st.cl_stamp[(size_t)L * st.num_instances + x], b.run_id) != b.run_id;
  if (!ok) epx_report(st.status, FPX_EINVAL, i);
}

// Accept: transitionToAcceptPhase at the proposer (Replica.scala:732-792), one thread per message
__global__ void __launch_bounds__(256) k_cl_propose(const EpxState st, const ClBatch b) {
  if (st.status[0] == FPX_EINVAL) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.m) return;
  const int n = st.n, P = b.b_rep[i], ballot = b.b_ord[i] * 8 + P;
  const size_t c = ((size_t)P * n + b.leader[i]) * st.num_instances + b.number[i];
  const int kind = st.cl_status[c];
  // :740-744 a CommittedEntry is logger.fatal; :749-757 logger.checkLe(entry.ballot / voteBallot, ballot)
  const bool refuse = kind == CL_COMMITTED || (kind != CL_NONE && st.cl_ballot[c] > ballot) ||
                      (kind >= CL_PRE_ACCEPTED && st.cl_vote[c] > ballot);
  b.skip[i] = refuse ? 1 : 0;
  if (refuse) {
    if (atomicCAS(&st.status[0], 0, FPX_EFATAL_PROTOCOL) == 0) st.status[1] = i;
    return;
  }
  st.cl_status[c] = CL_ACCEPTED, st.cl_ballot[c] = ballot, st.cl_vote[c] = ballot, st.cl_triple[c] = b.triple[i];  // :759-762
  deps_by_id(st, c);
  index_put(st, P, b.key[i], b.is_set[i], b.leader[i], b.number[i]);  // :763
}

// handlePrepare (:1632-1757) / handleAccept (:1421-1511) at replica r for message i: one thread per (i, r)
__global__ void __launch_bounds__(256) k_cl_handle(const EpxState st, const ClBatch b) {
  if (st.status[0] == FPX_EINVAL) return;
  const int n = st.n;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)b.m * n) return;
  const int r = (int)(t / b.m), i = (int)(t % b.m);  // replica-major: the scratch rows are contiguous per replica
  const int ballot = b.b_ord[i] * 8 + b.b_rep[i];
  int contrib = -1, rs = -1, rv = -1, rt = -1;
  uint8_t nack = 0;
  const bool live = ((b.target[i] >> r) & 1u) && !(b.accept && b.skip[i]);
  if (live) {
    const size_t c = ((size_t)r * n + b.leader[i]) * st.num_instances + b.number[i];
    const int kind = st.cl_status[c];
    if (!b.accept) contrib = ballot;  // :1637 largestBallot = max(.., prepare.ballot) before anything else
    if (kind == CL_COMMITTED) {
      atomicOr(reinterpret_cast<unsigned int*>(b.commit_bits) + (i >> 2), (1u << r) << (8 * (i & 3)));
    } else if (kind != CL_NONE && ballot < st.cl_ballot[c]) {
      nack = 1;  // Nack(instance, largestBallot): the value comes from the scan
      atomicOr(reinterpret_cast<unsigned int*>(b.nack_bits) + (i >> 2), (1u << r) << (8 * (i & 3)));
    } else {
      atomicOr(reinterpret_cast<unsigned int*>(b.ok_bits) + (i >> 2), (1u << r) << (8 * (i & 3)));
      if (!b.accept) {
        if (kind == CL_NONE || kind == CL_NO_COMMAND) {  // :1654-1669, :1686-1701
          rs = 0;
          st.cl_status[c] = CL_NO_COMMAND, st.cl_vote[c] = -1, st.cl_triple[c] = -1;
        } else {  // :1711-1743 the entry keeps its vote, only `ballot` moves
          rs = kind, rv = st.cl_vote[c], rt = st.cl_triple[c];
        }
        st.cl_ballot[c] = ballot;
      } else if (!(kind == CL_ACCEPTED && ballot == st.cl_vote[c])) {  // (:1451-1461: already answered, re-send only)
        contrib = ballot;  // :1487
        st.cl_status[c] = CL_ACCEPTED, st.cl_ballot[c] = ballot, st.cl_vote[c] = ballot, st.cl_triple[c] = b.triple[i];
        deps_by_id(st, c);
        index_put(st, r, b.key[i], b.is_set[i], b.leader[i], b.number[i]);  // :1503
      }
    }
  }
  b.contrib[(size_t)r * b.m + i] = contrib;
  b.nackflag[(size_t)r * b.m + i] = nack;
  if (!b.accept) {
    if (b.reply_status) b.reply_status[(size_t)i * n + r] = rs;
    if (b.reply_vote) b.reply_vote[(size_t)i * n + r] = rv;
    if (b.reply_triple) b.reply_triple[(size_t)i * n + r] = rt;
  }
}

__device__ __forceinline__ int block_max_256(int v, int* sh) {
#pragma unroll
  for (int k = 1; k < 64; k <<= 1) v = imax(v, __shfl_xor(v, k));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const int out = imax(imax(sh[0], sh[1]), imax(sh[2], sh[3]));
  __syncthreads();
  return out;
}

__global__ void __launch_bounds__(256) k_cl_tilemax(const ClBatch b, int tiles) {
  __shared__ int sh[4];
  const int r = blockIdx.y, t = blockIdx.x;
  int v = -1;
  for (int k = threadIdx.x; k < CL_TILE; k += 256) {
    const int i = t * CL_TILE + k;
    if (i < b.m) v = imax(v, b.contrib[(size_t)r * b.m + i]);
  }
  v = block_max_256(v, sh);
  if (threadIdx.x == 0) b.tilemax[(size_t)r * tiles + t] = v;
}

// one block per replica: tilemax -> what the replica had seen BEFORE each tile; the replica's new largestBallot
__global__ void __launch_bounds__(256) k_cl_tilescan(const EpxState st, const ClBatch b, int tiles) {
  if (st.status[0] == FPX_EINVAL) return;
  const int r = blockIdx.x;
  if (threadIdx.x != 0) return;  // tiles <= a few thousand: a serial walk of one thread is microseconds
  int run = st.largest[r];
  for (int t = 0; t < tiles; ++t) {
    const int v = b.tilemax[(size_t)r * tiles + t];
    b.tilemax[(size_t)r * tiles + t] = run;
    run = imax(run, v);
  }
  st.largest[r] = run;
}

// the largestBallot a Nack of (i, r) carries = max(before the tile, inclusive prefix inside the tile)
__global__ void __launch_bounds__(256) k_cl_nacks(const ClBatch b, int tiles) {
  __shared__ int wmax[4];
  const int r = blockIdx.y, t = blockIdx.x;
  int carry = b.tilemax[(size_t)r * tiles + t];
  for (int k0 = 0; k0 < CL_TILE; k0 += 256) {
    const int i = t * CL_TILE + k0 + threadIdx.x;
    const int v = i < b.m ? b.contrib[(size_t)r * b.m + i] : -1;
    int inc = wave_incl_max(v + 1) - 1;  // the DPP scan's identity is 0: shift ballots (>= -1) up by one
    if ((threadIdx.x & 63) == 63) wmax[threadIdx.x >> 6] = inc;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before = imax(before, wmax[w]);
    const int total = imax(imax(wmax[0], wmax[1]), imax(wmax[2], wmax[3]));
    inc = imax(inc, before);
    if (i < b.m && b.nackflag[(size_t)r * b.m + i] && b.nack_ballot) atomicMax(&b.nack_ballot[i], inc);
    carry = imax(carry, total);
    __syncthreads();
  }
}

// handleAcceptOk (:1513-1565): f + 1 responses, the proposer's own included -> commit (:815-860) at every replica
__global__ void __launch_bounds__(256) k_cl_commit(const EpxState st, const ClBatch b) {
  if (st.status[0] == FPX_EINVAL) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.m) return;
  const int n = st.n, f = (n - 1) / 2;
  uint8_t done = 0;
  if (!b.skip[i]) {
    const unsigned ok = b.ok_bits[i] | (1u << b.b_rep[i]);  // :780-789 the proposer's own AcceptOk
    b.ok_bits[i] = (uint8_t)ok;
    if ((int)__popc(ok) >= f + 1) {
      done = 1;
      for (int r = 0; r < n; ++r) {
        const size_t c = ((size_t)r * n + b.leader[i]) * st.num_instances + b.number[i];
        st.cl_status[c] = CL_COMMITTED, st.cl_ballot[c] = -1, st.cl_vote[c] = -1, st.cl_triple[c] = b.triple[i];
        deps_by_id(st, c);
        index_put(st, r, b.key[i], b.is_set[i], b.leader[i], b.number[i]);  // commit :828, at every replica (Commit)
      }
    }
  }
  if (b.committed) b.committed[i] = done;
}

// Replica.handleCommit (:1567-1575) -> commit (:815-830) at replica r for message i: one thread per (i, r).  A Commit is
// final: no ballot is looked at, whatever entry the replica held is replaced (:826-827), the conflict index learns the
// command (:828).
struct LcBatch {
  int m;
  const int32_t* leader;
  const int32_t* number;
  const int32_t* triple;
  const int32_t* key;
  const uint8_t* is_set;
  const int32_t* deps;      // [m][n] or null: the triple is known by its id alone
  const int32_t* deps_end;  // [m] or null
  const uint8_t* target;
  const uint8_t* writer;    // [m] the replicas whose ENTRY this message writes: target minus the replicas a later message of the
                            // batch for the same instance goes to -- of several Commits for one instance the last one wins
                            // WHOLE (triple, dependencies and all), as when the reference applies them in order; cell by cell
                            // from different threads the entry could end up with one message's triple and another's
                            // dependencies
};
__global__ void __launch_bounds__(256) k_cl_learn_commit(const EpxState st, const LcBatch b) {
  const int n = st.n;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)b.m * n) return;
  const int r = (int)(t / b.m), i = (int)(t % b.m);
  if (!((b.target[i] >> r) & 1u)) return;
  if ((b.writer[i] >> r) & 1u) {
    const size_t c = ((size_t)r * n + b.leader[i]) * st.num_instances + b.number[i];
    st.cl_status[c] = CL_COMMITTED, st.cl_ballot[c] = -1, st.cl_vote[c] = -1, st.cl_triple[c] = b.triple[i];
    if (b.deps) {
      for (int l = 0; l < n; ++l) st.cl_deps[c * n + l] = b.deps[(size_t)i * n + l];
      st.cl_dend[c] = b.deps_end ? b.deps_end[i] : 0;
    } else {
      deps_by_id(st, c);
    }
  }
  index_put(st, r, b.key[i], b.is_set[i], b.leader[i], b.number[i]);  // (a maximum: every message's put, in any order)
}

// ---- K8: Replica.handlePrepareOk (Replica.scala:1759-1884), the recovering replica's decision -- one thread per instance
struct RcBatch {
  int m, as_intended;
  const int32_t* leader;
  const int32_t* number;
  const int32_t* b_ord;
  const int32_t* b_rep;
  const uint8_t* resp_mask;
  const int32_t* rs;  // [m][n] PrepareOk.status as the entry kind
  const int32_t* rv;  // [m][n] PrepareOk.voteBallot, encoded
  const int32_t* rt;  // [m][n] triple id
  int32_t* action;
  int32_t* source;
  int32_t* triple;
};

__global__ void __launch_bounds__(256) k_cl_recover(const EpxState st, const RcBatch b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.m) return;
  const int n = st.n, f = (n - 1) / 2;
  const int L = b.leader[i], x = b.number[i], me = b.b_rep[i];
  const unsigned mask = b.resp_mask[i];
  bool ok = L >= 0 && L < n && x >= 0 && x < st.num_instances && me >= 0 && me < n && b.b_ord[i] >= 0 && (mask >> n) == 0;
  const int32_t* rs = b.rs + (size_t)i * n;
  const int32_t* rv = b.rv + (size_t)i * n;
  const int32_t* rt = b.rt + (size_t)i * n;
  for (int r = 0; ok && r < n; ++r)
    if (((mask >> r) & 1u) && rs[r] < 0) ok = false;  // no PrepareOk from r
  if (!ok) {
    epx_report(st.status, FPX_EINVAL, i);
    return;
  }
  if (st.status[0] != 0) return;
  int act = 0, src = -1, tr = -1;
  if ((int)__popc(mask) >= f + 1) {  // :1799 responses.size < slowQuorumSize -> wait
    int maxvb = -2;                  // :1805-1807 only the responses of the highest voteBallot count
    for (int r = 0; r < n; ++r)
      if (((mask >> r) & 1u) && rv[r] > maxvb) maxvb = rv[r];
    if (b.as_intended) {             // :1810-1824 -- never true as the reference evaluates it (status is no Option)
      for (int r = 0; r < n && act == 0; ++r)
        if (((mask >> r) & 1u) && rv[r] == maxvb && rs[r] == CL_ACCEPTED) act = 1, src = r, tr = rt[r];
    }
    if (act == 0) {
      // :1830-1851 f matching PreAccepted triples of the default ballot, not from the recovering replica.  The
      // reference tests the ballot of the Prepare that was answered (p.ballot), which is the recovery ballot
      const int dflt = L;  // Ballot(0, leader) encoded: 0 * 8 + L
      const bool in_default = b.as_intended ? maxvb == dflt : (b.b_ord[i] * 8 + me) == dflt;
      for (int r = 0; r < n && act == 0 && in_default; ++r) {
        if (!((mask >> r) & 1u) || rv[r] != maxvb || rs[r] != CL_PRE_ACCEPTED || r == me) continue;
        const size_t cr = ((size_t)r * n + L) * st.num_instances + x;
        int same = 0;
        for (int q = 0; q < n; ++q) {
          if (!((mask >> q) & 1u) || rv[q] != maxvb || rs[q] != CL_PRE_ACCEPTED || q == me) continue;
          const size_t cq = ((size_t)q * n + L) * st.num_instances + x;
          // (the triples' dependencies are read from the command log, not from the replies: the caller keeps these
          // instances untouched between fpx_epx_prepare and this call -- include/fpx.h)
          bool eq = rt[q] == rt[r] && st.cl_dend[cq] == st.cl_dend[cr];
          for (int l = 0; l < n && eq; ++l) eq = st.cl_deps[cq * n + l] == st.cl_deps[cr * n + l];
          same += eq ? 1 : 0;
        }
        if (same >= f) act = 1, src = r, tr = rt[r];  // Util.popularItems(.., config.f)
      }
    }
    if (act == 0) {  // :1856-1868 start over, avoiding the fast path: with a pre-accepted command, else with a Noop
      act = 3;
      for (int r = 0; r < n && act == 3; ++r)
        if (((mask >> r) & 1u) && rv[r] == maxvb && rs[r] == CL_PRE_ACCEPTED) act = 2, src = r, tr = rt[r];
    }
  }
  if (b.action) b.action[i] = act;
  if (b.source) b.source[i] = src;
  if (b.triple) b.triple[i] = tr;
}

// ---- handlePreAccept in full (Replica.scala:1159-1289): ballots, Nacks, re-sent replies ------------------------
// Like Prepare / Accept above: messages delivered in array order to the replicas of target[i], instances pairwise
// distinct per batch -- so what a replica does with a message (process / Nack / answer again / ignore / answer with
// the Commit) follows from its command-log entry alone (k_hp_gate); the messages a replica processes run through
// K5's conflict scan in array order (sort by key, segmented prefix max) and k_hp_reply forms the PreAcceptOk's.
enum { HP_NONE = 0, HP_PROCESS = 1, HP_NACK = 2, HP_RESEND = 3, HP_IGNORE = 4, HP_COMMIT = 5 };
struct HpBatch {
  int m;
  const int32_t* leader;
  const int32_t* number;
  const int32_t* b_ord;
  const int32_t* b_rep;
  const int32_t* key;       // -1 = Noop
  const uint8_t* is_set;
  const int32_t* triple;    // may be null
  const int32_t* deps_in;   // [m][n]
  const int32_t* dend_in;   // [m] or null
  const uint8_t* target;
  uint8_t* ok_bits;
  uint8_t* resend_bits;
  uint8_t* nack_bits;
  uint8_t* commit_bits;
  int32_t* reply_deps;      // [m][n][n]
  int32_t* reply_end;       // [m][n]
  int32_t* reply_triple;    // [m][n]
  uint8_t* act;             // [n][m] scratch
  int32_t* contrib;         // [n][m] scratch (k_cl_tilemax / k_cl_nacks)
  uint8_t* nackflag;        // [n][m] scratch
  uint2* kv;                // [n][m] sort pairs
  const int32_t* conf;      // [m][n][NP] from k_epx_scan
  const int32_t* tick;      // [n][num_keys][2][n] from k_epx_scan
  uint32_t run_id;
};

__global__ void __launch_bounds__(256) k_hp_validate(const EpxState st, const HpBatch b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.m) return;
  const int n = st.n, L = b.leader[i], x = b.number[i], bo = b.b_ord[i], br = b.b_rep[i], k = b.key[i];
  bool ok = L >= 0 && L < n && x >= 0 && x < st.num_instances && bo >= 0 && bo < (1 << 27) && br >= 0 && br < n &&
            (b.target[i] >> n) == 0 && k >= -1 && k < st.num_keys;
  if (ok) {
    for (int l = 0; l < n; ++l) ok = ok && b.deps_in[(size_t)i * n + l] >= 0;
    // a PreAccept never depends on its own instance (:582); explicit values are the run number + 1 .. end - 1
    const int w = b.deps_in[(size_t)i * n + L], end = b.dend_in ? b.dend_in[i] : 0;
    ok = ok && (end == 0 ? w <= x : (w == x && end >= x + 2));
  }
  if (ok) ok = atomicExch(&st.cl_stamp[(size_t)L * st.num_instances + x], b.run_id) != b.run_id;
  if (!ok) epx_report(st.status, FPX_EINVAL, i);
}

// one thread per (replica, message), replica-major
__global__ void __launch_bounds__(256) k_hp_gate(const EpxState st, const HpBatch b) {
  const int n = st.n;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)b.m * n) return;
  const int r = (int)(t / b.m), i = (int)(t % b.m);
  int act = HP_NONE;
  const int L = b.leader[i], ballot = b.b_ord[i] * 8 + b.b_rep[i], k = b.key[i];
  // (a malformed message is reported by k_hp_validate; here it only must not index out of bounds)
  if (((b.target[i] >> r) & 1u) && L >= 0 && L < n && b.number[i] >= 0 && b.number[i] < st.num_instances) {
    const size_t c = ((size_t)r * n + L) * st.num_instances + b.number[i];
    const int kind = st.cl_status[c];
    if (kind == CL_COMMITTED) act = HP_COMMIT;                                        // :1227-1238
    else if (kind != CL_NONE && ballot < st.cl_ballot[c]) act = HP_NACK;              // :1180-1184, 1189-1192, 1215-1218
    else if (kind == CL_PRE_ACCEPTED && ballot == st.cl_vote[c]) act = HP_RESEND;     // :1196-1210
    else if (kind == CL_ACCEPTED && ballot == st.cl_vote[c]) act = HP_IGNORE;         // :1222-1224
    else act = HP_PROCESS;
  }
  const size_t o = (size_t)r * b.m + i;
  b.act[o] = (uint8_t)act;
  b.contrib[o] = act == HP_PROCESS ? ballot : -1;   // :1251 largestBallot = max(largestBallot, preAccept.ballot)
  b.nackflag[o] = act == HP_NACK ? 1 : 0;
  // only what the replica processes takes part in its conflict scan; the rest (and Noops, :592-593) sorts last
  const bool scanned = act == HP_PROCESS && k >= 0 && k < st.num_keys;
  const uint32_t flags = ((uint32_t)(b.is_set[i] ? 1 : 0) << EPX_SET_SHIFT) | ((uint32_t)(L & 7) << EPX_LEADER_SHIFT);
  b.kv[o] = make_uint2((scanned ? (uint32_t)k : (uint32_t)st.num_keys) | flags, (uint32_t)i);
}

// one thread per (message, replica): the reply and the new command-log entry
template <int N>
__global__ void __launch_bounds__(256) k_hp_reply(const EpxState st, const HpBatch b) {
  if (st.status[0] == FPX_EINVAL) return;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)b.m * N) return;
  const int i = (int)(t / N), r = (int)(t % N);
  constexpr int NP = ConfRow<N>::NP;
  const int act = b.act[(size_t)r * b.m + i];
  const int L = b.leader[i], x = b.number[i];
  const size_t c = ((size_t)r * N + L) * st.num_instances + x;
  int out[N], end = 0, tr = -1;
#pragma unroll
  for (int l = 0; l < N; ++l) out[l] = 0;
  uint8_t* bits = nullptr;
  if (act == HP_PROCESS) {
    const int ballot = b.b_ord[i] * 8 + b.b_rep[i];
    int row[N];
#pragma unroll
    for (int l = 0; l < N; ++l) row[l] = 0;
    if (b.key[i] >= 0) {  // computeSequenceNumberAndDependencies :569-600: the conflicts the scan found
      const int32_t* cr = b.conf + ((size_t)i * N + r) * NP;
#pragma unroll
      for (int l = 0; l < N; ++l) row[l] = cr[l];
    }
    const int in_end = b.dend_in ? b.dend_in[i] : 0;
#pragma unroll
    for (int l = 0; l < N; ++l) {
      const int in = b.deps_in[(size_t)i * N + l];
      if (l == L) {
        // local {0 .. row-1} \ {x}  U  the message's own column (cover in_end, or its plain watermark): :582, :1257-1262
        const int cover = imax(row[l], in_end ? in_end : in);
        own_column(cover, x, &out[l], &end);
      } else {
        out[l] = imax(row[l], in);
      }
    }
    tr = b.triple ? b.triple[i] : -1;
    st.cl_status[c] = CL_PRE_ACCEPTED, st.cl_ballot[c] = ballot, st.cl_vote[c] = ballot, st.cl_triple[c] = tr;  // :1265-1276
#pragma unroll
    for (int l = 0; l < N; ++l) st.cl_deps[c * N + l] = out[l];
    st.cl_dend[c] = end;
    bits = b.ok_bits;
  } else if (act == HP_RESEND || act == HP_COMMIT) {
#pragma unroll
    for (int l = 0; l < N; ++l) out[l] = st.cl_deps[c * N + l];
    end = st.cl_dend[c], tr = st.cl_triple[c];
    bits = act == HP_RESEND ? b.resend_bits : b.commit_bits;
  } else if (act == HP_NACK) {
    bits = b.nack_bits;
  }
  if (bits) atomicOr(reinterpret_cast<unsigned int*>(bits) + (i >> 2), (1u << r) << (8 * (i & 3)));
  if (b.reply_deps) {
#pragma unroll
    for (int l = 0; l < N; ++l) b.reply_deps[((size_t)i * N + r) * N + l] = out[l];
  }
  if (b.reply_end) b.reply_end[(size_t)i * N + r] = end;
  if (b.reply_triple) b.reply_triple[(size_t)i * N + r] = tr;
}

// updateConflictIndex (:1279) at the replicas that processed: replica r's index learns what ITS scan saw
__global__ void __launch_bounds__(256) k_hp_commit(const EpxState st, const HpBatch b) {
  if (st.status[0] == FPX_EINVAL) return;
  const long long per = (long long)st.num_keys * st.n;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per * st.n) return;
  const int r = (int)(t / per);
  const long long e = t % per;
  const int k = (int)(e / st.n), l = (int)(e % st.n);
  const int32_t* seen = b.tick + (((size_t)r * st.num_keys + k) * 2) * st.n;
  int32_t* g = &st.gets[(size_t)r * per + e];
  int32_t* s2 = &st.sets[(size_t)r * per + e];
  if (seen[l] > *g) *g = seen[l];
  if (seen[st.n + l] > *s2) *s2 = seen[st.n + l];
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
  Buf kv, kv2, seg, conf, tmp, tick, h_leader, h_number, h_key, h_set, h_mask, h_seen, h_rank, h_triple, o_fast, o_deps, o_ldeps, o_own, cl, hp, fusedb, metab;
  Buf p_fast, p_deps, p_ldeps, p_own;      // the four output arrays when a packed tick goes the first form's way
  Buf kp_hist, kp_recs, kp_misc;          // K5 second form (fpx_epaxos_kp.hpp)
  Buf dg_msg, dg_direct, dg_clo, dg_pre, dg_tmax, dg_pairs, dg_pairs2, dg_ctl, dg_key;  // device dependency-graph execution
  Buf dgh_in, dgh_out;                    // fpx_epx_execute: the host arrays' stay on the device
  int32_t dg_seq = 0;
  int dg_rounds_hint = 8;                 // closure rounds the first chunk of the next fpx_epx_execute_dev enqueues (DG_ROUNDS at first)
  uint32_t* kp_flag = nullptr;            // page-locked: [0] sequence number of the tick whose count [1] is valid
  uint32_t* kp_flag_dev = nullptr;
  uint32_t kp_seq = 0;
  bool kp_lds_allowed = false, kp_off = false;
  uint32_t cl_run = 0;
  bool lds_allowed = false, sort_lds_allowed = false;
  int num_cus = 256;
};

namespace {

// the context's device is current inside every entry point, the caller's is restored on return (see fpx_api.hip)
struct EpxDeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit EpxDeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) switched = hipSetDevice(device) == hipSuccess;
  }
  ~EpxDeviceGuard() {
    if (switched && prev >= 0) (void)hipSetDevice(prev);
  }
};

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
  if (b.fused) {
    // keys whose commands fit the on-chip tables are scanned and decided by one workgroup each; the two kernels
    // below then only see what is left (usually nothing: they return at once)
    if (!e->lds_allowed) {  // more than the default 64 KiB of LDS needs an opt-in, per device: once per context
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_epx_key<N>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)KeyTile<N>::BYTES);
      e->lds_allowed = true;
    }
    hipLaunchKernelGGL((k_epx_key<N>), dim3(std::min(e->st.num_keys, e->num_cus)), dim3(KeyTile<N>::THREADS), KeyTile<N>::BYTES, e->stream, e->st, b);
  }
  hipLaunchKernelGGL((k_epx_scan<N>), dim3((segs + 3) / 4), dim3(256), 0, e->stream, e->st, b);
  hipLaunchKernelGGL((k_epx_decide<N>), dim3((b.m + 255) / 256), dim3(256), 0, e->stream, e->st, b);
}

template <int N>
void launch_hp(fpx_epx* e, const EpxBatch& sb, const HpBatch& hb) {
  const int segs = N * e->st.num_keys;
  hipLaunchKernelGGL((k_epx_scan<N>), dim3((segs + 3) / 4), dim3(256), 0, e->stream, e->st, sb);
  hipLaunchKernelGGL((k_hp_reply<N>), dim3((unsigned)(((long long)hb.m * N + 255) / 256)), dim3(256), 0, e->stream, e->st, hb);
}

void launch_segments(fpx_epx* e, const EpxBatch& b, const uint32_t* key_totals, int key_buckets) {
  if (key_totals)
    hipLaunchKernelGGL(k_epx_segments_from_totals, dim3(e->st.n), dim3(256), 0, e->stream, e->st, b, key_totals, key_buckets);
  else
    hipLaunchKernelGGL(k_epx_segments, dim3((e->st.n * e->st.num_keys + 255) / 256), dim3(256), 0, e->stream, e->st, b);
}

// stable LSD radix sort of nseq sequences of m pairs on the low `bits` bits of the key word; returns the buffer that
// holds the result
uint2* radix_sort_pairs(fpx_epx* e, int nseq, int m, unsigned bits, uint2* a_buf, uint2* b_buf, const int32_t* check_rank, int* rc_out,
                        const uint32_t** key_totals, int* key_buckets) {
  const int n = nseq;
  const unsigned passes = (bits + RS_MAXW - 1) / RS_MAXW, width = (bits + passes - 1) / passes;
  const unsigned B = 1u << width;
  RsArgs a;
  a.m = m, a.tiles = (m + RS_TILE - 1) / RS_TILE;
  *rc_out = grow(e, &e->tmp, ((size_t)n * a.tiles * B + (size_t)n * B) * 4);
  if (*rc_out) return nullptr;
  a.hist = (uint32_t*)e->tmp.p, a.tot = a.hist + (size_t)n * a.tiles * B;
  if (!e->sort_lds_allowed) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_rs_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)RS_SCATTER_LDS);
    e->sort_lds_allowed = true;
  }
  uint2* buf[2] = {a_buf, b_buf};
  int cur = 0;
  for (unsigned shift = 0; shift < bits; shift += width, cur ^= 1) {
    a.shift = (int)shift;
    a.width = (int)width;
    a.rank = shift == 0 ? check_rank : nullptr;
    a.status = e->st.status;
    a.src = buf[cur], a.dst = buf[cur ^ 1];
    const dim3 tg(a.tiles, n);
    hipLaunchKernelGGL(k_rs_hist, tg, dim3(256), 0, e->stream, a);
    hipLaunchKernelGGL(k_rs_scan, dim3((B + 63) / 64, n), dim3(256), 0, e->stream, a);
    hipLaunchKernelGGL(k_rs_scatter, tg, dim3(64 * RS_SW), RS_SCATTER_LDS, e->stream, a);
  }
  // one pass: the digit was the whole key, its totals are the sizes of the key segments
  if (key_totals) *key_totals = passes == 1 ? a.tot : nullptr;
  if (key_buckets) *key_buckets = (int)B;
  return buf[cur];  // where the last pass left the sequence
}

// the n replicas' sequences of a tick on the key bits
uint2* sort_by_key(fpx_epx* e, int m, uint2* a_buf, uint2* b_buf, const int32_t* check_rank, int* rc_out,
                   const uint32_t** key_totals, int* key_buckets) {
  unsigned bits = 1;
  while ((1u << bits) <= (unsigned)e->st.num_keys) ++bits;
  return radix_sort_pairs(e, e->st.n, m, bits, a_buf, b_buf, check_rank, rc_out, key_totals, key_buckets);
}

// K5, second form: returns FPX_OK with *done = true when the tick went through it, *done = false when the first
// form has to take the tick (a key with more commands than the on-chip tables hold; nothing was applied)
template <int N>
int launch_kp(fpx_epx* e, const EpxBatch& b, int32_t* d_packed, bool* done) {
  using T = KpTile<N>;
  *done = false;
  KpArgs a;
  memset(&a, 0, sizeof(a));
  a.m = b.m, a.tiles = (b.m + KP_TILE - 1) / KP_TILE, a.B = e->st.num_keys;
  a.groups = (a.tiles + KP_HG - 1) / KP_HG;
  a.seq = ++e->kp_seq;
  int rc;
  if ((rc = grow(e, &e->kp_hist, (size_t)a.tiles * a.B * 2))) return rc;
  if ((rc = grow(e, &e->kp_recs, (size_t)a.B * T::TC * T::NI * 4))) return rc;
  {
    // fingerprints, control and verdict words, one word per group of tiles (m < 2^21: at most 128 groups), then the claim
    // counters (one 64-byte sector per key): they must read zero before the first tick, later the key kernel leaves
    // them at zero
    const size_t need = 1024 + (size_t)a.B * KP_TOT_STRIDE * 4;
    if (need > e->kp_misc.cap) {
      if ((rc = grow(e, &e->kp_misc, need))) return rc;
      EHIP(e, hipMemsetAsync(e->kp_misc.p, 0, e->kp_misc.cap, e->stream));
    }
  }
  char* misc = (char*)e->kp_misc.p;
  a.hist = (uint16_t*)e->kp_hist.p, a.recs = (uint32_t*)e->kp_recs.p;
  a.fp = (unsigned long long*)misc;              // 16 words of 8 bytes
  a.ctl = (uint32_t*)(misc + 128);               // 1 word
  a.bad = (uint32_t*)(misc + 192);               // 2 words
  a.big = (uint32_t*)(misc + 512);               // [groups <= 128]
  a.tot = (uint32_t*)(misc + 1024);
  a.host_flag = e->kp_flag_dev, a.tc = T::TC;
  a.packed = d_packed, a.stride = fpx_epx_packed_stride(N);
  {
    uintptr_t bits = (uintptr_t)b.leader | (uintptr_t)b.key | (uintptr_t)b.number | (uintptr_t)b.rank | (uintptr_t)b.resp_mask |
                     (uintptr_t)b.is_set | (uintptr_t)b.seen_mask;
    a.vec = (bits & 15u) == 0 && (b.m & 3) == 0;
  }
  if (!e->kp_lds_allowed) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_epx_key2<N, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)T::BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_epx_key2<N, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)T::BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_kp_hist), hipFuncAttributeMaxDynamicSharedMemorySize,
                              KP_HG * KP_MAXB * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_kp_scatter<N>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)KpScat<N>::BYTES);
    e->kp_lds_allowed = true;
  }
  // (Round 4 tried the partition of tick t + 1 on a stream of its own beside the key kernel of tick t: the key kernel's
  // 15 wavefronts of 128 registers fill the register files of three of a CU's four SIMDs, no partition workgroup finds
  // room beside it, and the two event hops between the streams cost 15 us per tick -- 0.125 ms against 0.119 in order.)
  hipStream_t ps = e->stream;
  hipLaunchKernelGGL(k_kp_hist, dim3(a.groups), dim3(128 * KP_HG), (size_t)KP_HG * a.B * 4, ps, e->st, b, a);
  const int units = (a.tiles + KpScat<N>::TILES - 1) / KpScat<N>::TILES;
  hipLaunchKernelGGL((k_kp_scatter<N>), dim3(8 * ((units + 7) / 8)), dim3(KpScat<N>::THREADS), KpScat<N>::BYTES, ps, e->st, b, a);
  // k_epx_key2 is enqueued at once -- it returns at its first instruction when a key does not fit -- and the host then
  // learns, while the GPU works on, whether the first form has to take the tick after all (launching the kernel
  // only after the answer left the GPU idle for ~30 us per tick when ticks were enqueued back to back)
  const int grid = std::min(e->st.num_keys, e->num_cus);
  if (e->st.num_instances > 0)
    hipLaunchKernelGGL((k_epx_key2<N, true>), dim3(grid), dim3(T::THREADS), T::BYTES, e->stream, e->st, b, a);
  else
    hipLaunchKernelGGL((k_epx_key2<N, false>), dim3(grid), dim3(T::THREADS), T::BYTES, e->stream, e->st, b, a);
  volatile uint32_t* flag = e->kp_flag;
  bool seen = false;
  for (long spin = 0; spin < 200000000L; ++spin) {
    if (flag[0] == a.seq) { seen = true; break; }
    if ((spin & 0xffff) == 0xffff && hipStreamQuery(ps) != hipErrorNotReady) break;  // finished, or failed
  }
  if (!seen) {
    EHIP(e, hipStreamSynchronize(ps));
    if (flag[0] != a.seq) return FPX_EHIP;
  }
  if (flag[1] != 0) return FPX_OK;  // a hot key: the first form takes the whole tick
  *done = true;
  return FPX_OK;
}

// bits of the closure hash that groups the cyclic vertices of one sort key into components.  FPX_DG_HASH_BITS (2 .. 22) is a
// test hook: with few bits different closures of one key share a hash, which is exactly the case needs_host_path reports
int dg_hash_bits() {
  const char* s = getenv("FPX_DG_HASH_BITS");
  const int b = s ? atoi(s) : DG_HASH_BITS;
  return b < 2 ? 2 : b > DG_HASH_BITS ? DG_HASH_BITS : b;
}

// the packed path of the device dependency graph (fpx_depgraph_pk.hpp): n <= 5, columns of fewer than 2^21 - 2 instances
template <int N>
int dg_execute_packed(fpx_epx* e, int m, const int32_t* d_leader, const int32_t* d_number, const int32_t* d_packed, const uint8_t* d_mask,
                      const int32_t* first, const int32_t* count, int32_t* d_order, int32_t* d_comp, int64_t* nexec, int64_t* ncomp,
                      int32_t* needs_host) {
  static_assert(N <= 5, "five 21-bit watermarks per 16-byte row");
  DpArgs a;
  memset(&a, 0, sizeof(a));
  a.m = m, a.n = N, a.stride = fpx_epx_packed_stride(N);
  long long total = 0;
  for (int l = 0; l < N; ++l) {
    a.first[l] = first[l], a.count[l] = count[l], a.base[l] = (int32_t)total;
    a.nblk[l] = (count[l] + 255) / 256, a.blk_base[l] = a.nblocks;
    a.nblocks += a.nblk[l], total += count[l];
  }
  for (int l = N; l < 8; ++l) a.base[l] = (int32_t)total, a.blk_base[l] = a.nblocks;
  int rc;
  const int out_tiles = (m + DG_TILE - 1) / DG_TILE;
  const size_t nb = (size_t)std::max(a.nblocks, 1);
  if ((rc = grow(e, &e->dg_msg, (size_t)m * 4))) return rc;
  if ((rc = grow(e, &e->dg_direct, (size_t)m * 16))) return rc;
  if ((rc = grow(e, &e->dg_clo, (size_t)m * 16))) return rc;
  if ((rc = grow(e, &e->dg_pre, (size_t)m * 32))) return rc;
  if ((rc = grow(e, &e->dg_tmax, nb * 16 * 4 + (size_t)out_tiles * 4 + (size_t)((m + 255) / 256) * 4 + 64))) return rc;
  if ((rc = grow(e, &e->dg_pairs, (size_t)m * 8))) return rc;
  if ((rc = grow(e, &e->dg_pairs2, (size_t)m * 8))) return rc;
  if ((rc = grow(e, &e->dg_key, (size_t)m * 4))) return rc;
  if ((rc = grow(e, &e->dg_ctl, 256))) return rc;
  if (!e->kp_flag_dev) return FPX_EHIP;
  a.leader = d_leader, a.number = d_number, a.packed = d_packed, a.mask = d_mask;
  a.msg_of = (int32_t*)e->dg_msg.p, a.direct = (ulonglong2*)e->dg_direct.p, a.clo = (ulonglong2*)e->dg_clo.p;
  a.lp[0] = (ulonglong2*)e->dg_pre.p, a.lp[1] = a.lp[0] + m;
  a.bt[0] = (ulonglong2*)e->dg_tmax.p, a.bt[1] = a.bt[0] + nb, a.cy[0] = a.bt[1] + nb, a.cy[1] = a.cy[0] + nb;
  a.tstarts = (int32_t*)(a.cy[1] + nb);
  a.belig = a.tstarts + out_tiles;
  a.pairs = (uint2*)e->dg_pairs.p, a.pairs2 = (uint2*)e->dg_pairs2.p, a.ctl = (int32_t*)e->dg_ctl.p;
  a.key32 = (uint32_t*)e->dg_key.p;
  a.host = reinterpret_cast<volatile int32_t*>(e->kp_flag_dev + 8);
  a.order = d_order, a.comp = d_comp;
  volatile int32_t* host = reinterpret_cast<volatile int32_t*>(e->kp_flag + 8);
  const int call = (++e->dg_seq) & 0xffff;
  a.count_moved = getenv("FPX_DG_DEBUG") ? 1 : 0;
  a.hash_bits = dg_hash_bits();
  auto wait_for = [&](int round) -> int {
    const int32_t want = call * 64 + round;
    for (long spin = 0; spin < 400000000L; ++spin) {
      if (host[7] == want) return FPX_OK;
      if ((spin & 0xfffff) == 0xfffff) {
        const hipError_t q = hipStreamQuery(e->stream);
        (void)hipGetLastError();
        if (q != hipErrorNotReady) break;
      }
    }
    EHIP(e, hipStreamSynchronize(e->stream));
    return host[7] == want ? FPX_OK : FPX_EHIP;
  };
  EHIP(e, hipMemsetAsync(a.msg_of, 0xFF, (size_t)m * 4, e->stream));
  EHIP(e, hipMemsetAsync(a.ctl, 0, 256, e->stream));
  const int grid = (m + 255) / 256;
  hipLaunchKernelGGL((k_dp_scatter<N>), dim3(grid), dim3(256), 0, e->stream, a);
  hipLaunchKernelGGL((k_dp_scan0<N>), dim3(dp_grid(a.nblocks)), dim3(256), 0, e->stream, a);
  hipLaunchKernelGGL((k_dp_carry<N>), dim3(N), dim3(1024), 0, e->stream, a, 0, 1);
  int cur = 0, executables = 0;
  for (int chunk = 0;; ++chunk) {
    if (chunk > 0) EHIP(e, hipMemsetAsync(a.ctl + 8, 0, (DG_ROUNDS + 1) * 4, e->stream));
    // How many rounds to enqueue: a round that finds "the one before moved nothing" leaves at its first instruction, but it
    // is still a launch (and its carry kernel another): ~8 us per skipped round, four of them per FIFO tick.  The first
    // chunk enqueues one round more than moved something in the context's previous call (ticks of one deployment need the
    // same depth, 4 with FIFO channels, 7 - 8 with reordering ones); a tick that needs more gets full chunks as before.
    const int rounds = chunk == 0 ? std::max(2, std::min(DG_ROUNDS, e->dg_rounds_hint)) : DG_ROUNDS;
    for (int k = 1; k <= rounds; ++k) {
      // round k gathers from half `cur` and leaves its scan in the other half.  (A skipped round does not flip anything on
      // the device, but after the round that moved nothing both halves hold the same, final values.)
      hipLaunchKernelGGL((k_dp_relax<N>), dim3(dp_grid(a.nblocks)), dim3(256), 0, e->stream, a, cur, k);
      hipLaunchKernelGGL((k_dp_carry<N>), dim3(N), dim3(1024), 0, e->stream, a, cur ^ 1, k);
      cur ^= 1;
    }
    // (ctl[3], the executables, is WRITTEN by the rekey kernel's first workgroup since round 6 and ctl[4], the components,
    // by the emit kernel's last: nothing accumulates in them any more, so nothing has to be cleared in front of the keys)
    hipLaunchKernelGGL((k_dp_keys<N>), dim3(dp_grid(grid)), dim3(256), 0, e->stream, a);
    uint2* sorted = radix_sort_pairs(e, 1, m, (unsigned)a.hash_bits, a.pairs, a.pairs2, nullptr, &rc, nullptr, nullptr);
    if (rc) return rc;
    if (sorted != a.pairs) std::swap(a.pairs, a.pairs2);
    hipLaunchKernelGGL(k_dp_rekey, dim3(grid), dim3(256), 0, e->stream, a);
    unsigned key_bits = 2;
    while (((1ull << key_bits) - 1) <= 3ull * (unsigned long long)m + 2) ++key_bits;
    sorted = radix_sort_pairs(e, 1, m, key_bits, a.pairs, a.pairs2, nullptr, &rc, nullptr, nullptr);
    if (rc) return rc;
    if (sorted != a.pairs) std::swap(a.pairs, a.pairs2);
    hipLaunchKernelGGL(k_dp_count_starts, dim3(out_tiles), dim3(256), 0, e->stream, a);
    hipLaunchKernelGGL(k_dp_emit, dim3(out_tiles), dim3(256), 0, e->stream, a);
    a.seq = call * 64 + (chunk & 31) + 1;
    hipLaunchKernelGGL(k_dp_publish, dim3(1), dim3(64), 0, e->stream, a, rounds);
    if ((rc = wait_for((chunk & 31) + 1))) return rc;
    if (host[1] != 0) return FPX_EINVAL;
    e->dg_rounds_hint = host[5] != 0 ? DG_ROUNDS : host[6] + 1;
    if (a.count_moved) {
      int32_t dbg[64];
      EHIP(e, hipMemcpy(dbg, a.ctl, sizeof(dbg), hipMemcpyDeviceToHost));
      fprintf(stderr, "libfpx: depgraph (packed) chunk %d, vertices moved per round:", chunk);
      for (int k = 1; k <= DG_ROUNDS; ++k) fprintf(stderr, " %d", dbg[24 + k]);
      fprintf(stderr, "\n");
    }
    executables = host[3];
    if (host[5] == 0) break;
    if (chunk >= 6) return FPX_EHIP;
  }
  if (nexec) *nexec = executables;
  if (ncomp) *ncomp = executables > 0 ? host[4] : 0;
  if (needs_host) *needs_host = host[2];
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    e->last_hip = (int)le;
    return FPX_EHIP;
  }
  return FPX_OK;
}

// device dependency-graph execution of one tick's commits (fpx_depgraph_dev.hpp)
template <int N>
int dg_execute(fpx_epx* e, int m, const int32_t* d_leader, const int32_t* d_number, const int32_t* d_packed, const uint8_t* d_mask,
               const int32_t* first, const int32_t* count, int32_t* d_order, int32_t* d_comp, int64_t* nexec, int64_t* ncomp,
               int32_t* needs_host) {
  constexpr int NP = DgRow<N>::NP;
  DgArgs a;
  memset(&a, 0, sizeof(a));
  a.m = m, a.n = N, a.stride = fpx_epx_packed_stride(N);
  long long total = 0;
  for (int l = 0; l < N; ++l) {
    if (first[l] < 0 || count[l] < 0) return FPX_EINVAL;
    a.first[l] = first[l], a.count[l] = count[l], a.base[l] = (int32_t)total;
    a.tiles[l] = (count[l] + DG_TILE - 1) / DG_TILE, a.tile_base[l] = a.ntiles;
    a.ntiles += a.tiles[l], total += count[l];
  }
  for (int l = N; l < 8; ++l) a.base[l] = (int32_t)total, a.tile_base[l] = a.ntiles;
  if (total != m) return FPX_EINVAL;  // the columns are dense: every instance first[l] .. first[l] + count[l] - 1, once
  if constexpr (N <= 5) {
    bool fits = !getenv("FPX_DG_WIDE");
    for (int l = 0; l < N; ++l) fits = fits && count[l] <= PK_MAX_COUNT;
    if (fits) return dg_execute_packed<N>(e, m, d_leader, d_number, d_packed, d_mask, first, count, d_order, d_comp, nexec, ncomp, needs_host);
  }
  int rc;
  const int out_tiles = (m + DG_TILE - 1) / DG_TILE;
  if ((rc = grow(e, &e->dg_msg, (size_t)m * 4))) return rc;
  if ((rc = grow(e, &e->dg_direct, (size_t)m * NP * 4))) return rc;
  if ((rc = grow(e, &e->dg_clo, (size_t)m * NP * 4))) return rc;
  if ((rc = grow(e, &e->dg_pre, (size_t)m * NP * 4))) return rc;
  if ((rc = grow(e, &e->dg_tmax, ((size_t)DG_SUB * a.ntiles * NP + out_tiles + (size_t)((m + 255) / 256)) * 4 + 64))) return rc;
  if ((rc = grow(e, &e->dg_pairs, (size_t)m * 8))) return rc;
  if ((rc = grow(e, &e->dg_pairs2, (size_t)m * 8))) return rc;
  if ((rc = grow(e, &e->dg_key, (size_t)m * 4))) return rc;
  if ((rc = grow(e, &e->dg_ctl, 256))) return rc;
  if (!e->kp_flag_dev) return FPX_EHIP;
  a.leader = d_leader, a.number = d_number, a.packed = d_packed, a.mask = d_mask;
  a.msg_of = (int32_t*)e->dg_msg.p, a.direct = (int32_t*)e->dg_direct.p, a.clo = (int32_t*)e->dg_clo.p, a.pre = (int32_t*)e->dg_pre.p;
  a.tmax = (int32_t*)e->dg_tmax.p, a.tstarts = a.tmax + (size_t)DG_SUB * a.ntiles * NP;
  a.belig = a.tstarts + out_tiles;
  a.pairs = (uint2*)e->dg_pairs.p, a.pairs2 = (uint2*)e->dg_pairs2.p, a.ctl = (int32_t*)e->dg_ctl.p;
  a.key32 = (uint32_t*)e->dg_key.p;
  a.host = reinterpret_cast<volatile int32_t*>(e->kp_flag_dev + 8);  // the second half of the page-locked line
  a.order = d_order, a.comp = d_comp;
  volatile int32_t* host = reinterpret_cast<volatile int32_t*>(e->kp_flag + 8);
  const int call = (++e->dg_seq) & 0xffff;
  a.count_moved = getenv("FPX_DG_DEBUG") ? 1 : 0;
  a.hash_bits = dg_hash_bits();
  auto wait_for = [&](int round) -> int {
    const int32_t want = call * 64 + round;
    for (long spin = 0; spin < 400000000L; ++spin) {
      if (host[7] == want) return FPX_OK;
      if ((spin & 0xfffff) == 0xfffff) {  // now and then: did the stream die?  (hipErrorNotReady is what a live one answers;
        const hipError_t q = hipStreamQuery(e->stream);  // it must not stay behind as the thread's last error)
        (void)hipGetLastError();
        if (q != hipErrorNotReady) break;
      }
    }
    EHIP(e, hipStreamSynchronize(e->stream));
    return host[7] == want ? FPX_OK : FPX_EHIP;
  };
  EHIP(e, hipMemsetAsync(a.msg_of, 0xFF, (size_t)m * 4, e->stream));
  EHIP(e, hipMemsetAsync(a.ctl, 0, 256, e->stream));
  const int grid = (m + 255) / 256;
  hipLaunchKernelGGL((k_dg_scatter<N>), dim3(grid), dim3(256), 0, e->stream, a);
  // round 1 scans from the direct covers' tile maxima; every later round from what the gather before it folded
  hipLaunchKernelGGL((k_dg_tilemax<N>), dim3(a.ntiles), dim3(256), 0, e->stream, a);
  int round = 0, executables = 0;
  for (int chunk = 0;; ++chunk) {
    // DG_ROUNDS closure rounds at once: a round returns at its first instruction when the one before it moved nothing,
    // and everything behind them is enqueued right away -- one host read per chunk, and one chunk covers 2^8 hops
    if (chunk > 0) EHIP(e, hipMemsetAsync(a.ctl + 8, 0, (DG_ROUNDS + 1) * 4, e->stream));
    for (int k = 1; k <= DG_ROUNDS; ++k) {
      ++round;
      hipLaunchKernelGGL((k_dg_prefix<N>), dim3(a.ntiles), dim3(256), 0, e->stream, a, round, k);
      hipLaunchKernelGGL((k_dg_relax<N>), dim3(a.ntiles * DG_SUB), dim3(256), 0, e->stream, a, round, k);
    }
    // (ctl[3], the executables, is WRITTEN by the rekey kernel's first workgroup since round 6 and ctl[4], the components,
    // by the emit kernel's last: nothing accumulates in them any more, so nothing has to be cleared in front of the keys)
    hipLaunchKernelGGL((k_dg_keys<N>), dim3(grid), dim3(256), 0, e->stream, a);
    // least significant first: the closures' hashes, then (stable) the closure sums and kinds
    uint2* sorted = radix_sort_pairs(e, 1, m, (unsigned)a.hash_bits, a.pairs, a.pairs2, nullptr, &rc, nullptr, nullptr);
    if (rc) return rc;
    if (sorted != a.pairs) std::swap(a.pairs, a.pairs2);
    hipLaunchKernelGGL(k_dg_rekey, dim3(grid), dim3(256), 0, e->stream, a);
    unsigned key_bits = 2;  // the keys are 3 x sum + kind <= 3 m + 2; an inexecutable vertex is all ones: strictly behind them
    while (((1ull << key_bits) - 1) <= 3ull * (unsigned long long)m + 2) ++key_bits;
    sorted = radix_sort_pairs(e, 1, m, key_bits, a.pairs, a.pairs2, nullptr, &rc, nullptr, nullptr);
    if (rc) return rc;
    if (sorted != a.pairs) std::swap(a.pairs, a.pairs2);
    // (the number of executables stays on the device: both kernels cover all m positions and leave at once beyond it)
    hipLaunchKernelGGL((k_dg_count_starts<N>), dim3(out_tiles), dim3(256), 0, e->stream, a);
    hipLaunchKernelGGL((k_dg_emit<N>), dim3(out_tiles), dim3(256), 0, e->stream, a);
    a.seq = call * 64 + (chunk & 31) + 1;
    hipLaunchKernelGGL(k_dg_publish, dim3(1), dim3(64), 0, e->stream, a, chunk);
    if ((rc = wait_for((chunk & 31) + 1))) return rc;
    if (host[1] != 0) return FPX_EINVAL;        // an instance outside its column, twice, or missing
    if (a.count_moved) {
      int32_t dbg[64];
      EHIP(e, hipMemcpy(dbg, a.ctl, sizeof(dbg), hipMemcpyDeviceToHost));
      fprintf(stderr, "libfpx: depgraph chunk %d, vertices moved per round:", chunk);
      for (int k = 1; k <= DG_ROUNDS; ++k) fprintf(stderr, " %d", dbg[24 + k]);
      fprintf(stderr, "\n");
    }
    executables = host[3];
    if (host[5] == 0) break;                    // the chunk's last round moved nothing (or never ran): converged
    if (round >= 48) return FPX_EHIP;           // (a round doubles the hops covered: 2^48 hops do not exist)
  }
  if (nexec) *nexec = executables;
  if (ncomp) *ncomp = executables > 0 ? host[4] : 0;
  if (needs_host) *needs_host = host[2];
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    e->last_hip = (int)le;
    return FPX_EHIP;
  }
  return FPX_OK;
}

}  // namespace

extern "C" {

int32_t fpx_epx_create(const fpx_epx_config* cfg, fpx_epx** out) {
  if (!cfg || !out) return FPX_EINVAL;
  *out = nullptr;
  const int n = cfg->num_replicas;
  if (!(n == 3 || n == 5 || n == 7) || cfg->num_keys < 1 || cfg->num_keys > (1 << 24) || cfg->num_instances < 0 ||
      (int64_t)cfg->num_instances * n > (int64_t)1 << 30)
    return FPX_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev)
    return FPX_ENODEVICE;
  fpx_epx* e = new (std::nothrow) fpx_epx();
  if (!e) return FPX_ENOMEM;
  e->cfg = *cfg;
  e->st.n = n;
  e->st.num_keys = cfg->num_keys;
  e->st.gets = e->st.sets = e->st.status = nullptr;
  e->st.num_instances = cfg->num_instances;
  e->st.cl_status = nullptr, e->st.cl_ballot = e->st.cl_vote = e->st.cl_triple = e->st.largest = nullptr, e->st.cl_stamp = nullptr;
  e->st.cl_deps = e->st.cl_dend = nullptr;
  auto fail = [&](int code) {
    fpx_epx_destroy(e);
    return code;
  };
  EpxDeviceGuard _dg(cfg->device);
  if (hipSetDevice(cfg->device) != hipSuccess) return fail(FPX_ENODEVICE);
  if (hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) != hipSuccess) return fail(FPX_EHIP);
  e->stream = e->own_stream;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0)
      e->num_cus = prop.multiProcessorCount;
  }
  const size_t cells = (size_t)n * cfg->num_keys * n;
  if (hipMalloc((void**)&e->st.gets, cells * 4) != hipSuccess) return fail(FPX_ENOMEM);
  if (hipMalloc((void**)&e->st.sets, cells * 4) != hipSuccess) return fail(FPX_ENOMEM);
  if (hipMalloc((void**)&e->st.status, 32) != hipSuccess) return fail(FPX_ENOMEM);
  // TopOne.scala:10: every watermark starts at 0
  if (hipMemsetAsync(e->st.gets, 0, cells * 4, e->stream) != hipSuccess) return fail(FPX_EHIP);
  if (hipMemsetAsync(e->st.sets, 0, cells * 4, e->stream) != hipSuccess) return fail(FPX_EHIP);
  if (hipMemsetAsync(e->st.status, 0, 32, e->stream) != hipSuccess) return fail(FPX_EHIP);
  {
    // Replica.scala:458  largestBallot = Ballot(0, index)  (encoded 0 * 8 + index)
    int32_t init[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    if (hipMalloc((void**)&e->st.largest, 32) != hipSuccess) return fail(FPX_ENOMEM);
    if (hipMemcpy(e->st.largest, init, 32, hipMemcpyHostToDevice) != hipSuccess) return fail(FPX_EHIP);
  }
  if (cfg->num_instances > 0) {
    const size_t ce = (size_t)n * n * cfg->num_instances;
    if (hipMalloc((void**)&e->st.cl_status, ce) != hipSuccess) return fail(FPX_ENOMEM);
    if (hipMalloc((void**)&e->st.cl_ballot, ce * 4) != hipSuccess) return fail(FPX_ENOMEM);
    if (hipMalloc((void**)&e->st.cl_vote, ce * 4) != hipSuccess) return fail(FPX_ENOMEM);
    if (hipMalloc((void**)&e->st.cl_triple, ce * 4) != hipSuccess) return fail(FPX_ENOMEM);
    if (hipMalloc((void**)&e->st.cl_stamp, (size_t)n * cfg->num_instances * 4) != hipSuccess) return fail(FPX_ENOMEM);
    if (hipMalloc((void**)&e->st.cl_deps, ce * n * 4) != hipSuccess) return fail(FPX_ENOMEM);
    if (hipMalloc((void**)&e->st.cl_dend, ce * 4) != hipSuccess) return fail(FPX_ENOMEM);
    if (hipMemsetAsync(e->st.cl_deps, 0, ce * n * 4, e->stream) != hipSuccess) return fail(FPX_EHIP);
    if (hipMemsetAsync(e->st.cl_dend, 0, ce * 4, e->stream) != hipSuccess) return fail(FPX_EHIP);
    if (hipMemsetAsync(e->st.cl_status, 0, ce, e->stream) != hipSuccess) return fail(FPX_EHIP);
    if (hipMemsetAsync(e->st.cl_ballot, 0xFF, ce * 4, e->stream) != hipSuccess) return fail(FPX_EHIP);
    if (hipMemsetAsync(e->st.cl_vote, 0xFF, ce * 4, e->stream) != hipSuccess) return fail(FPX_EHIP);
    if (hipMemsetAsync(e->st.cl_triple, 0xFF, ce * 4, e->stream) != hipSuccess) return fail(FPX_EHIP);
    if (hipMemsetAsync(e->st.cl_stamp, 0, (size_t)n * cfg->num_instances * 4, e->stream) != hipSuccess) return fail(FPX_EHIP);
  }
  // the word k_kp_scatter tells the host through (did a key of the tick outgrow the on-chip tables?)
  if (hipHostMalloc((void**)&e->kp_flag, 64, hipHostMallocDefault) == hipSuccess) {
    memset(e->kp_flag, 0, 64);
    if (hipHostGetDevicePointer((void**)&e->kp_flag_dev, e->kp_flag, 0) != hipSuccess) e->kp_flag_dev = nullptr;
  } else {
    e->kp_flag = nullptr;
    (void)hipGetLastError();
  }
  e->kp_off = getenv("FPX_EPX_V1") != nullptr;
  if (hipStreamSynchronize(e->stream) != hipSuccess) return fail(FPX_EHIP);
  *out = e;
  return FPX_OK;
}

int32_t fpx_epx_destroy(fpx_epx* e) {
  if (!e) return FPX_EINVAL;
  EpxDeviceGuard _dg(e->cfg.device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  void* ps[] = {e->st.gets, e->st.sets, e->st.status, e->st.cl_status, e->st.cl_ballot, e->st.cl_vote, e->st.cl_triple,
                e->st.largest, e->st.cl_stamp, e->st.cl_deps, e->st.cl_dend};
  for (void* p : ps)
    if (p) (void)hipFree(p);
  Buf* bs[] = {&e->kv, &e->kv2, &e->seg, &e->conf, &e->tmp, &e->tick, &e->h_leader, &e->h_number,
               &e->h_key, &e->h_set, &e->h_mask, &e->h_seen, &e->h_rank, &e->h_triple, &e->o_fast, &e->o_deps, &e->o_ldeps,
               &e->o_own, &e->cl, &e->hp, &e->fusedb, &e->metab};
  for (Buf* b : bs)
    if (b->p) (void)hipFree(b->p);
  for (Buf* b : {&e->kp_hist, &e->kp_recs, &e->kp_misc, &e->p_fast, &e->p_deps, &e->p_ldeps, &e->p_own, &e->dg_msg, &e->dg_direct,
                 &e->dg_clo, &e->dg_pre, &e->dg_tmax, &e->dg_pairs, &e->dg_pairs2, &e->dg_ctl, &e->dg_key, &e->dgh_in, &e->dgh_out})
    if (b->p) (void)hipFree(b->p);
  if (e->kp_flag) (void)hipHostFree(e->kp_flag);
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  delete e;
  return FPX_OK;
}

int32_t fpx_epx_set_stream(fpx_epx* e, void* hip_stream) {
  if (!e) return FPX_EINVAL;
  EpxDeviceGuard _dg(e->cfg.device);
  EHIP(e, hipStreamSynchronize(e->stream));
  e->stream = hip_stream == FPX_STREAM_OWN ? e->own_stream : (hipStream_t)hip_stream;
  return FPX_OK;
}

int32_t fpx_epx_sync(fpx_epx* e) {
  if (!e) return FPX_EINVAL;
  EpxDeviceGuard _dg(e->cfg.device);
  int32_t h[2] = {0, 0};
  EHIP(e, hipMemcpyAsync(h, e->st.status, sizeof(h), hipMemcpyDeviceToHost, e->stream));
  EHIP(e, hipStreamSynchronize(e->stream));
  if (h[0] != 0) {
    EHIP(e, hipMemsetAsync(e->st.status, 0, 32, e->stream));
    EHIP(e, hipStreamSynchronize(e->stream));
  }
  return h[0];
}

static int32_t preaccept_dev_impl(fpx_epx* e, int32_t m, const int32_t* d_leader, const int32_t* d_number,
                                  const int32_t* d_key, const uint8_t* d_is_set, const uint8_t* d_resp_mask,
                                  const uint8_t* d_seen_mask, const int32_t* d_rank, const int32_t* d_triple_id,
                                  uint8_t* d_fast, int32_t* d_deps, int32_t* d_leader_deps, int32_t* d_own_values_end,
                                  int32_t* d_packed) {
  if (!e || m < 0) return FPX_EINVAL;
  EpxDeviceGuard _dg(e->cfg.device);
  if (m == 0) return FPX_OK;
  const int n = e->st.n;
  int rc;
  // the second form (fpx_epaxos_kp.hpp): one partition pass by key, everything else on chip -- when the keys are
  // one LDS counter each and ranks and slots share a 32-bit sort word
  const int kp_tc = n == 3 ? KpTile<3>::TC : n == 5 ? KpTile<5>::TC : KpTile<7>::TC;
  if (!e->kp_off && e->kp_flag_dev && e->st.num_keys <= KP_MAXB && m < (1 << 21) &&
      (long long)m <= (long long)e->st.num_keys * kp_tc &&  // (else some key must overflow the on-chip tables)
      ((e->st.num_instances == 0 && !d_triple_id) || n >= 5)) {  // (n = 3 with a command log: the first form)
    EpxBatch kb;
    memset(&kb, 0, sizeof(kb));
    kb.m = m, kb.leader = d_leader, kb.number = d_number, kb.key = d_key, kb.is_set = d_is_set, kb.resp_mask = d_resp_mask;
    kb.seen_mask = d_seen_mask, kb.rank = d_rank, kb.triple = d_triple_id;
    kb.fast = d_fast, kb.deps = d_deps, kb.leader_deps = d_leader_deps, kb.own_values_end = d_own_values_end;
    bool done = false;
    switch (n) {
      case 3: rc = launch_kp<3>(e, kb, d_packed, &done); break;
      case 5: rc = launch_kp<5>(e, kb, d_packed, &done); break;
      default: rc = launch_kp<7>(e, kb, d_packed, &done); break;
    }
    if (rc) return rc;
    if (done) {
      hipError_t le = hipGetLastError();
      if (le != hipSuccess) {
        e->last_hip = (int)le;
        return FPX_EHIP;
      }
      return FPX_OK;
    }
  }
  if (d_packed) {  // the first form writes the four arrays: into the context's own, packed at the end
    if ((rc = grow(e, &e->p_fast, (size_t)m))) return rc;
    if ((rc = grow(e, &e->p_deps, (size_t)m * n * 4))) return rc;
    if ((rc = grow(e, &e->p_ldeps, (size_t)m * n * 4))) return rc;
    if ((rc = grow(e, &e->p_own, (size_t)m * 8))) return rc;
    d_fast = (uint8_t*)e->p_fast.p, d_deps = (int32_t*)e->p_deps.p, d_leader_deps = (int32_t*)e->p_ldeps.p;
    d_own_values_end = (int32_t*)e->p_own.p;
  }
  if ((rc = grow(e, &e->kv, (size_t)n * m * 8))) return rc;
  if ((rc = grow(e, &e->kv2, (size_t)n * m * 8))) return rc;
  if ((rc = grow(e, &e->tick, (size_t)n * e->st.num_keys * 2 * n * 4))) return rc;
  if ((rc = grow(e, &e->seg, (size_t)n * e->st.num_keys * 8))) return rc;
  if ((rc = grow(e, &e->conf, (size_t)m * n * (n <= 4 ? 4 : 8) * 4))) return rc;
  EpxBatch b;
  memset(&b, 0, sizeof(b));
  b.m = m, b.leader = d_leader, b.number = d_number, b.key = d_key, b.is_set = d_is_set, b.resp_mask = d_resp_mask;
  b.seen_mask = d_seen_mask;
  b.rank = d_rank;
  b.triple = d_triple_id;
  b.kv = (uint2*)e->kv.p, b.kv_sorted = (uint2*)e->kv2.p;
  b.tick = (int32_t*)e->tick.p;
  b.seg = (int32_t*)e->seg.p, b.conf = (int32_t*)e->conf.p;
  b.fast = d_fast, b.deps = d_deps, b.leader_deps = d_leader_deps, b.own_values_end = d_own_values_end;
  // the per-key on-chip path (k_epx_key): message indices must fit its 21-bit hash words, one workgroup per key
  if (m < (1 << 21) - 1 && e->st.num_keys <= (1 << 16) && !getenv("FPX_EPX_NO_KEY_TILES")) {
    if ((rc = grow(e, &e->fusedb, (size_t)e->st.num_keys + 64))) return rc;
    b.unfused = (int32_t*)e->fusedb.p;
    b.fused = (uint8_t*)e->fusedb.p + 64;
    if ((rc = grow(e, &e->metab, (size_t)m * 8))) return rc;
    b.meta = (int2*)e->metab.p;
  }
  // (counting the first pass's histogram in k_epx_keys with global atomics was tried: 5 M atomics on 327 k counters
  // took 450 us against 22 us for the histogram kernel)
  hipLaunchKernelGGL(k_epx_keys, dim3((m + 255) / 256), dim3(256), 0, e->stream, e->st, b);
  // stable LSD radix sort on the key bits only (the sequence already is in delivery order)
  const uint32_t* key_totals = nullptr;
  int key_buckets = 0;
  b.kv_sorted = sort_by_key(e, m, b.kv, b.kv_sorted, d_rank, &rc, &key_totals, &key_buckets);
  if (rc) return rc;
  launch_segments(e, b, key_totals, key_buckets);
  switch (n) {
    case 3: launch_scan_decide<3>(e, b); break;
    case 5: launch_scan_decide<5>(e, b); break;
    default: launch_scan_decide<7>(e, b); break;
  }
  const long long tot = (long long)e->st.num_keys * n * n;
  hipLaunchKernelGGL(k_epx_commit, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, e->stream, e->st, b);
  if (d_packed)
    hipLaunchKernelGGL(k_epx_pack, dim3((m + 255) / 256), dim3(256), 0, e->stream, e->st, m, d_fast, d_deps, d_leader_deps,
                       d_own_values_end, d_packed, fpx_epx_packed_stride(n));
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    e->last_hip = (int)le;
    return FPX_EHIP;
  }
  return FPX_OK;
}

int32_t fpx_epx_packed_stride(int32_t num_replicas) { return (2 * num_replicas + 3 + 3) / 4 * 4; }

int32_t fpx_epx_execute_dev(fpx_epx* e, int32_t m, const int32_t* d_leader, const int32_t* d_number, const int32_t* d_packed,
                            const uint8_t* d_committed, const int32_t* first, const int32_t* count, int32_t* d_order,
                            int32_t* d_component, int64_t* num_executed, int64_t* num_components, int32_t* needs_host_path) {
  if (!e || m < 0 || !first || !count || (m > 0 && (!d_leader || !d_number || !d_packed || !d_order || !d_component))) return FPX_EINVAL;
  EpxDeviceGuard _dg(e->cfg.device);
  if (num_executed) *num_executed = 0;
  if (num_components) *num_components = 0;
  if (needs_host_path) *needs_host_path = 0;
  if (m == 0) return FPX_OK;
  if (m >= (1 << 21)) return FPX_EINVAL;
  switch (e->st.n) {
    case 3: return dg_execute<3>(e, m, d_leader, d_number, d_packed, d_committed, first, count, d_order, d_component, num_executed, num_components, needs_host_path);
    case 5: return dg_execute<5>(e, m, d_leader, d_number, d_packed, d_committed, first, count, d_order, d_component, num_executed, num_components, needs_host_path);
    default: return dg_execute<7>(e, m, d_leader, d_number, d_packed, d_committed, first, count, d_order, d_component, num_executed, num_components, needs_host_path);
  }
}

int32_t fpx_epx_execute(fpx_epx* e, int32_t m, const int32_t* leader, const int32_t* number, const int32_t* deps,
                        const int32_t* deps_values_end, const uint8_t* committed, const int32_t* first, const int32_t* count,
                        int32_t* order, int32_t* component, int64_t* num_executed, int64_t* num_components,
                        int32_t* needs_host_path) {
  if (!e || m < 0 || !first || !count || (m > 0 && (!leader || !number || !deps || !order || !component))) return FPX_EINVAL;
  EpxDeviceGuard _dg(e->cfg.device);
  if (num_executed) *num_executed = 0;
  if (num_components) *num_components = 0;
  if (needs_host_path) *needs_host_path = 0;
  if (m == 0) return FPX_OK;
  const int n = e->st.n, stride = fpx_epx_packed_stride(n);
  // the packed lines fpx_epx_preaccept_packed_dev would have written: deps | leader_deps (unused here) | own_values_end | fast
  std::vector<int32_t> lines;
  try {
    lines.assign((size_t)m * stride, 0);
  } catch (const std::bad_alloc&) {  // (no exception crosses the C ABI)
    return FPX_ENOMEM;
  }
  for (int i = 0; i < m; ++i) {
    if (leader[i] < 0 || leader[i] >= n) return FPX_EINVAL;
    int32_t* line = lines.data() + (size_t)i * stride;
    for (int l = 0; l < n; ++l) line[l] = deps[(size_t)i * n + l];
    line[2 * n] = deps_values_end ? deps_values_end[i] : 0;
  }
  const size_t mp = ((size_t)m + 63) & ~(size_t)63;
  int rc;
  if ((rc = grow(e, &e->dgh_in, mp * 4 * 2 + mp + lines.size() * 4 + 256))) return rc;
  if ((rc = grow(e, &e->dgh_out, mp * 4 * 2))) return rc;
  char* p = (char*)e->dgh_in.p;
  int32_t *d_leader = (int32_t*)p, *d_number = (int32_t*)(p + mp * 4), *d_packed = (int32_t*)(p + mp * 8);
  uint8_t* d_mask = (uint8_t*)(p + mp * 8 + lines.size() * 4);
  int32_t *d_order = (int32_t*)e->dgh_out.p, *d_comp = d_order + mp;
  EHIP(e, hipMemcpyAsync(d_leader, leader, (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemcpyAsync(d_number, number, (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemcpyAsync(d_packed, lines.data(), lines.size() * 4, hipMemcpyHostToDevice, e->stream));
  if (committed) EHIP(e, hipMemcpyAsync(d_mask, committed, (size_t)m, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipStreamSynchronize(e->stream));  // (the host arrays and `lines` may be pageable)
  int64_t ne = 0, nc = 0;
  int32_t nh = 0;
  rc = fpx_epx_execute_dev(e, m, d_leader, d_number, d_packed, committed ? d_mask : nullptr, first, count, d_order, d_comp, &ne, &nc, &nh);
  if (rc) return rc;
  if (ne > 0 && !nh) {
    EHIP(e, hipMemcpyAsync(order, d_order, (size_t)ne * 4, hipMemcpyDeviceToHost, e->stream));
    EHIP(e, hipMemcpyAsync(component, d_comp, (size_t)ne * 4, hipMemcpyDeviceToHost, e->stream));
    EHIP(e, hipStreamSynchronize(e->stream));
  }
  if (num_executed) *num_executed = ne;
  if (num_components) *num_components = nc;
  if (needs_host_path) *needs_host_path = nh;
  return FPX_OK;
}

int32_t fpx_epx_preaccept_dev(fpx_epx* e, int32_t m, const int32_t* d_leader, const int32_t* d_number,
                              const int32_t* d_key, const uint8_t* d_is_set, const uint8_t* d_resp_mask,
                              const uint8_t* d_seen_mask, const int32_t* d_rank, const int32_t* d_triple_id,
                              uint8_t* d_fast, int32_t* d_deps, int32_t* d_leader_deps, int32_t* d_own_values_end) {
  return preaccept_dev_impl(e, m, d_leader, d_number, d_key, d_is_set, d_resp_mask, d_seen_mask, d_rank, d_triple_id, d_fast,
                            d_deps, d_leader_deps, d_own_values_end, nullptr);
}

int32_t fpx_epx_preaccept_packed_dev(fpx_epx* e, int32_t m, const int32_t* d_leader, const int32_t* d_number,
                                     const int32_t* d_key, const uint8_t* d_is_set, const uint8_t* d_resp_mask,
                                     const uint8_t* d_seen_mask, const int32_t* d_rank, const int32_t* d_triple_id,
                                     int32_t* d_packed) {
  if (!d_packed && m > 0) return FPX_EINVAL;
  return preaccept_dev_impl(e, m, d_leader, d_number, d_key, d_is_set, d_resp_mask, d_seen_mask, d_rank, d_triple_id, nullptr,
                            nullptr, nullptr, nullptr, d_packed);
}

int32_t fpx_epx_preaccept(fpx_epx* e, int32_t m, const int32_t* leader, const int32_t* number, const int32_t* key,
                          const uint8_t* is_set, const uint8_t* resp_mask, const uint8_t* seen_mask,
                          const int32_t* rank, const int32_t* triple_id, uint8_t* fast, int32_t* deps,
                          int32_t* leader_deps, int32_t* own_values_end) {
  if (!e || m < 0 || (m > 0 && (!leader || !number || !key || !is_set || !resp_mask || !rank))) return FPX_EINVAL;
  EpxDeviceGuard _dg(e->cfg.device);
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
  if (seen_mask && (rc = up(&e->h_seen, seen_mask, (size_t)m))) return rc;
  if ((rc = up(&e->h_rank, rank, (size_t)n * m * 4))) return rc;
  if (triple_id && (rc = up(&e->h_triple, triple_id, (size_t)m * 4))) return rc;
  if ((rc = grow(e, &e->o_fast, (size_t)m))) return rc;
  if ((rc = grow(e, &e->o_deps, (size_t)m * n * 4))) return rc;
  if ((rc = grow(e, &e->o_ldeps, (size_t)m * n * 4))) return rc;
  if ((rc = grow(e, &e->o_own, (size_t)m * 8))) return rc;
  rc = fpx_epx_preaccept_dev(e, m, (int32_t*)e->h_leader.p, (int32_t*)e->h_number.p, (int32_t*)e->h_key.p,
                             (uint8_t*)e->h_set.p, (uint8_t*)e->h_mask.p, seen_mask ? (uint8_t*)e->h_seen.p : nullptr,
                             (int32_t*)e->h_rank.p, triple_id ? (int32_t*)e->h_triple.p : nullptr, (uint8_t*)e->o_fast.p,
                             (int32_t*)e->o_deps.p, (int32_t*)e->o_ldeps.p, (int32_t*)e->o_own.p);
  if (rc) return rc;
  if (fast) EHIP(e, hipMemcpyAsync(fast, e->o_fast.p, (size_t)m, hipMemcpyDeviceToHost, e->stream));
  if (deps) EHIP(e, hipMemcpyAsync(deps, e->o_deps.p, (size_t)m * n * 4, hipMemcpyDeviceToHost, e->stream));
  if (leader_deps) EHIP(e, hipMemcpyAsync(leader_deps, e->o_ldeps.p, (size_t)m * n * 4, hipMemcpyDeviceToHost, e->stream));
  if (own_values_end) EHIP(e, hipMemcpyAsync(own_values_end, e->o_own.p, (size_t)m * 8, hipMemcpyDeviceToHost, e->stream));
  return fpx_epx_sync(e);
}

// Prepare / Accept on the command log: stage, validate, handle, scan the largestBallot's, (Accept) tally + commit
static int32_t cl_run(fpx_epx* e, int accept, int32_t m, const int32_t* leader, const int32_t* number,
                      const int32_t* b_ord, const int32_t* b_rep, const int32_t* triple, const int32_t* key,
                      const uint8_t* is_set, const uint8_t* target, uint8_t* ok_bits, uint8_t* nack_bits,
                      uint8_t* commit_bits, int32_t* nack_ballot, uint8_t* committed, int32_t* reply_status,
                      int32_t* reply_vote, int32_t* reply_triple) {
  if (!e || m < 0) return FPX_EINVAL;
  EpxDeviceGuard _dg(e->cfg.device);
  if (e->st.num_instances <= 0) return FPX_EINVAL;
  if (m == 0) return FPX_OK;
  if (!leader || !number || !b_ord || !b_rep || !target || (accept && (!triple || !key || !is_set))) return FPX_EINVAL;
  const int n = e->st.n;
  const int tiles = (m + CL_TILE - 1) / CL_TILE;
  const size_t mp = ((size_t)m + 63) & ~(size_t)63;
  // staging: 6 int32 inputs, is_set, target, 3 reply bit arrays, skip, committed, nack_ballot, 3 reply int arrays
  // [m][n], contrib [n][m], nackflag [n][m], tilemax [n][tiles]
  const size_t bytes = mp * 4 * 6 + mp * 7 + mp * 4 + (size_t)m * n * 4 * 3 + (size_t)n * mp * 4 + (size_t)n * mp +
                       (size_t)n * tiles * 4 + 2048;
  int rc;
  if ((rc = grow(e, &e->cl, bytes))) return rc;
  char* p = (char*)e->cl.p;
  auto take = [&](size_t sz) { char* q = p; p += (sz + 63) & ~(size_t)63; return q; };
  int32_t* d_leader = (int32_t*)take(mp * 4); int32_t* d_number = (int32_t*)take(mp * 4);
  int32_t* d_bo = (int32_t*)take(mp * 4); int32_t* d_br = (int32_t*)take(mp * 4); int32_t* d_tr = (int32_t*)take(mp * 4);
  int32_t* d_key = (int32_t*)take(mp * 4); uint8_t* d_set = (uint8_t*)take(mp);
  uint8_t* d_tgt = (uint8_t*)take(mp); uint8_t* d_ok = (uint8_t*)take(mp); uint8_t* d_nack = (uint8_t*)take(mp);
  uint8_t* d_com = (uint8_t*)take(mp); uint8_t* d_skip = (uint8_t*)take(mp); uint8_t* d_done = (uint8_t*)take(mp);
  int32_t* d_nb = (int32_t*)take(mp * 4);
  int32_t* d_rs = (int32_t*)take((size_t)m * n * 4); int32_t* d_rv = (int32_t*)take((size_t)m * n * 4);
  int32_t* d_rt = (int32_t*)take((size_t)m * n * 4);
  int32_t* d_contrib = (int32_t*)take((size_t)n * m * 4); uint8_t* d_flag = (uint8_t*)take((size_t)n * m);
  int32_t* d_tm = (int32_t*)take((size_t)n * tiles * 4);
  EHIP(e, hipMemcpyAsync(d_leader, leader, (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemcpyAsync(d_number, number, (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemcpyAsync(d_bo, b_ord, (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemcpyAsync(d_br, b_rep, (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
  if (accept) EHIP(e, hipMemcpyAsync(d_tr, triple, (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
  if (accept) EHIP(e, hipMemcpyAsync(d_key, key, (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
  if (accept) EHIP(e, hipMemcpyAsync(d_set, is_set, (size_t)m, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemcpyAsync(d_tgt, target, (size_t)m, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemsetAsync(d_ok, 0, mp * 5, e->stream));  // ok, nack, commit, skip, committed are contiguous
  EHIP(e, hipMemsetAsync(d_nb, 0xFF, mp * 4, e->stream));
  ClBatch b;
  memset(&b, 0, sizeof(b));
  b.m = m, b.accept = accept, b.leader = d_leader, b.number = d_number, b.b_ord = d_bo, b.b_rep = d_br, b.triple = d_tr;
  b.key = d_key, b.is_set = d_set;
  b.target = d_tgt, b.ok_bits = d_ok, b.nack_bits = d_nack, b.commit_bits = d_com, b.nack_ballot = d_nb;
  b.committed = d_done, b.reply_status = d_rs, b.reply_vote = d_rv, b.reply_triple = d_rt;
  b.contrib = d_contrib, b.nackflag = d_flag, b.tilemax = d_tm, b.skip = d_skip;
  if (++e->cl_run == 0) {  // stamp space exhausted: start over
    EHIP(e, hipMemsetAsync(e->st.cl_stamp, 0, (size_t)n * e->st.num_instances * 4, e->stream));
    e->cl_run = 1;
  }
  b.run_id = e->cl_run;
  const dim3 gm((m + 255) / 256), blk(256);
  hipLaunchKernelGGL(k_cl_validate, gm, blk, 0, e->stream, e->st, b);
  if (accept) hipLaunchKernelGGL(k_cl_propose, gm, blk, 0, e->stream, e->st, b);
  hipLaunchKernelGGL(k_cl_handle, dim3((unsigned)(((long long)m * n + 255) / 256)), blk, 0, e->stream, e->st, b);
  hipLaunchKernelGGL(k_cl_tilemax, dim3(tiles, n), blk, 0, e->stream, b, tiles);
  hipLaunchKernelGGL(k_cl_tilescan, dim3(n), blk, 0, e->stream, e->st, b, tiles);
  hipLaunchKernelGGL(k_cl_nacks, dim3(tiles, n), blk, 0, e->stream, b, tiles);
  if (accept) hipLaunchKernelGGL(k_cl_commit, gm, blk, 0, e->stream, e->st, b);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    e->last_hip = (int)le;
    return FPX_EHIP;
  }
  if (ok_bits) EHIP(e, hipMemcpyAsync(ok_bits, d_ok, (size_t)m, hipMemcpyDeviceToHost, e->stream));
  if (nack_bits) EHIP(e, hipMemcpyAsync(nack_bits, d_nack, (size_t)m, hipMemcpyDeviceToHost, e->stream));
  if (commit_bits) EHIP(e, hipMemcpyAsync(commit_bits, d_com, (size_t)m, hipMemcpyDeviceToHost, e->stream));
  if (nack_ballot) EHIP(e, hipMemcpyAsync(nack_ballot, d_nb, (size_t)m * 4, hipMemcpyDeviceToHost, e->stream));
  if (committed) EHIP(e, hipMemcpyAsync(committed, d_done, (size_t)m, hipMemcpyDeviceToHost, e->stream));
  if (reply_status) EHIP(e, hipMemcpyAsync(reply_status, d_rs, (size_t)m * n * 4, hipMemcpyDeviceToHost, e->stream));
  if (reply_vote) EHIP(e, hipMemcpyAsync(reply_vote, d_rv, (size_t)m * n * 4, hipMemcpyDeviceToHost, e->stream));
  if (reply_triple) EHIP(e, hipMemcpyAsync(reply_triple, d_rt, (size_t)m * n * 4, hipMemcpyDeviceToHost, e->stream));
  return fpx_epx_sync(e);
}

int32_t fpx_epx_prepare(fpx_epx* e, int32_t m, const int32_t* leader, const int32_t* number, const int32_t* ballot_ordering,
                        const int32_t* ballot_replica, const uint8_t* target_mask, uint8_t* ok_bits, uint8_t* nack_bits,
                        uint8_t* commit_bits, int32_t* nack_ballot, int32_t* reply_status, int32_t* reply_vote_ballot,
                        int32_t* reply_triple) {
  return cl_run(e, 0, m, leader, number, ballot_ordering, ballot_replica, nullptr, nullptr, nullptr, target_mask, ok_bits,
                nack_bits, commit_bits, nack_ballot, nullptr, reply_status, reply_vote_ballot, reply_triple);
}

int32_t fpx_epx_handle_commit(fpx_epx* e, int32_t m, const int32_t* leader, const int32_t* number, const int32_t* triple_id,
                              const int32_t* key, const uint8_t* is_set, const int32_t* deps, const int32_t* deps_values_end,
                              const uint8_t* target_mask) {
  if (!e || m < 0) return FPX_EINVAL;
  EpxDeviceGuard _dg(e->cfg.device);
  if (e->st.num_instances <= 0) return FPX_EINVAL;
  if (m == 0) return FPX_OK;
  if (!leader || !number || !triple_id || !key || !is_set || !target_mask) return FPX_EINVAL;
  const int n = e->st.n;
  for (int i = 0; i < m; ++i) {  // host arrays: checked here, nothing is applied on a bad one
    if (leader[i] < 0 || leader[i] >= n || number[i] < 0 || number[i] >= e->st.num_instances || key[i] < -1 || key[i] >= e->st.num_keys ||
        (target_mask[i] >> n) != 0)
      return FPX_EINVAL;
    if (deps) {
      for (int l = 0; l < n; ++l)
        if (deps[(size_t)i * n + l] < 0) return FPX_EINVAL;
      const int end = deps_values_end ? deps_values_end[i] : 0;
      if (end != 0 && (end <= number[i] + 1 || deps[(size_t)i * n + leader[i]] > number[i])) return FPX_EINVAL;
    }
  }
  const size_t mp = ((size_t)m + 63) & ~(size_t)63;
  int rc;
  // who writes the entry: walking the batch from its end, a message leaves to the later messages of its instance the
  // replicas they go to
  std::vector<uint8_t> writer;
  std::vector<std::pair<long long, int>> order;
  try {
    writer.resize((size_t)m), order.resize((size_t)m);
  } catch (const std::bad_alloc&) {  // (no exception crosses the C ABI)
    return FPX_ENOMEM;
  }
  {
    for (int i = 0; i < m; ++i) order[i] = {(long long)leader[i] * e->st.num_instances + number[i], i};
    std::sort(order.begin(), order.end());
    for (size_t a = 0; a < order.size();) {
      size_t z = a;
      while (z < order.size() && order[z].first == order[a].first) ++z;
      unsigned later = 0;
      for (size_t q = z; q-- > a;) {  // the instance's messages, last first
        const int i = order[q].second;
        writer[i] = (uint8_t)(target_mask[i] & ~later);
        later |= target_mask[i];
      }
      a = z;
    }
  }
  if ((rc = grow(e, &e->cl, mp * 4 * 5 + mp * 3 + (size_t)m * n * 4 + 1024))) return rc;
  char* p = (char*)e->cl.p;
  auto take = [&](size_t sz) { char* q = p; p += (sz + 63) & ~(size_t)63; return q; };
  int32_t *d_leader = (int32_t*)take(mp * 4), *d_number = (int32_t*)take(mp * 4), *d_tr = (int32_t*)take(mp * 4);
  int32_t *d_key = (int32_t*)take(mp * 4), *d_end = (int32_t*)take(mp * 4);
  uint8_t *d_set = (uint8_t*)take(mp), *d_tgt = (uint8_t*)take(mp), *d_wr = (uint8_t*)take(mp);
  EHIP(e, hipMemcpyAsync(d_wr, writer.data(), (size_t)m, hipMemcpyHostToDevice, e->stream));
  int32_t* d_deps = (int32_t*)take((size_t)m * n * 4);
  EHIP(e, hipMemcpyAsync(d_leader, leader, (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemcpyAsync(d_number, number, (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemcpyAsync(d_tr, triple_id, (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemcpyAsync(d_key, key, (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemcpyAsync(d_set, is_set, (size_t)m, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemcpyAsync(d_tgt, target_mask, (size_t)m, hipMemcpyHostToDevice, e->stream));
  if (deps) EHIP(e, hipMemcpyAsync(d_deps, deps, (size_t)m * n * 4, hipMemcpyHostToDevice, e->stream));
  if (deps && deps_values_end) EHIP(e, hipMemcpyAsync(d_end, deps_values_end, (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
  LcBatch b;
  b.m = m, b.leader = d_leader, b.number = d_number, b.triple = d_tr, b.key = d_key, b.is_set = d_set;
  b.deps = deps ? d_deps : nullptr, b.deps_end = (deps && deps_values_end) ? d_end : nullptr, b.target = d_tgt, b.writer = d_wr;
  hipLaunchKernelGGL(k_cl_learn_commit, dim3((unsigned)(((long long)m * n + 255) / 256)), dim3(256), 0, e->stream, e->st, b);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    e->last_hip = (int)le;
    return FPX_EHIP;
  }
  return fpx_epx_sync(e);
}

int32_t fpx_epx_handle_prepare_oks(fpx_epx* e, int32_t m, const int32_t* leader, const int32_t* number,
                                   const int32_t* ballot_ordering, const int32_t* ballot_replica, const uint8_t* resp_mask,
                                   const int32_t* reply_status, const int32_t* reply_vote_ballot, const int32_t* reply_triple,
                                   int32_t as_intended, int32_t* action, int32_t* source, int32_t* triple) {
  if (!e || m < 0) return FPX_EINVAL;
  EpxDeviceGuard _dg(e->cfg.device);
  if (e->st.num_instances <= 0) return FPX_EINVAL;
  if (m == 0) return FPX_OK;
  if (!leader || !number || !ballot_ordering || !ballot_replica || !resp_mask || !reply_status || !reply_vote_ballot || !reply_triple)
    return FPX_EINVAL;
  const int n = e->st.n;
  const size_t mp = ((size_t)m + 63) & ~(size_t)63;
  int rc;
  if ((rc = grow(e, &e->cl, mp * 4 * 7 + mp + (size_t)m * n * 4 * 3 + 2048))) return rc;
  char* p = (char*)e->cl.p;
  auto take = [&](size_t sz) { char* q = p; p += (sz + 63) & ~(size_t)63; return q; };
  int32_t *d_leader = (int32_t*)take(mp * 4), *d_number = (int32_t*)take(mp * 4), *d_bo = (int32_t*)take(mp * 4);
  int32_t *d_br = (int32_t*)take(mp * 4), *d_act = (int32_t*)take(mp * 4), *d_src = (int32_t*)take(mp * 4), *d_tr = (int32_t*)take(mp * 4);
  uint8_t* d_mask = (uint8_t*)take(mp);
  int32_t *d_rs = (int32_t*)take((size_t)m * n * 4), *d_rv = (int32_t*)take((size_t)m * n * 4), *d_rt = (int32_t*)take((size_t)m * n * 4);
  EHIP(e, hipMemcpyAsync(d_leader, leader, (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemcpyAsync(d_number, number, (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemcpyAsync(d_bo, ballot_ordering, (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemcpyAsync(d_br, ballot_replica, (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemcpyAsync(d_mask, resp_mask, (size_t)m, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemcpyAsync(d_rs, reply_status, (size_t)m * n * 4, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemcpyAsync(d_rv, reply_vote_ballot, (size_t)m * n * 4, hipMemcpyHostToDevice, e->stream));
  EHIP(e, hipMemcpyAsync(d_rt, reply_triple, (size_t)m * n * 4, hipMemcpyHostToDevice, e->stream));
  RcBatch b;
  memset(&b, 0, sizeof(b));
  b.m = m, b.as_intended = as_intended ? 1 : 0, b.leader = d_leader, b.number = d_number, b.b_ord = d_bo, b.b_rep = d_br;
  b.resp_mask = d_mask, b.rs = d_rs, b.rv = d_rv, b.rt = d_rt, b.action = d_act, b.source = d_src, b.triple = d_tr;
  hipLaunchKernelGGL(k_cl_recover, dim3((m + 255) / 256), dim3(256), 0, e->stream, e->st, b);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    e->last_hip = (int)le;
    return FPX_EHIP;
  }
  if (action) EHIP(e, hipMemcpyAsync(action, d_act, (size_t)m * 4, hipMemcpyDeviceToHost, e->stream));
  if (source) EHIP(e, hipMemcpyAsync(source, d_src, (size_t)m * 4, hipMemcpyDeviceToHost, e->stream));
  if (triple) EHIP(e, hipMemcpyAsync(triple, d_tr, (size_t)m * 4, hipMemcpyDeviceToHost, e->stream));
  return fpx_epx_sync(e);
}

int32_t fpx_epx_accept(fpx_epx* e, int32_t m, const int32_t* leader, const int32_t* number, const int32_t* ballot_ordering,
                       const int32_t* ballot_replica, const int32_t* triple_id, const int32_t* key, const uint8_t* is_set,
                       const uint8_t* target_mask, uint8_t* ok_bits, uint8_t* nack_bits, uint8_t* commit_bits,
                       int32_t* nack_ballot, uint8_t* committed) {
  return cl_run(e, 1, m, leader, number, ballot_ordering, ballot_replica, triple_id, key, is_set, target_mask, ok_bits,
                nack_bits, commit_bits, nack_ballot, committed, nullptr, nullptr, nullptr);
}

int32_t fpx_epx_handle_preaccept(fpx_epx* e, int32_t m, const int32_t* leader, const int32_t* number,
                                 const int32_t* ballot_ordering, const int32_t* ballot_replica, const int32_t* key,
                                 const uint8_t* is_set, const int32_t* triple_id, const int32_t* deps_in,
                                 const int32_t* deps_in_values_end, const uint8_t* target_mask, uint8_t* ok_bits,
                                 uint8_t* resend_bits, uint8_t* nack_bits, uint8_t* commit_bits, int32_t* nack_ballot,
                                 int32_t* reply_deps, int32_t* reply_values_end, int32_t* reply_triple) {
  if (!e || m < 0) return FPX_EINVAL;
  EpxDeviceGuard _dg(e->cfg.device);
  if (e->st.num_instances <= 0) return FPX_EINVAL;
  if (m == 0) return FPX_OK;
  if (!leader || !number || !ballot_ordering || !ballot_replica || !key || !is_set || !deps_in || !target_mask)
    return FPX_EINVAL;
  const int n = e->st.n;
  const int tiles = (m + CL_TILE - 1) / CL_TILE;
  const size_t mp = ((size_t)m + 63) & ~(size_t)63;
  int rc;
  if ((rc = grow(e, &e->kv, (size_t)n * m * 8))) return rc;
  if ((rc = grow(e, &e->kv2, (size_t)n * m * 8))) return rc;
  if ((rc = grow(e, &e->tick, (size_t)n * e->st.num_keys * 2 * n * 4))) return rc;
  if ((rc = grow(e, &e->seg, (size_t)n * e->st.num_keys * 8))) return rc;
  if ((rc = grow(e, &e->conf, (size_t)m * n * (n <= 4 ? 4 : 8) * 4))) return rc;
  // staging: 7 int32 inputs + deps_in [m][n], is_set, target, 4 reply bit arrays, nack_ballot, reply_deps [m][n][n],
  // reply_end / reply_triple [m][n], act / nackflag [n][m], contrib [n][m], tilemax [n][tiles]
  const size_t bytes = mp * 4 * 7 + (size_t)m * n * 4 + mp * 6 + mp * 4 + (size_t)m * n * n * 4 + (size_t)m * n * 4 * 2 +
                       (size_t)n * mp * 2 + (size_t)n * mp * 4 + (size_t)n * tiles * 4 + 2048;
  if ((rc = grow(e, &e->hp, bytes))) return rc;
  char* p = (char*)e->hp.p;
  auto take = [&](size_t sz) { char* q = p; p += (sz + 63) & ~(size_t)63; return q; };
  int32_t* d_leader = (int32_t*)take(mp * 4); int32_t* d_number = (int32_t*)take(mp * 4);
  int32_t* d_bo = (int32_t*)take(mp * 4); int32_t* d_br = (int32_t*)take(mp * 4); int32_t* d_key = (int32_t*)take(mp * 4);
  int32_t* d_tr = (int32_t*)take(mp * 4); int32_t* d_dend = (int32_t*)take(mp * 4);
  int32_t* d_din = (int32_t*)take((size_t)m * n * 4);
  uint8_t* d_set = (uint8_t*)take(mp); uint8_t* d_tgt = (uint8_t*)take(mp);
  uint8_t* d_ok = (uint8_t*)take(mp); uint8_t* d_resend = (uint8_t*)take(mp); uint8_t* d_nack = (uint8_t*)take(mp);
  uint8_t* d_com = (uint8_t*)take(mp);
  int32_t* d_nb = (int32_t*)take(mp * 4);
  int32_t* d_rd = (int32_t*)take((size_t)m * n * n * 4); int32_t* d_re = (int32_t*)take((size_t)m * n * 4);
  int32_t* d_rt = (int32_t*)take((size_t)m * n * 4);
  uint8_t* d_act = (uint8_t*)take((size_t)n * m); uint8_t* d_flag = (uint8_t*)take((size_t)n * m);
  int32_t* d_contrib = (int32_t*)take((size_t)n * m * 4); int32_t* d_tm = (int32_t*)take((size_t)n * tiles * 4);
  auto up = [&](void* dst, const void* src, size_t sz) { return hipMemcpyAsync(dst, src, sz, hipMemcpyHostToDevice, e->stream); };
  EHIP(e, up(d_leader, leader, (size_t)m * 4));
  EHIP(e, up(d_number, number, (size_t)m * 4));
  EHIP(e, up(d_bo, ballot_ordering, (size_t)m * 4));
  EHIP(e, up(d_br, ballot_replica, (size_t)m * 4));
  EHIP(e, up(d_key, key, (size_t)m * 4));
  if (triple_id) EHIP(e, up(d_tr, triple_id, (size_t)m * 4));
  if (deps_in_values_end) EHIP(e, up(d_dend, deps_in_values_end, (size_t)m * 4));
  EHIP(e, up(d_din, deps_in, (size_t)m * n * 4));
  EHIP(e, up(d_set, is_set, (size_t)m));
  EHIP(e, up(d_tgt, target_mask, (size_t)m));
  EHIP(e, hipMemsetAsync(d_ok, 0, mp * 4, e->stream));  // ok, resend, nack, commit are contiguous
  EHIP(e, hipMemsetAsync(d_nb, 0xFF, mp * 4, e->stream));
  HpBatch hb;
  memset(&hb, 0, sizeof(hb));
  hb.m = m, hb.leader = d_leader, hb.number = d_number, hb.b_ord = d_bo, hb.b_rep = d_br, hb.key = d_key, hb.is_set = d_set;
  hb.triple = triple_id ? d_tr : nullptr, hb.deps_in = d_din, hb.dend_in = deps_in_values_end ? d_dend : nullptr;
  hb.target = d_tgt, hb.ok_bits = d_ok, hb.resend_bits = d_resend, hb.nack_bits = d_nack, hb.commit_bits = d_com;
  hb.reply_deps = d_rd, hb.reply_end = d_re, hb.reply_triple = d_rt;
  hb.act = d_act, hb.contrib = d_contrib, hb.nackflag = d_flag, hb.kv = (uint2*)e->kv.p;
  hb.conf = (int32_t*)e->conf.p, hb.tick = (int32_t*)e->tick.p;
  if (++e->cl_run == 0) {  // stamp space exhausted: start over
    EHIP(e, hipMemsetAsync(e->st.cl_stamp, 0, (size_t)n * e->st.num_instances * 4, e->stream));
    e->cl_run = 1;
  }
  hb.run_id = e->cl_run;
  const dim3 gm((m + 255) / 256), gmn((unsigned)(((long long)m * n + 255) / 256)), blk(256);
  hipLaunchKernelGGL(k_hp_validate, gm, blk, 0, e->stream, e->st, hb);
  hipLaunchKernelGGL(k_hp_gate, gmn, blk, 0, e->stream, e->st, hb);
  // the conflict scan of what each replica processes, in array order: K5's sort / segments / scan
  EpxBatch sb;
  memset(&sb, 0, sizeof(sb));
  sb.m = m, sb.number = d_number, sb.kv = hb.kv;
  const uint32_t* key_totals = nullptr;
  int key_buckets = 0;
  sb.kv_sorted = sort_by_key(e, m, hb.kv, (uint2*)e->kv2.p, nullptr, &rc, &key_totals, &key_buckets);
  if (rc) return rc;
  sb.tick = (int32_t*)e->tick.p, sb.seg = (int32_t*)e->seg.p, sb.conf = (int32_t*)e->conf.p;
  launch_segments(e, sb, key_totals, key_buckets);
  // the largestBallot every Nack carries: prefix max per replica over the ballots it took in (as for Prepare / Accept)
  ClBatch cb;
  memset(&cb, 0, sizeof(cb));
  cb.m = m, cb.contrib = d_contrib, cb.nackflag = d_flag, cb.tilemax = d_tm, cb.nack_ballot = d_nb;
  hipLaunchKernelGGL(k_cl_tilemax, dim3(tiles, n), blk, 0, e->stream, cb, tiles);
  hipLaunchKernelGGL(k_cl_tilescan, dim3(n), blk, 0, e->stream, e->st, cb, tiles);
  hipLaunchKernelGGL(k_cl_nacks, dim3(tiles, n), blk, 0, e->stream, cb, tiles);
  switch (n) {
    case 3: launch_hp<3>(e, sb, hb); break;
    case 5: launch_hp<5>(e, sb, hb); break;
    default: launch_hp<7>(e, sb, hb); break;
  }
  const long long tot = (long long)e->st.num_keys * n * n;
  hipLaunchKernelGGL(k_hp_commit, dim3((unsigned)((tot + 255) / 256)), blk, 0, e->stream, e->st, hb);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    e->last_hip = (int)le;
    return FPX_EHIP;
  }
  auto down = [&](void* dst, const void* src, size_t sz) { return hipMemcpyAsync(dst, src, sz, hipMemcpyDeviceToHost, e->stream); };
  if (ok_bits) EHIP(e, down(ok_bits, d_ok, (size_t)m));
  if (resend_bits) EHIP(e, down(resend_bits, d_resend, (size_t)m));
  if (nack_bits) EHIP(e, down(nack_bits, d_nack, (size_t)m));
  if (commit_bits) EHIP(e, down(commit_bits, d_com, (size_t)m));
  if (nack_ballot) EHIP(e, down(nack_ballot, d_nb, (size_t)m * 4));
  if (reply_deps) EHIP(e, down(reply_deps, d_rd, (size_t)m * n * n * 4));
  if (reply_values_end) EHIP(e, down(reply_values_end, d_re, (size_t)m * n * 4));
  if (reply_triple) EHIP(e, down(reply_triple, d_rt, (size_t)m * n * 4));
  return fpx_epx_sync(e);
}

int32_t fpx_epx_read_cmdlog_deps(fpx_epx* e, int32_t replica, int32_t leader, int32_t number, int32_t* deps,
                                 int32_t* values_end) {
  if (!e || !deps || !values_end || e->st.num_instances <= 0 || replica < 0 || replica >= e->st.n || leader < 0 ||
      leader >= e->st.n || number < 0 || number >= e->st.num_instances)
    return FPX_EINVAL;
  EpxDeviceGuard _dg(e->cfg.device);
  const size_t c = ((size_t)replica * e->st.n + leader) * e->st.num_instances + number;
  EHIP(e, hipStreamSynchronize(e->stream));
  EHIP(e, hipMemcpy(deps, e->st.cl_deps + c * e->st.n, (size_t)e->st.n * 4, hipMemcpyDeviceToHost));
  EHIP(e, hipMemcpy(values_end, e->st.cl_dend + c, 4, hipMemcpyDeviceToHost));
  return FPX_OK;
}

int32_t fpx_epx_read_cmdlog(fpx_epx* e, int32_t replica, int32_t leader, int32_t number, int32_t out[5]) {
  if (!e || !out || e->st.num_instances <= 0 || replica < 0 || replica >= e->st.n || leader < 0 || leader >= e->st.n ||
      number < 0 || number >= e->st.num_instances)
    return FPX_EINVAL;
  EpxDeviceGuard _dg(e->cfg.device);
  const size_t c = ((size_t)replica * e->st.n + leader) * e->st.num_instances + number;
  uint8_t kind = 0;
  EHIP(e, hipStreamSynchronize(e->stream));
  EHIP(e, hipMemcpy(&kind, e->st.cl_status + c, 1, hipMemcpyDeviceToHost));
  out[0] = kind;
  EHIP(e, hipMemcpy(&out[1], e->st.cl_ballot + c, 4, hipMemcpyDeviceToHost));
  EHIP(e, hipMemcpy(&out[2], e->st.cl_vote + c, 4, hipMemcpyDeviceToHost));
  EHIP(e, hipMemcpy(&out[3], e->st.cl_triple + c, 4, hipMemcpyDeviceToHost));
  EHIP(e, hipMemcpy(&out[4], e->st.largest + replica, 4, hipMemcpyDeviceToHost));
  return FPX_OK;
}

int32_t fpx_epx_info(fpx_epx* e, int32_t* num_replicas, int32_t* num_keys, int32_t* num_instances) {
  if (!e) return FPX_EINVAL;
  if (num_replicas) *num_replicas = e->st.n;
  if (num_keys) *num_keys = e->st.num_keys;
  if (num_instances) *num_instances = e->st.num_instances;
  return FPX_OK;
}

int32_t fpx_epx_read_index(fpx_epx* e, int32_t replica, int32_t key, int32_t* gets, int32_t* sets) {
  if (!e || replica < 0 || replica >= e->st.n || key < 0 || key >= e->st.num_keys) return FPX_EINVAL;
  EpxDeviceGuard _dg(e->cfg.device);
  const size_t off = ((size_t)replica * e->st.num_keys + key) * e->st.n;
  EHIP(e, hipStreamSynchronize(e->stream));
  if (gets) EHIP(e, hipMemcpy(gets, e->st.gets + off, (size_t)e->st.n * 4, hipMemcpyDeviceToHost));
  if (sets) EHIP(e, hipMemcpy(sets, e->st.sets + off, (size_t)e->st.n * 4, hipMemcpyDeviceToHost));
  return FPX_OK;
}

}  // extern "C"
