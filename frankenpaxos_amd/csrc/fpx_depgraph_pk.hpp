// fpx_depgraph_pk.hpp -- the device dependency graph's PACKED path (round 5): the algorithm of fpx_depgraph_dev.hpp with
// half the bytes per vertex and one kernel less per closure round.  Included by fpx_epaxos.hip behind fpx_depgraph_dev.hpp.
//
// What bounds the closure rounds is HBM traffic: every round every vertex reads its closure, gathers n prefix rows and
// writes a row or two (fpx_depgraph_dev.hpp: 8 ints = 32 bytes per row, a prefix kernel and a gather kernel per round:
// 288 B per vertex and round; 4 - 7 rounds at 2^20 commands because nearly every vertex moves in every round but the last
// one or two -- FPX_DG_DEBUG prints the counts).  Here, for n <= 5 replicas and columns of fewer than 2^21 - 2 instances:
//   * a row is 16 bytes: n watermarks RELATIVE to the column's executed prefix, 21 bits each (values beyond the column --
//     an uncommitted instance's "everything" -- saturate at count + 1, which is all eligibility and the gathers ask);
//   * the prefix max of a column is kept as (prefix inside the vertex's 256-vertex workgroup, carry of the workgroups before
//     it): the gather kernel itself scans its workgroup's new closures for the next round (`lp`, `bt`), a one-workgroup-
//     per-column kernel turns the workgroup totals into carries (`cy`: 4096 rows, L2-resident), and a gather reads
//     lp[j] and cy[j / 256].  No separate prefix pass over the vertices: 128 B per vertex and round.
// Everything else -- cycle test, keys, the two sorts, component starts -- is the wide path's, on packed rows.
#pragma once

constexpr int PK_BITS = 21;
constexpr unsigned long long PK_MASK = (1ull << PK_BITS) - 1ull;
constexpr int PK_MAX_COUNT = (1 << PK_BITS) - 3;

__device__ __forceinline__ void pk_unpack(const ulonglong2 r, int* c) {  // c[5]
  c[0] = (int)(r.x & PK_MASK), c[1] = (int)((r.x >> PK_BITS) & PK_MASK), c[2] = (int)((r.x >> (2 * PK_BITS)) & PK_MASK);
  c[3] = (int)(r.y & PK_MASK), c[4] = (int)((r.y >> PK_BITS) & PK_MASK);
}
__device__ __forceinline__ ulonglong2 pk_pack(const int* c) {
  ulonglong2 r;
  r.x = (unsigned long long)(unsigned)c[0] | ((unsigned long long)(unsigned)c[1] << PK_BITS) | ((unsigned long long)(unsigned)c[2] << (2 * PK_BITS));
  r.y = (unsigned long long)(unsigned)c[3] | ((unsigned long long)(unsigned)c[4] << PK_BITS);
  return r;
}
__device__ __forceinline__ uint32_t pk_hash(const ulonglong2 r, int bits) {
  uint32_t h = 0x9E3779B9u;
  h = kp_mix32(h ^ (uint32_t)r.x) + 0x7F4A7C15u;
  h = kp_mix32(h ^ (uint32_t)(r.x >> 32)) + 0x7F4A7C15u;
  h = kp_mix32(h ^ (uint32_t)r.y) + 0x7F4A7C15u;
  h = kp_mix32(h ^ (uint32_t)(r.y >> 32)) + 0x7F4A7C15u;
  return h >> (32 - bits);
}

struct DpArgs {
  int m, n, stride;
  int32_t first[8], count[8], base[8];   // per column: first id, instances, first vertex
  int32_t nblk[8], blk_base[8];          // per column: workgroups of 256 vertices, the first one's index
  int nblocks;
  const int32_t* leader;
  const int32_t* number;
  const int32_t* packed;
  const uint8_t* mask;
  int32_t* msg_of;                       // [m]
  ulonglong2* direct;                    // [m] direct covers
  ulonglong2* clo;                       // [m] closures
  ulonglong2* lp[2];                     // [m] inclusive prefix max of the closures INSIDE the vertex's workgroup
  ulonglong2* bt[2];                     // [nblocks] the workgroups' maxima
  ulonglong2* cy[2];                     // [nblocks] max over the workgroups of the column BEFORE this one
  uint2* pairs;
  uint2* pairs2;
  uint32_t* key32;
  int32_t* tstarts;
  int32_t* belig;                        // [ceil(m / 256)] executable vertices per workgroup of k_dp_keys (summed by k_dp_rekey)
  int32_t* ctl;                          // as DgArgs::ctl
  volatile int32_t* host;
  int32_t seq;
  int32_t count_moved;
  int32_t hash_bits;                     // as DgArgs::hash_bits
  int32_t* order;
  int32_t* comp;
};

// The gathers of a closure round (and of the cycle test) read rows of OTHER vertices: the prefix row just below each of a
// vertex's n watermarks.  Neighbouring vertices of a column were proposed at neighbouring times, so their watermarks point
// at neighbouring rows of every column -- a window of a few thousand rows per column that slides along with the vertices.
// Workgroups are dealt round-robin to the 8 XCDs (each with an L2 of its own): with workgroup b on vertices 256 b .. the
// 256 workgroups resident on one XCD sat 8 blocks apart, their windows covered half of every column (20 MB against a 4 MB
// L2) and every 64-byte line of `lp` came from HBM ~17 times per round (280 MB read per round for a 16 MB array,
// profiles/r06_cfg_pmc.md).  Here the workgroups of one XCD take CONSECUTIVE blocks: an XCD works its way through one
// eighth of the vertices and its resident workgroups' windows overlap (profiles/r06_depgraph_dev.md).  Placement is a
// matter of speed only: any mapping of workgroups to XCDs gives the same result.
__device__ __forceinline__ int dp_logical_block(int bid, int nblocks) {
  const int per = (nblocks + 7) >> 3;
  return (bid & 7) * per + (bid >> 3);  // (may be >= nblocks: the caller leaves)
}
__host__ __device__ __forceinline__ int dp_grid(int nblocks) { return 8 * ((nblocks + 7) >> 3); }

__device__ __forceinline__ int dp_col_of_block(const DpArgs& a, int n, int b) {
  int col = 0;
  while (col + 1 < n && b >= a.blk_base[col + 1]) ++col;
  return col;
}
__device__ __forceinline__ int dp_col_of_vertex(const DpArgs& a, int n, int v) {
  int col = 0;
  while (col + 1 < n && v >= a.base[col + 1]) ++col;
  return col;
}

template <int N>
__global__ void __launch_bounds__(256) k_dp_scatter(const DpArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.m) return;
  const int L = a.leader[i], x = a.number[i];
  int v = -1;
  if (L >= 0 && L < N) {
    const int j = x - a.first[L];
    if (j >= 0 && j < a.count[L]) v = a.base[L] + j;
  }
  if (v < 0) {  // outside its column
    a.ctl[1] = 1;
    return;
  }
  // (An instance handed in twice leaves another one of its column without a message -- the columns hold exactly m
  // instances for m messages -- and k_dp_scan0 reports that one: no returning atomic per message is needed.  The plain
  // store may race with the other copy's; either message's index is a valid owner for the error path that follows.)
  a.msg_of[v] = i;
  const int32_t* line = a.packed + (size_t)i * a.stride;
  const bool committed = !a.mask || a.mask[i];
  int d[5] = {0, 0, 0, 0, 0};
#pragma unroll
  for (int l = 0; l < N; ++l) {
    int w = line[l];
    if (w < 0) a.ctl[1] = 1, w = 0;
    if (l == L) {
      const int end = line[2 * N];  // explicit ids x + 1 .. end - 1 of the own column (dependencies.subtractOne, Replica.scala:582)
      if (committed && end > 0) {
        if (end <= x + 1 || w != x) a.ctl[1] = 1;
        w = end;
      }
    }
    // relative to the column's executed prefix, saturating one past the column; an instance that is not committed can be
    // waited for only: its cover is beyond every column, so is the closure of whatever reaches it
    const int rel = w - a.first[l];
    d[l] = committed ? min(max(rel, 0), a.count[l] + 1) : a.count[l] + 1;
  }
  a.direct[v] = pk_pack(d);  // (k_dp_scan0 copies it into the closure: in vertex order, as whole lines)
}

// inclusive max-scan of one row per thread over the workgroup's 256 vertices; returns the workgroup's maximum in tot[]
template <int N>
__device__ __forceinline__ void dp_block_scan(int* c, int (*sh)[5], int* tot) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int l = 0; l < N; ++l) {
    c[l] = wave_incl_max(c[l]);
    if (lane == 63) sh[w][l] = c[l];
  }
  __syncthreads();
#pragma unroll
  for (int l = 0; l < N; ++l) {
    int before = 0;
    for (int w2 = 0; w2 < w; ++w2) before = imax(before, sh[w2][l]);
    c[l] = imax(c[l], before);
    tot[l] = imax(imax(sh[0][l], sh[1][l]), imax(sh[2][l], sh[3][l]));
  }
}

// lp / bt of the direct covers (what round 1 gathers from)
template <int N>
__global__ void __launch_bounds__(256) k_dp_scan0(const DpArgs a) {
  __shared__ int sh[4][5];
  const int lb = dp_logical_block((int)blockIdx.x, a.nblocks);
  if (lb >= a.nblocks) return;
  const int col = dp_col_of_block(a, N, lb);
  const int j = (lb - a.blk_base[col]) * 256 + (int)threadIdx.x;
  const bool live = j < a.count[col];
  const int v = a.base[col] + j;
  int c[5] = {0, 0, 0, 0, 0}, tot[5] = {0, 0, 0, 0, 0};
  if (live) {
    if (a.msg_of[v] < 0) {
      a.ctl[1] = 1;  // an instance of the column that was not handed in: the columns are not dense
    } else {
      const ulonglong2 r = a.direct[v];
      a.clo[v] = r;
      pk_unpack(r, c);
    }
  }
  dp_block_scan<N>(c, sh, tot);
  if (live) a.lp[0][v] = pk_pack(c);
  if (threadIdx.x == 0) a.bt[0][lb] = pk_pack(tot);
}

// cy[b] = max of bt over the workgroups of b's column before b: one workgroup per column
template <int N>
__global__ void __launch_bounds__(1024) k_dp_carry(const DpArgs a, int buf, int k) {
  __shared__ int sh[16][5];
  __shared__ int run[5];
  if (k > 1 && a.ctl[8 + k - 1] == 0) return;
  const int col = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x < 5) run[threadIdx.x] = 0;
  __syncthreads();
  for (int b0 = 0; b0 < a.nblk[col]; b0 += 1024) {
    const int b = b0 + (int)threadIdx.x;
    int c[5] = {0, 0, 0, 0, 0};
    if (b < a.nblk[col]) pk_unpack(a.bt[buf][a.blk_base[col] + b], c);
#pragma unroll
    for (int l = 0; l < N; ++l) {
      c[l] = wave_incl_max(c[l]);
      if (lane == 63) sh[w][l] = c[l];
    }
    __syncthreads();
    int excl[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int l = 0; l < N; ++l) {
      int before = run[l];
      for (int w2 = 0; w2 < w; ++w2) before = imax(before, sh[w2][l]);
      // exclusive: what lies before this workgroup = the carry so far v the wavefront's inclusive scan shifted by one lane
      const int prev = wave_shr1(c[l]);
      excl[l] = imax(before, prev);
    }
    if (b < a.nblk[col]) a.cy[buf][a.blk_base[col] + b] = pk_pack(excl);
    __syncthreads();
    if (threadIdx.x < N) {
      int nx = run[threadIdx.x];
      for (int w2 = 0; w2 < 16; ++w2) nx = imax(nx, sh[w2][threadIdx.x]);
      run[threadIdx.x] = nx;
    }
    __syncthreads();
  }
}

// c_v = c_v  v  max over l of (prefix max of column l below c_v[l]); then this workgroup's scan for the next round
template <int N>
__global__ void __launch_bounds__(256) k_dp_relax(const DpArgs a, int cur, int k) {
  __shared__ int sh[4][5];
  if (k > 1 && a.ctl[8 + k - 1] == 0) return;  // the round before moved nothing: lp / cy of both halves are final
  const int lb = dp_logical_block((int)blockIdx.x, a.nblocks);
  if (lb >= a.nblocks) return;
  const int col = dp_col_of_block(a, N, lb);
  const int j = (lb - a.blk_base[col]) * 256 + (int)threadIdx.x;
  const bool live = j < a.count[col];
  const int v = a.base[col] + j;
  int c[5] = {0, 0, 0, 0, 0}, tot[5] = {0, 0, 0, 0, 0};
  bool moved = false;
  if (live) {
    int o[5];
    pk_unpack(a.clo[v], o);
    ulonglong2 rows[N], carry[N];
#pragma unroll
    for (int l = 0; l < N; ++l) {
      const int jj = min(o[l], a.count[l]) - 1;  // the last vertex of column l below the watermark
      const bool has = jj >= 0;
      rows[l] = has ? a.lp[cur][a.base[l] + jj] : make_ulonglong2(0ull, 0ull);
      carry[l] = has ? a.cy[cur][a.blk_base[l] + (jj >> 8)] : make_ulonglong2(0ull, 0ull);
    }
#pragma unroll
    for (int l = 0; l < 5; ++l) c[l] = l < N ? o[l] : 0;
#pragma unroll
    for (int l = 0; l < N; ++l) {
      int r1[5], r2[5];
      pk_unpack(rows[l], r1), pk_unpack(carry[l], r2);
#pragma unroll
      for (int q = 0; q < N; ++q) c[q] = imax(c[q], imax(r1[q], r2[q]));
    }
#pragma unroll
    for (int l = 0; l < N; ++l) moved = moved || c[l] != o[l];
    if (moved) a.clo[v] = pk_pack(c);
  }
  if (__any(moved) && (threadIdx.x & 63) == 0 && a.ctl[8 + k] == 0) a.ctl[8 + k] = 1;
  if (a.count_moved) {
    const unsigned long long bal = __ballot(moved);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&a.ctl[24 + k], (int)__popcll(bal));
  }
  dp_block_scan<N>(c, sh, tot);
  if (live) a.lp[cur ^ 1][v] = pk_pack(c);
  if (threadIdx.x == 0) a.bt[cur ^ 1][lb] = pk_pack(tot);
}

// the prefix max of column l below relative watermark w (w >= 1), from either half (both are final after the rounds)
__device__ __forceinline__ void dp_prefix(const DpArgs& a, int l, int w, int* out) {
  const int jj = min(w, a.count[l]) - 1;
  int r1[5], r2[5];
  pk_unpack(a.lp[0][a.base[l] + jj], r1), pk_unpack(a.cy[0][a.blk_base[l] + (jj >> 8)], r2);
#pragma unroll
  for (int q = 0; q < 5; ++q) out[q] = imax(r1[q], r2[q]);
}

// the cycle test of one executable vertex whose closure reaches past it in its own column (c[L] > x): kind 0 = on a cycle
// (the closure of a direct dependency covers x in column L), 1 = not
template <int N>
__device__ __forceinline__ uint32_t dp_kind_of_candidate(const DpArgs& a, int L, int x, const int* d) {
  int back = 0;
#pragma unroll
  for (int l = 0; l < N; ++l) {
    const int bound = l == L ? min(d[l], x) : d[l];  // own column: the prefix below x here, the explicit ids below
    if (bound >= 1) {
      int row[5];
      dp_prefix(a, l, bound, row);
      back = imax(back, row[L]);
    }
  }
  for (int y = x + 1; y < min(d[L], a.count[L]); ++y) {
    int row[5];
    pk_unpack(a.clo[a.base[L] + y], row);
    back = imax(back, row[L]);
  }
  return back > x ? 0u : 1u;
}

template <int N>
__global__ void __launch_bounds__(256) k_dp_keys(const DpArgs a) {
  __shared__ int block_eligible, ncand;
  __shared__ int cand[256];
  const int vblocks = (a.m + 255) >> 8;
  const int lb = dp_logical_block((int)blockIdx.x, vblocks);
  if (lb >= vblocks) return;
  const int v = lb * 256 + (int)threadIdx.x;
  if (threadIdx.x == 0) block_eligible = 0, ncand = 0;
  __syncthreads();
  bool eligible = false;
  if (v < a.m) {
    int c[5];
    const ulonglong2 cr = a.clo[v];
    pk_unpack(cr, c);
    const int L = dp_col_of_vertex(a, N, v);
    const int x = v - a.base[L];  // relative id
    eligible = true;
    uint32_t sum = 0;
#pragma unroll
    for (int l = 0; l < N; ++l) {
      eligible = eligible && c[l] <= a.count[l];
      sum += (uint32_t)c[l];
    }
    // Every row the cycle test gathers is the closure of something v reaches, so its column L lies below c[L]: a vertex
    // whose own closure does not reach past itself in its column (c[L] <= x: nine in ten of a FIFO tick) is on no cycle
    // and has no explicit ids to walk -- kind 2, nothing asked.  (An inexecutable vertex has "everything" in its rows and
    // would walk the rest of its column: it is not asked either.)  The others -- a few per wavefront -- are handed to the
    // workgroup's FIRST lanes below, so that one wavefront gathers with its lanes filled instead of four with a tenth each.
    if (eligible && c[L] > x) {
      cand[atomicAdd(&ncand, 1)] = v;
    } else {
      a.key32[v] = eligible ? (sum * 3u + 2u) : 0xffffffffu;
      a.pairs[v] = make_uint2(0u, (uint32_t)v);
    }
  }
  const unsigned long long bal = __ballot(eligible);
  if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&block_eligible, (int)__popcll(bal));
  __syncthreads();
  if ((int)threadIdx.x < ncand) {
    const int u = cand[threadIdx.x];
    int c[5], d[5];
    const ulonglong2 cr = a.clo[u];
    pk_unpack(cr, c), pk_unpack(a.direct[u], d);
    const int L = dp_col_of_vertex(a, N, u);
    const int x = u - a.base[L];
    uint32_t sum = 0;
#pragma unroll
    for (int l = 0; l < N; ++l) sum += (uint32_t)c[l];
    const uint32_t kind = dp_kind_of_candidate<N>(a, L, x, d);
    a.key32[u] = sum * 3u + kind;
    a.pairs[u] = make_uint2(kind == 0u ? pk_hash(cr, a.hash_bits) : 0u, (uint32_t)u);
  }
  // (4096 workgroups adding to ONE word serialise in its L2 channel: ~35 of this kernel's 54 us were that atomic, the same
  // effect k_validate's round word showed in round 2; each workgroup leaves its count, k_dp_rekey's first workgroup adds them up)
  if (threadIdx.x == 0) a.belig[lb] = block_eligible;
}

__global__ void __launch_bounds__(256) k_dp_rekey(const DpArgs a) {
  __shared__ uint32_t sh[8];
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < a.m) a.pairs[p].x = a.key32[a.pairs[p].y];
  if (blockIdx.x == 0) {  // the number of executables (what k_dp_count_starts / k_dp_emit / k_dp_publish read from ctl[3])
    const int vblocks = (a.m + 255) >> 8;
    uint32_t s = 0;
    for (int b = (int)threadIdx.x; b < vblocks; b += 256) s += (uint32_t)a.belig[b];
    const uint32_t ex = block_excl_sum(s, sh);
    if (threadIdx.x == 255) a.ctl[3] = (int32_t)(ex + s);
  }
}

// does a component start at position p of the sorted order?  (dg_starts on packed rows)
__device__ __forceinline__ uint32_t dp_starts(const DpArgs& a, int p, int executables) {
  if (p >= executables) return 0u;
  const uint2 e = a.pairs[p];
  if (e.x % 3u != 0u || p == 0) return 1u;
  const uint2 f = a.pairs[p - 1];
  if (f.x != e.x) return 1u;
  const ulonglong2 ce = a.clo[e.y], cf = a.clo[f.y];
  if (ce.x == cf.x && ce.y == cf.y) return 0u;
  if (pk_hash(ce, a.hash_bits) == pk_hash(cf, a.hash_bits)) a.ctl[2] = 1;  // two different closures with one key and one hash: their members may interleave
  return 1u;
}

__global__ void __launch_bounds__(256) k_dp_count_starts(const DpArgs a) {
  __shared__ uint32_t sh[8];
  const int executables = a.ctl[3];
  const int t0 = blockIdx.x * DG_TILE;
  uint32_t total = 0;
  for (int j = 0; j < DG_TILE / 256; ++j) total += dp_starts(a, t0 + j * 256 + threadIdx.x, executables);
  const uint32_t ex = block_excl_sum(total, sh);
  if (threadIdx.x == 255) a.tstarts[blockIdx.x] = (int32_t)(ex + total);
}

__global__ void __launch_bounds__(256) k_dp_emit(const DpArgs a) {
  __shared__ uint32_t sh[8];
  __shared__ uint32_t before_tile;
  const int executables = a.ctl[3];
  const int t0 = blockIdx.x * DG_TILE;
  uint32_t mine[DG_TILE / 256];
#pragma unroll
  for (int j = 0; j < DG_TILE / 256; ++j) mine[j] = dp_starts(a, t0 + j * 256 + threadIdx.x, executables);
  {
    // the components that start in the tiles before this one: 256 threads add up the tiles' counts (one thread walking up to
    // 512 of them was ~10 us of the kernel)
    uint32_t s = 0;
    for (int t = (int)threadIdx.x; t < (int)blockIdx.x; t += 256) s += (uint32_t)a.tstarts[t];
    const uint32_t ex = block_excl_sum(s, sh);
    if (threadIdx.x == 255) before_tile = ex + s;
    __syncthreads();
  }
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

// last_k: the rounds of the chunk that were enqueued.  [5] = the last of them still moved (another chunk is needed),
// [6] = how many of them moved something (the next call enqueues one more than that: see dg_execute_packed)
__global__ void k_dp_publish(const DpArgs a, int last_k) {
  if (threadIdx.x < 7) {
    int v = a.ctl[threadIdx.x];
    if (threadIdx.x == 5) v = a.ctl[8 + last_k];
    if (threadIdx.x == 6) {
      v = 0;
      for (int k = 1; k <= last_k; ++k) v += a.ctl[8 + k] != 0 ? 1 : 0;
    }
    a.host[threadIdx.x] = v;
  }
  __threadfence_system();
  if (threadIdx.x == 0) a.host[7] = a.seq;
}
