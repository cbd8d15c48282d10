// fpx_epaxos_kp.hpp -- K5, second form: the tick is partitioned by KEY once, each key is ordered per replica ON CHIP.
// Included by fpx_epaxos.hip inside its anonymous namespace (uses EpxState, EpxBatch, scan_chunk, own_column, ...).
//
// The first form (k_epx_keys -> radix sort of n x m (key, message) pairs -> k_epx_key) moves every command n times
// through HBM as an 8-byte pair, twice each way, and gathers its fields again by message index afterwards: 651 MB
// per 2^20-command tick for 84 B of compulsory traffic per command (profiles/r02_k5_pmc.md).  Here a command
// crosses HBM once more than it has to, as ONE record that holds everything the per-key work needs:
//
//   k_kp_hist      per tile of 2048 messages: how many of each key           (reads key[]: 4 B per command)
//   k_kp_scan      tile offsets within every key, key segments [start, count), and how many keys hold more
//                  commands than the on-chip tables take (written to a page-locked word the host waits for while
//                  the next kernel runs: such a tick goes the first form's way, nothing of it is applied here)
//   k_kp_scatter   validates the tick, writes record(i) = {i, number, leader | is_set | resp | seen, rank[0..n)} to
//                  its place in its key's segment (a tile-local LDS counter + the tile's offset: no order is needed
//                  inside a key), and accumulates the fingerprints that tell a permutation from a non-permutation
//   k_epx_key2<N>  one persistent workgroup per CU, key after key: the key's records -> LDS; per replica the
//                  participating commands as (rank << 11 | slot) words, sorted by an LSD radix sort in LDS (7-bit
//                  digits, wavefront-private counters, ballots rank equal digits); the segmented scans of
//                  scan_chunk on the sorted order (conflict rows stay in registers); the leader's rows D go to
//                  LDS, every responder folds max(conflicts, D) into a per-command max and min (fast path <=> max
//                  == min in every column, and the union the slow path proposes IS the max); decisions leave as
//                  one packed line per command (or the four arrays of the first form); the key's conflict index
//                  is updated in place (what k_epx_commit did for all keys).
//
// HBM traffic per command (n = 5): 35 B of inputs + 32 B record out + 32 B record in + the outputs.
// With a command log (num_instances > 0, n >= 5): k_epx_key2<N, true> scans a second time after the decisions and
// writes the entries of every replica that saw the PreAccept (or of every replica, for a fast-path commit).
#pragma once

constexpr int KP_TILE = 2048;   // messages per partition tile
constexpr int KP_MAXB = 2048;   // keys (one LDS counter each in the partition passes)
constexpr int KP_SLOT_BITS = 11;
constexpr uint32_t KP_SLOT_MASK = (1u << KP_SLOT_BITS) - 1u;
constexpr int KP_RADIX_BITS = 7, KP_RADIX = 1 << KP_RADIX_BITS;
#ifndef KP_SCATTER_MB
#define KP_SCATTER_MB 4
#endif
constexpr int KP_MAX_OCC = 24;   // fullest rank bucket the bucket sort accepts before the key is radix-sorted instead

template <int N> struct KpTile {
  static constexpr int NI = N <= 5 ? 8 : 12;                       // ints per record: i, number, flags, rank[N], padding
  static constexpr int TC = N <= 3 ? 1536 : N <= 5 ? 1152 : 640;   // commands of one key held on chip (<= 2^11)
  static constexpr int W = N <= 3 ? 4 : N <= 5 ? 3 : 2;            // wavefronts per replica
  static constexpr int THREADS = 64 * N * W;
  static constexpr int MAXC = (TC + 63) / 64;
  static constexpr int CPW = (MAXC + W - 1) / W;                   // 64-command chunks per wavefront
  static constexpr int NBK = N <= 5 ? 1024 : 512;                  // rank buckets of the bucket sort
  static constexpr int RQ = (TC * (NI / 4) + THREADS - 1) / THREADS;  // int4 loads per thread for one key's records
  // LDS: records | region R | radix counters + cursors | part totals | misc.  R holds the two sort buffers and the rank
  // buckets while a key is sorted, then the conflict rows [TC][N - 1][N]: the n-2 counted answers and the leader's own
  static constexpr size_t SORT_BYTES = (size_t)2 * N * TC * 4 + (size_t)N * NBK * 4;
  static constexpr int RSTR = (N - 1) * N + 1;  // ints between two commands' conflict rows: odd, so that random commands
                                                // spread over all LDS banks ((N - 1) N = 20 reaches 16 of 64: 40 % of the
                                                // kernel's LDS cycles were bank conflicts)
  static constexpr size_t ROWS_BYTES = (size_t)TC * RSTR * 4;
  static constexpr size_t R_BYTES = SORT_BYTES > ROWS_BYTES ? SORT_BYTES : ROWS_BYTES;
  static constexpr size_t BYTES = (size_t)TC * NI * 4 + R_BYTES + (size_t)2 * N * W * KP_RADIX * 4 + (size_t)N * W * 2 * N * 4 + 256 +
                                  (size_t)N * 2 * N * 4;
  static_assert(BYTES <= 160 * 1024, "LDS of one CU");
};

// record flags: leader | is_set << 3 | resp_mask << 8 | seen_mask << 16
__device__ __forceinline__ int kp_flags(int L, int is_set, unsigned resp, unsigned seen) {
  return L | (is_set ? 8 : 0) | (int)(resp << 8) | (int)(seen << 16);
}

struct KpArgs {
  int m, tiles, B;                  // B = num_keys
  uint32_t* hist;                   // [B][tiles] per-tile key counts -> exclusive offsets of the tile within the key
  uint32_t* tot;                    // [B]
  int32_t* seg;                     // [B][2] start, count of the key's records
  uint32_t* ctl;                    // [0] scan blocks done, [1] keys that do not fit the on-chip tables
  unsigned long long* fp;           // [2 (N + 1)] additive fingerprints: of the indices 0..m-1, then of every rank row
  int32_t* recs;                    // [m][NI]
  volatile uint32_t* host_flag;     // page-locked: [1] = ctl[1], then [0] = seq
  uint32_t seq;
  int tc;                           // KpTile<N>::TC
  int32_t* packed;                  // [m][stride] or null
  int stride;
};

__device__ __forceinline__ unsigned long long kp_mix(unsigned long long z) {  // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// wave64 exclusive sum of one value per lane
__device__ __forceinline__ uint32_t kp_wave_excl_sum(uint32_t v) {
  const int lane = threadIdx.x & 63;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d);
    if (lane >= d) inc += o;
  }
  return inc - v;
}

// KP_HG tiles per workgroup (128 threads each): the counts of a key for the workgroup's tiles are neighbours in
// hist[key][tile] and leave as one 32-byte sector (one tile per workgroup wrote 2 MB of counts as 17 MB of partial sectors)
constexpr int KP_HG = 8;
__global__ void __launch_bounds__(128 * KP_HG) k_kp_hist(const EpxState st, const EpxBatch b, const KpArgs a) {
  extern __shared__ uint32_t kp_h[];  // [KP_HG][B]
  if (blockIdx.x == 0) {
    if (threadIdx.x < 2) a.ctl[threadIdx.x] = 0;
    if (threadIdx.x < 2 * (st.n + 1)) a.fp[threadIdx.x] = 0ull;
  }
  for (int j = threadIdx.x; j < KP_HG * a.B; j += 128 * KP_HG) kp_h[j] = 0;
  __syncthreads();
  const int sub = threadIdx.x >> 7, t = threadIdx.x & 127;
  const int tile = blockIdx.x * KP_HG + sub;
  uint32_t* h = kp_h + sub * a.B;
  const int first = tile * KP_TILE;
  int k[KP_TILE / 128];
#pragma unroll
  for (int j = 0; j < KP_TILE / 128; ++j) {
    const int i = first + j * 128 + t;
    k[j] = (tile < a.tiles && i < a.m) ? b.key[i] : -1;
  }
#pragma unroll
  for (int j = 0; j < KP_TILE / 128; ++j) {
    const int i = first + j * 128 + t;
    if (tile < a.tiles && i < a.m) {
      if (k[j] < 0 || k[j] >= a.B) epx_report(st.status, FPX_EINVAL, i);
      else atomicAdd(&h[k[j]], 1u);
    }
  }
  __syncthreads();
  const int t0 = blockIdx.x * KP_HG, nt = min(KP_HG, a.tiles - t0);
  for (int j = threadIdx.x; j < a.B; j += 128 * KP_HG) {
    uint32_t* out = a.hist + (size_t)j * a.tiles + t0;  // [key][tile]
    if (nt == KP_HG && (a.tiles & 3) == 0) {
      reinterpret_cast<uint4*>(out)[0] = make_uint4(kp_h[j], kp_h[a.B + j], kp_h[2 * a.B + j], kp_h[3 * a.B + j]);
      reinterpret_cast<uint4*>(out)[1] = make_uint4(kp_h[4 * a.B + j], kp_h[5 * a.B + j], kp_h[6 * a.B + j], kp_h[7 * a.B + j]);
    } else {
      for (int q = 0; q < nt; ++q) out[q] = kp_h[q * a.B + j];
    }
  }
}

// exclusive scan of every key's per-tile counts: one wavefront per key, the key's counts are contiguous ([key][tile];
// with [tile][key] a key's column is one word every 4 KB -- 16 workgroups x 128 dependent round trips were 53 us, and
// finer splits of the columns only moved the time around: 23 - 46 us).  No "last workgroup" epilogue here: its
// __threadfence() per workgroup (an L2 write-back on this multi-die GPU) cost 30 us; k_kp_scatter turns the totals into
// key segments itself
constexpr int KP_SCAN_WAVES = 4;
__global__ void __launch_bounds__(64 * KP_SCAN_WAVES) k_kp_scan(const KpArgs a) {
  const int lane = threadIdx.x & 63;
  const int d = blockIdx.x * KP_SCAN_WAVES + (threadIdx.x >> 6);
  if (d < a.B) {
    uint32_t* row = a.hist + (size_t)d * a.tiles;
    const int per = (a.tiles + 63) / 64, t0 = min(a.tiles, lane * per), t1 = min(a.tiles, t0 + per);
    uint32_t sum = 0;
    for (int t = t0; t < t1; ++t) sum += row[t];
    uint32_t run = kp_wave_excl_sum(sum);
    if (lane == 63) a.tot[d] = run + sum;
    for (int t = t0; t < t1; ++t) {
      const uint32_t v = row[t];
      row[t] = run, run += v;
    }
  }
}

// validation of one message exactly as k_epx_keys, record out, fingerprints
template <int N>
__global__ void __launch_bounds__(256) k_kp_scatter(const EpxState st, const EpxBatch b, const KpArgs a) {
  using T = KpTile<N>;
  __shared__ uint32_t cnt[KP_MAXB];
  __shared__ uint32_t goff[KP_MAXB];
  __shared__ uint32_t sh[4];
  __shared__ unsigned long long fsum[4][2 * (N + 1)];
  // workgroups go round-robin over the 8 XCDs: the ones of one XCD take consecutive tiles, whose records are neighbours
  // in every key's segment -- their 32-byte stores meet in the same L2 and leave it as whole lines
  const int tps = (a.tiles + 7) / 8, tile = ((int)blockIdx.x % 8) * tps + (int)blockIdx.x / 8;
  if ((int)blockIdx.x / 8 >= tps || tile >= a.tiles) return;
  // where every key's records start: the exclusive prefix of the key totals, recomputed by every workgroup (4 KB from
  // L2); workgroup 0 also publishes the segments and tells the host how many keys are too big for the on-chip tables
  {
    const int per_t = (a.B + 255) / 256, k0 = threadIdx.x * per_t;
    uint32_t mine = 0, big = 0;
    for (int j = 0; j < per_t; ++j)
      if (k0 + j < a.B) {
        const uint32_t c = a.tot[k0 + j];
        mine += c, big += c > (uint32_t)a.tc ? 1u : 0u;
      }
    uint32_t start = block_excl_sum(mine, sh);
    for (int j = 0; j < per_t; ++j)
      if (k0 + j < a.B) {
        const uint32_t c = a.tot[k0 + j];
        cnt[k0 + j] = 0, goff[k0 + j] = start + a.hist[(size_t)(k0 + j) * a.tiles + tile];
        if (blockIdx.x == 0) a.seg[(size_t)(k0 + j) * 2] = (int32_t)start, a.seg[(size_t)(k0 + j) * 2 + 1] = (int32_t)c;
        start += c;
      }
    if (blockIdx.x == 0) {
      __syncthreads();
      const uint32_t nbig_before = block_excl_sum(big, sh);
      if (threadIdx.x == 255) {
        const uint32_t nbig = nbig_before + big;
        a.ctl[1] = nbig;
        if (a.host_flag) {
          a.host_flag[1] = nbig;
          __threadfence_system();
          a.host_flag[0] = a.seq;
        }
      }
    }
  }
  __syncthreads();
  unsigned long long f[2 * (N + 1)];
#pragma unroll
  for (int q = 0; q < 2 * (N + 1); ++q) f[q] = 0ull;
  const int first = tile * KP_TILE;
  constexpr int MB = KP_SCATTER_MB;  // messages of one thread whose loads are in flight together
  for (int j0 = 0; j0 < KP_TILE / 256; j0 += MB) {
    int Lq[MB], kq[MB], xq[MB], rkq[MB][N];
    unsigned mq[MB], sq[MB], tq[MB];
#pragma unroll
    for (int u = 0; u < MB; ++u) {
      const int i = first + (j0 + u) * 256 + threadIdx.x;
      const bool in = i < a.m;
      Lq[u] = in ? b.leader[i] : 0, kq[u] = in ? b.key[i] : 0, xq[u] = in ? b.number[i] : 0;
      mq[u] = in ? b.resp_mask[i] : 0u, tq[u] = in ? b.is_set[i] : 0u;
      sq[u] = in ? (b.seen_mask ? b.seen_mask[i] : mq[u]) : 0u;
#pragma unroll
      for (int r = 0; r < N; ++r) rkq[u][r] = in ? b.rank[(size_t)r * a.m + i] : 0;
    }
#pragma unroll
    for (int u = 0; u < MB; ++u) {
      const int i = first + (j0 + u) * 256 + threadIdx.x;
      if (i >= a.m) continue;
      const int L = Lq[u], k = kq[u], x = xq[u];
      const unsigned mask = mq[u], seen = sq[u];
      const int is_set = tq[u] ? 1 : 0;
      bool ok = L >= 0 && L < N && x >= 0 && k >= 0 && k < a.B;
      ok = ok && !((mask >> (ok ? L : 0)) & 1u) && (mask >> N) == 0 && (int)__popc(mask) == N - 2;
      ok = ok && (mask & ~seen) == 0 && !((seen >> (ok ? L : 0)) & 1u) && (seen >> N) == 0;
      f[0] += kp_mix((unsigned long long)i + 0x9E3779B97F4A7C15ull), f[1] += kp_mix((unsigned long long)i ^ 0xD1B54A32D192ED03ull);
#pragma unroll
      for (int r = 0; r < N; ++r) {
        ok = ok && rkq[u][r] >= 0 && rkq[u][r] < a.m;
        f[2 + 2 * r] += kp_mix((unsigned long long)(unsigned)rkq[u][r] + 0x9E3779B97F4A7C15ull);
        f[3 + 2 * r] += kp_mix((unsigned long long)(unsigned)rkq[u][r] ^ 0xD1B54A32D192ED03ull);
      }
      if (ok && st.num_instances > 0) {
        // this tick-at-once form covers handlePreAccept's `cmdLog.get(instance) == None` branch only: an instance a
        // participating replica already knows is rejected (nothing of the tick is applied)
        ok = x < st.num_instances;
        const unsigned part = seen | (1u << L);
        for (int r = 0; ok && r < N; ++r)
          if (((part >> r) & 1u) && st.cl_status[((size_t)r * N + L) * st.num_instances + x] != CL_NONE) ok = false;
      }
      if (!ok) {
        epx_report(st.status, FPX_EINVAL, i);
        continue;
      }
      const uint32_t pos = goff[k] + atomicAdd(&cnt[k], 1u);
      int4* rec = reinterpret_cast<int4*>(a.recs + (size_t)pos * T::NI);
      int w[T::NI];
#pragma unroll
      for (int q = 0; q < T::NI; ++q) w[q] = 0;
      w[0] = i, w[1] = x, w[2] = kp_flags(L, is_set, mask, seen);
#pragma unroll
      for (int r = 0; r < N; ++r) w[3 + r] = rkq[u][r];
#pragma unroll
      for (int q = 0; q < T::NI / 4; ++q) rec[q] = make_int4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    }
  }
  // the fingerprints: wavefront sums, then one 64-bit atomic per workgroup and word
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 2 * (N + 1); ++q) {
    unsigned long long v = f[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) fsum[wv][q] = v;
  }
  __syncthreads();
  if (threadIdx.x < 2 * (N + 1))
    atomicAdd(&a.fp[threadIdx.x], fsum[0][threadIdx.x] + fsum[1][threadIdx.x] + fsum[2][threadIdx.x] + fsum[3][threadIdx.x]);
}

// LOG: the command log is kept (its own instantiation: the extra pass costs registers the plain tick must not pay for)
template <int N, bool LOG>
__global__ void __launch_bounds__(KpTile<N>::THREADS) k_epx_key2(const EpxState st, const EpxBatch b, const KpArgs a) {
  using T = KpTile<N>;
  extern __shared__ __align__(16) unsigned char kp_smem[];
  int* recs = reinterpret_cast<int*>(kp_smem);                                 // [NI][TC] field-major: i, number, flags, rank[N]
#define RF(f, sl) recs[(f) * T::TC + (sl)]
  unsigned char* region = reinterpret_cast<unsigned char*>(recs + (size_t)T::TC * T::NI);
  uint32_t* sortA = reinterpret_cast<uint32_t*>(region);                        // [N][TC]
  uint32_t* sortB = sortA + (size_t)N * T::TC;                                  // [N][TC]
  uint32_t* bk = sortB + (size_t)N * T::TC;                                     // [N][NBK] rank buckets
  int* rows = reinterpret_cast<int*>(region);                                   // [TC][N - 1][N] (after the sort)
  uint32_t* rcnt = reinterpret_cast<uint32_t*>(region + T::R_BYTES);            // [N][W][128]
  uint32_t* rcur = rcnt + N * T::W * KP_RADIX;                                  // [N][W][128]
  int* tot = reinterpret_cast<int*>(rcur + N * T::W * KP_RADIX);                // [N][W][2N]
  int* cntr = tot + N * T::W * 2 * N;                                           // [N] participants of replica r
  int* rmin = cntr + N;                                                         // [N] smallest / largest rank among them
  int* rmax = rmin + N;
  int* degenerate = rmax + N;                                                   // a rank bucket is too full: radix sort
  int* base = degenerate + 1;                                                   // [N][2N] the replicas' TopOne vectors of the key
  if (st.status[0] != 0) return;
  if (a.ctl[1] != 0) return;  // a key does not fit the tables: the host sends the whole tick the first form's way
  // a rank row is a permutation of 0..m-1 iff (values are in range, checked by k_kp_scatter, and) its multiset of
  // values is that of the indices: compared through two additive 64-bit fingerprints of independently mixed values
  {
    bool bad = false;
    for (int r = 0; r < N; ++r) bad = bad || a.fp[2 + 2 * r] != a.fp[0] || a.fp[3 + 2 * r] != a.fp[1];
    if (bad) {
      if (threadIdx.x == 0 && blockIdx.x == 0) epx_report(st.status, FPX_EINVAL, -1);
      return;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = wave / T::W, w = wave - r * T::W;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int rank_bits = 1;
  while ((1 << rank_bits) < a.m) ++rank_bits;
#ifdef KP_X_NOSORT
  const int passes = 0;
#else
  const int passes = (rank_bits + KP_RADIX_BITS - 1) / KP_RADIX_BITS;
#endif

  struct Next {
    int lo, len;
    int4 q[T::RQ];
  };
  auto fetch = [&](int k, Next& s) {
    s.lo = a.seg[(size_t)k * 2], s.len = a.seg[(size_t)k * 2 + 1];
    const int4* src = reinterpret_cast<const int4*>(a.recs + (size_t)s.lo * T::NI);
    const int total = s.len * (T::NI / 4);
#pragma unroll
    for (int j = 0; j < T::RQ; ++j) {
      const int p = j * T::THREADS + threadIdx.x;
      s.q[j] = p < total ? src[p] : make_int4(0, 0, 0, 0);
    }
  };

  Next cur, nxt;
  int k = blockIdx.x;
  if (k >= st.num_keys) return;
  fetch(k, cur);
  for (; k < st.num_keys; k += gridDim.x) {
    const int c = cur.len;  // <= TC: the kernel does not run otherwise
    int carry_in = 0;
    // ---- the records, the counters
    {
      const int total = c * (T::NI / 4);
#pragma unroll
      for (int j = 0; j < T::RQ; ++j) {
        const int p = j * T::THREADS + threadIdx.x;
        if (p < total) {  // int4 p = fields 4 h .. 4 h + 3 of record sl; field-major in LDS (a record-major table is read
                          // with a stride of NI words: 8 lanes per bank)
          const int sl = p / (T::NI / 4), h = p - sl * (T::NI / 4);
          RF(4 * h, sl) = cur.q[j].x, RF(4 * h + 1, sl) = cur.q[j].y, RF(4 * h + 2, sl) = cur.q[j].z, RF(4 * h + 3, sl) = cur.q[j].w;
        }
      }
      for (int j = threadIdx.x; j < N * T::W * 2 * N + N; j += T::THREADS) tot[j] = 0;  // tot and cntr
      for (int j = threadIdx.x; j < N * T::NBK; j += T::THREADS) bk[j] = 0;
      if (threadIdx.x < N) rmin[threadIdx.x] = 0x7fffffff, rmax[threadIdx.x] = 0;
      if (threadIdx.x == 0) *degenerate = 0;
      // the replicas' TopOne vectors of the key (KeyValueStore.scala:229-230), the carries of the scans: requested
      // here, parked in LDS before the scans need them (held in registers across the key they spilled)
      if (w == 0 && lane < 2 * N) {
        const size_t ib = ((size_t)r * st.num_keys + k) * N;
        carry_in = lane < N ? st.gets[ib + lane] : st.sets[ib + lane - N];
      }
    }
    const int kn = k + gridDim.x;
    const bool more = kn < st.num_keys;
    __syncthreads();
    // ---- per replica: (rank << 11 | slot) of the commands it takes part in, compacted with one counter bump per
    // wavefront and 64 slots
    for (int base = w * 64; base < c; base += T::W * 64) {
      const int j = base + lane;
      bool part = false;
      uint32_t code = 0;
      if (j < c) {
        const int fl = RF(2, j);
        const int L = fl & 7;
        part = r == L || ((fl >> (16 + r)) & 1);
        code = ((uint32_t)RF(3 + r, j) << KP_SLOT_BITS) | (uint32_t)j;
      }
      const unsigned long long bal = __ballot(part);
      if (bal) {
        int at = 0;
        if (lane == 0) at = atomicAdd(&cntr[r], (int)__popcll(bal));
        at = __builtin_amdgcn_readfirstlane(at);
        if (part) sortA[(size_t)r * T::TC + at + (int)__popcll(bal & lt)] = code;
        int lo = part ? (int)(code >> KP_SLOT_BITS) : 0x7fffffff, hi = part ? (int)(code >> KP_SLOT_BITS) : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lo = min(lo, __shfl_xor(lo, o)), hi = imax(hi, __shfl_xor(hi, o));
        if (lane == 0) atomicMin(&rmin[r], lo), atomicMax(&rmax[r], hi);
      }
    }
    __syncthreads();
    const int cr = cntr[r];
    const int per = ((cr + T::W * 64 - 1) / (T::W * 64)) * 64;  // elements of one wavefront's part
    const int p0 = w * per, p1 = min(cr, p0 + per);
    // ---- the replica's words in rank order.  Bucket sort: the ranks of one key spread over [rmin, rmax]; NBK buckets
    // of equal width hold about one word each, a word's place is its bucket's start + the words of the bucket below
    // it (found by looking at them: buckets are tiny) -- 4 LDS round trips per word where an LSD radix sort of the
    // 21 rank bits takes 3 passes of ~90 instructions per 64 words.  Ranks that clump (a bucket with more than
    // KP_MAX_OCC words) send the key through the radix sort below instead: any order of ranks is sorted correctly.
    uint32_t* src = sortA + (size_t)r * T::TC;
    uint32_t* dst = sortB + (size_t)r * T::TC;
    uint32_t* mycnt = rcnt + (r * T::W + w) * KP_RADIX;
    uint32_t* mycur = rcur + (r * T::W + w) * KP_RADIX;
    int sort_passes = passes;
#if !defined(KP_X_RADIX_ONLY) && !defined(KP_X_NOSORT)
    {
      uint32_t* bkr = bk + r * T::NBK;
      const int lo = rmin[r];
      const unsigned span1 = cr > 0 ? (unsigned)(rmax[r] - lo) : 0u;  // span - 1
      constexpr int LOG_NBK = T::NBK == 1024 ? 10 : 9;
      const int sh = max(0, (32 - __clz((int)span1 | 1)) - LOG_NBK);   // (span - 1) >> sh < NBK
      uint32_t ec[T::CPW], eq[T::CPW], ea[T::CPW];
#pragma unroll
      for (int cc = 0; cc < T::CPW; ++cc) {
        const int p = p0 + cc * 64 + lane;
        ec[cc] = 0xffffffffu, eq[cc] = 0, ea[cc] = 0;
        if (p < p1) {
          ec[cc] = src[p];
          eq[cc] = (uint32_t)((int)(ec[cc] >> KP_SLOT_BITS) - lo) >> sh;
          ea[cc] = atomicAdd(&bkr[eq[cc]], 1u);
        }
      }
      __syncthreads();
      if (w == 0) {  // bucket counts -> bucket starts, by one wavefront per replica: NBK / 64 consecutive buckets per lane
        constexpr int PL = T::NBK / 64;
        uint32_t v[PL], sum = 0, big = 0;
#pragma unroll
        for (int j = 0; j < PL; ++j) v[j] = bkr[lane * PL + j], sum += v[j], big = max(big, v[j]);
        uint32_t run = kp_wave_excl_sum(sum);
#pragma unroll
        for (int j = 0; j < PL; ++j) bkr[lane * PL + j] = run, run += v[j];
        if (big > (uint32_t)KP_MAX_OCC) *degenerate = 1;
      }
      __syncthreads();
      if (*degenerate == 0) {
#pragma unroll
        for (int cc = 0; cc < T::CPW; ++cc)
          if (ec[cc] != 0xffffffffu) dst[bkr[eq[cc]] + ea[cc]] = ec[cc];
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < T::CPW; ++cc)
          if (ec[cc] != 0xffffffffu) {
            const uint32_t s0 = bkr[eq[cc]], s1 = eq[cc] + 1 < (uint32_t)T::NBK ? bkr[eq[cc] + 1] : (uint32_t)cr;
            uint32_t below = 0;
            for (uint32_t j = s0; j < s1; ++j) below += dst[j] < ec[cc] ? 1u : 0u;
            src[s0 + below] = ec[cc];
          }
        __syncthreads();
        sort_passes = 0;  // sorted, in sortA
      }
    }
#endif
    // ---- LSD radix sort of the replica's words on the rank bits, in LDS
    for (int pass = 0; pass < sort_passes; ++pass) {
      const int shift = KP_SLOT_BITS + pass * KP_RADIX_BITS;
      mycnt[lane] = 0, mycnt[lane + 64] = 0;
      __builtin_amdgcn_wave_barrier();
      for (int p = p0 + lane; p < p1; p += 64) atomicAdd(&mycnt[(src[p] >> shift) & (KP_RADIX - 1)], 1u);
      __syncthreads();
      {
        // lane owns digits 2 lane, 2 lane + 1: where they start in the replica's sequence, then where this
        // wavefront's elements of them go (after those of the wavefronts before it)
        uint32_t t0 = 0, t1 = 0, b0 = 0, b1 = 0;
        for (int w2 = 0; w2 < T::W; ++w2) {
          const uint32_t c0 = rcnt[(r * T::W + w2) * KP_RADIX + 2 * lane], c1 = rcnt[(r * T::W + w2) * KP_RADIX + 2 * lane + 1];
          if (w2 < w) b0 += c0, b1 += c1;
          t0 += c0, t1 += c1;
        }
        const uint32_t ex = kp_wave_excl_sum(t0 + t1);
        mycur[2 * lane] = ex + b0, mycur[2 * lane + 1] = ex + t0 + b1;
      }
      __builtin_amdgcn_wave_barrier();
      for (int pb = p0; pb < p1; pb += 64) {
        const int p = pb + lane;
        const bool valid = p < p1;
        const uint32_t code = valid ? src[p] : 0u;
        const uint32_t dg = (code >> shift) & (KP_RADIX - 1);
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < KP_RADIX_BITS; ++bit) {
          const bool on = (dg >> bit) & 1u;
          const unsigned long long bb = __ballot(on);
          peers &= on ? bb : ~bb;
        }
        const uint32_t before = __popcll(peers & lt);
        const uint32_t at = valid ? mycur[dg] : 0u;
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0) mycur[dg] = at + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        if (valid) dst[at + before] = code;
      }
      __syncthreads();
      uint32_t* t = src;
      src = dst, dst = t;
    }
    // ---- the sorted words of this wavefront's part -> registers (the buffers are reused below); the puts of the
    // part (per column the largest id + 1) are the carry of the parts behind it
    uint32_t code[T::CPW];
    int* mytot = tot + (r * T::W + w) * 2 * N;
#pragma unroll
    for (int cc = 0; cc < T::CPW; ++cc) {
      const int p = p0 + cc * 64 + lane;
      code[cc] = p < p1 ? src[p] : 0xffffffffu;
      if (p < p1) {
        const int sl = (int)(code[cc] & KP_SLOT_MASK);
        const int fl = RF(2, sl);
        atomicMax(&mytot[((fl >> 3) & 1) * N + (fl & 7)], RF(1, sl) + 1);
      }
    }
    if (w == 0 && lane < 2 * N) base[r * 2 * N + lane] = carry_in;
    __syncthreads();
    // ---- the segmented scans (the sort buffers are spent: their place takes the conflict rows).  Row n-2 of a command
    // is its leader's own conflicts D (the PreAccept's dependencies), rows 0 .. n-3 those of the replicas whose
    // answers the leader counts, in replica order
    {
      int cg[N], cs[N], ng[N], ns[N];
#pragma unroll
      for (int l = 0; l < N; ++l) cg[l] = base[r * 2 * N + l], cs[l] = base[r * 2 * N + N + l], ng[l] = 0, ns[l] = 0;
      for (int w2 = 0; w2 < w; ++w2) {
        const int* o = tot + (r * T::W + w2) * 2 * N;
#pragma unroll
        for (int l = 0; l < N; ++l) cg[l] = imax(cg[l], o[l]), cs[l] = imax(cs[l], o[N + l]);
      }
#pragma unroll
      for (int cc = 0; cc < T::CPW; ++cc) {
        const bool valid = code[cc] != 0xffffffffu;
        const int sl = valid ? (int)(code[cc] & KP_SLOT_MASK) : 0;
        const int fl = valid ? RF(2, sl) : 0;
        const int id1 = valid ? RF(1, sl) + 1 : 0;  // TopOne.put: max(.., id + 1), util/TopOne.scala:12-15
        const int L = fl & 7;
        int dep[N];
#ifndef KP_X_NOSCAN
        if (p0 + cc * 64 < p1)
#else
        if (p0 + cc * 64 < -1)
#endif
          scan_chunk<N>(valid, (fl >> 3) & 1, L, id1, cg, cs, ng, ns, dep);
        const unsigned resp = ((unsigned)fl >> 8) & 0xffu;
        if (valid && (L == r || ((resp >> r) & 1u))) {
          const int ri = L == r ? N - 2 : (int)__popc(resp & ((1u << r) - 1u));
#pragma unroll
          for (int l = 0; l < N; ++l) rows[sl * T::RSTR + ri * N + l] = dep[l];
        }
      }
    }
    if (more) fetch(kn, nxt);  // the next key's loads fly while this one is decided
    __syncthreads();
    // ---- handlePreAcceptOk (Replica.scala:1291-1419): every counted answer is local conflicts U the PreAccept's
    // dependencies (handlePreAccept :1257-1262); fast path iff the n-2 answers are identical (popularItems); the union
    // the slow path proposes (preAcceptingSlowPath :796-813) is their column-wise max, which is the agreed row as well
    constexpr bool COOP = (N - 1) * N >= 2 * N + 6;  // a command's conflict rows can hold its packed line
    constexpr bool CLOG = LOG && (N - 1) * N >= 2 * N + 8;  // ... and what the command-log pass needs behind it (n >= 5)
    for (int sl = threadIdx.x; sl < c; sl += T::THREADS) {
      const int i = RF(0, sl), x = RF(1, sl), L = RF(2, sl) & 7;
      bool fast = true;
      int od[N], ol[N], oe0 = 0, oe1 = 0, raw_hi = 0, raw_d = 0;
      const int* row = rows + (size_t)sl * T::RSTR;
#pragma unroll
      for (int l = 0; l < N; ++l) {
        const int dl = row[(N - 2) * N + l];
        int hi = imax(row[l], dl);
#pragma unroll
        for (int q = 1; q < N - 2; ++q) {
          const int v = imax(row[q * N + l], dl);
          fast = fast && v == hi;
          hi = imax(hi, v);
        }
        od[l] = hi, ol[l] = dl;
        if (l == L) raw_hi = hi, raw_d = dl, own_column(hi, x, &od[l], &oe0), own_column(dl, x, &ol[l], &oe1);
      }
#ifdef KP_X_NOOUT
      if (fast && x == -12345) b.fast[i] = 1;
      continue;
#endif
      if (a.packed) {
        // the packed line takes the place of the command's conflict rows (read above) and leaves below, four lanes
        // per 64-byte line: a store instruction then covers 16 whole lines instead of 16 bytes of 64 different ones
        int line[2 * N + 6];
        int* o = COOP ? rows + (size_t)sl * T::RSTR : line;  // (n = 3: the rows are too short, the thread stores its line)
#pragma unroll
        for (int l = 0; l < N; ++l) o[l] = od[l], o[N + l] = ol[l];
        o[2 * N] = oe0, o[2 * N + 1] = oe1, o[2 * N + 2] = fast ? 1 : 0;
#pragma unroll
        for (int l = 2 * N + 3; l < 2 * N + 6; ++l) o[l] = 0;
        if constexpr (!COOP) {
          int4* out = reinterpret_cast<int4*>(a.packed + (size_t)i * a.stride);
#pragma unroll
          for (int q = 0; q < (2 * N + 6) / 4; ++q) out[q] = make_int4(line[4 * q], line[4 * q + 1], line[4 * q + 2], line[4 * q + 3]);
        }
      } else {
        if (b.fast) b.fast[i] = fast ? 1 : 0;
        if (b.own_values_end) *reinterpret_cast<int2*>(b.own_values_end + (size_t)i * 2) = make_int2(oe0, oe1);
        // the rows leave below as whole n-int lines: the command's conflict rows are spent, rows 0 and 1 take them
#pragma unroll
        for (int l = 0; l < N; ++l) rows[sl * T::RSTR + l] = od[l], rows[sl * T::RSTR + N + l] = ol[l];
      }
      if constexpr (CLOG) {
        if (st.num_instances > 0) {  // what the command-log pass below needs of the decision, behind the packed line
          int* o = rows + (size_t)sl * T::RSTR;
          o[2 * N] = oe0, o[2 * N + 1] = oe1, o[2 * N + 2] = fast ? 1 : 0, o[2 * N + 6] = raw_hi, o[2 * N + 7] = raw_d;
        }
      }
    }
    if (COOP && a.packed) {
      __syncthreads();
      constexpr int Q = (2 * N + 3 + 3) / 4;  // int4's of a line: 3, 4, 5 for n = 3, 5, 7
      for (int t = threadIdx.x; t < c * Q; t += T::THREADS) {
        const int sl = t / Q, q = t - sl * Q;
        const int* o = rows + (size_t)sl * T::RSTR + 4 * q;
        reinterpret_cast<int4*>(a.packed + (size_t)RF(0, sl) * a.stride)[q] = make_int4(o[0], o[1], o[2], o[3]);
      }
    }
    if (!a.packed) {
      __syncthreads();
      for (int t = threadIdx.x; t < c * N; t += T::THREADS) {
        const int sl = t / N, l = t - sl * N;
        const size_t o = (size_t)RF(0, sl) * N + l;
        if (b.deps) b.deps[o] = rows[sl * T::RSTR + l];
        if (b.leader_deps) b.leader_deps[o] = rows[sl * T::RSTR + N + l];
      }
    }
    // ---- the command log (num_instances > 0; Replica.scala:688-696, 815-823, 1259-1271): a fast-path commit is a
    // CommittedEntry with the agreed dependencies at EVERY replica; otherwise every replica that processed the PreAccept
    // holds PreAcceptedEntry(Ballot(0, leader), Ballot(0, leader), triple) with what IT answered (its conflicts U the
    // PreAccept's; the leader: what it proposed).  The answers of the replicas that are not counted were never kept: the
    // scans run once more (registers only), now knowing every command's decision.
    if constexpr (CLOG) {
      if (st.num_instances > 0) {
        int cg[N], cs[N], ng[N], ns[N];
#pragma unroll
        for (int l = 0; l < N; ++l) cg[l] = base[r * 2 * N + l], cs[l] = base[r * 2 * N + N + l], ng[l] = 0, ns[l] = 0;
        for (int w2 = 0; w2 < w; ++w2) {
          const int* o = tot + (r * T::W + w2) * 2 * N;
#pragma unroll
          for (int l = 0; l < N; ++l) cg[l] = imax(cg[l], o[l]), cs[l] = imax(cs[l], o[N + l]);
        }
#pragma unroll
        for (int cc = 0; cc < T::CPW; ++cc) {
          const bool valid = code[cc] != 0xffffffffu;
          const int sl = valid ? (int)(code[cc] & KP_SLOT_MASK) : 0;
          const int fl = valid ? RF(2, sl) : 0;
          const int x = valid ? RF(1, sl) : 0;
          const int L = fl & 7;
          int dep[N];
          if (p0 + cc * 64 < p1) scan_chunk<N>(valid, (fl >> 3) & 1, L, x + 1, cg, cs, ng, ns, dep);
          if (!valid) continue;
          const unsigned seen = (((unsigned)fl >> 16) & 0xffu) | (1u << L);
          const int* o = rows + (size_t)sl * T::RSTR;
          const bool fast = o[2 * N + 2] != 0;
          const int tr = b.triple ? b.triple[RF(0, sl)] : -1;
          int t[N], end = 0;
          if (fast) {
#pragma unroll
            for (int l = 0; l < N; ++l) t[l] = o[l];
            end = o[2 * N];
          } else if (r == L) {
#pragma unroll
            for (int l = 0; l < N; ++l) t[l] = o[N + l];
            end = o[2 * N + 1];
          } else {
#pragma unroll
            for (int l = 0; l < N; ++l) {
              const int v = imax(dep[l], l == L ? o[2 * N + 7] : o[N + l]);
              t[l] = v;
              if (l == L) own_column(v, x, &t[l], &end);
            }
          }
          for (int rr = 0; rr < N; ++rr) {
            // my own entry; the leader's lane also writes those of the replicas that never saw a fast-path commit's PreAccept
            if (!(rr == r || (fast && r == L && !((seen >> rr) & 1u)))) continue;
            const size_t e = ((size_t)rr * N + L) * st.num_instances + x;
#pragma unroll
            for (int l = 0; l < N; ++l) st.cl_deps[e * N + l] = t[l];
            st.cl_dend[e] = end;
            st.cl_status[e] = fast ? CL_COMMITTED : CL_PRE_ACCEPTED;
            st.cl_ballot[e] = fast ? -1 : L, st.cl_vote[e] = fast ? -1 : L, st.cl_triple[e] = tr;  // Ballot(0, L) = 0 * 8 + L
          }
        }
      }
    }
    // ---- commit -> updateConflictIndex at every replica (Replica.scala:815-828): the key's watermarks learn every
    // instance of the tick (each command was scanned by its leader's replica: the max over replicas is the tick)
    if (w == 0 && lane < 2 * N) {
      int v = 0;
      for (int j = 0; j < N * T::W; ++j) v = imax(v, tot[j * 2 * N + lane]);
      const size_t ib = ((size_t)r * st.num_keys + k) * N;
      int32_t* p = lane < N ? &st.gets[ib + lane] : &st.sets[ib + lane - N];
      if (v > base[r * 2 * N + lane]) *p = v;
    }
    __syncthreads();  // the tables are reused by the next key
    cur = nxt;
  }
}
#undef RF
