// fpx_epaxos_kp.hpp -- K5, second form: the tick is partitioned by KEY once, each key is ordered per replica ON CHIP.
// Included by fpx_epaxos.hip inside its anonymous namespace (uses EpxState, EpxBatch, own_column, wave_incl_max, ...).
//
// The first form (k_epx_keys -> radix sort of n x m (key, message) pairs -> k_epx_key) moves every command n times
// through HBM as an 8-byte pair, twice each way, and gathers its fields again by message index afterwards: 651 MB
// per 2^20-command tick for 84 B of compulsory traffic per command (profiles/r02_k5_pmc.md).  Here a command
// crosses HBM once more than it has to, as ONE record that holds everything the per-key work needs:
//
//   k_kp_hist      8 tiles of 2048 messages per workgroup: how many of each key; every key of the group CLAIMS its run of
//                  records in the key's segment with one returning atomic (no order is needed inside a key, so no
//                  scan over tiles: round 3's k_kp_scan and its table of per-tile counts in key order are gone); a
//                  claim that runs past the segment (more commands of one key than the on-chip tables take) is
//                  reported: such a tick goes the first form's way, nothing of it is applied here
//   k_kp_scatter   validates the tick, packs record(i) = {i | header, number, rank[0..n) packed}, stages the tile's records
//                  in LDS in key order and writes every key's run to its place (the tile's claimed run) from consecutive
//                  lanes, accumulates the fingerprints that tell a permutation from a non-permutation, and tells the
//                  host through a page-locked word whether a claim overflowed
//   k_epx_key2<N>  one persistent workgroup per CU, key after key, the next key's records in flight: a thread unpacks a
//                  record into the per-replica sort words (rank << 11 | slot; ~0 where the replica takes no part) IN
//                  PLACE of a compaction pass; the words are ordered by a bucket sort in LDS (LSD radix sort for
//                  clumped ranks); the segmented scans run on the sorted order; conflict rows meet in LDS
//                  [command][n-1][n]; one thread per command decides; decisions leave as one packed line per command
//                  (or the four arrays of the first form); the key's conflict index is updated in place.
//
// HBM traffic per command (n = 5): 35 B of inputs + 24 B record out + 24 B record in + the outputs.
// With a command log (num_instances > 0, n >= 5): k_epx_key2<N, true> scans a second time after the decisions and
// writes the entries of every replica that saw the PreAccept (or of every replica, for a fast-path commit).
#pragma once

#ifndef FPX_K5_NT_OUT
#define FPX_K5_NT_OUT 1  // the packed lines leave k_epx_key2 as nontemporal stores: written once, read by nobody on the device --
                         // 0.1057 -> 0.1030 ms per tick when no buffer is touched twice (profiles/r06_k5.md; the tick's inputs as
                         // nontemporal loads on top of it: 0.1033, not kept)
#endif

constexpr int KP_TILE = 2048;   // messages per partition tile
constexpr int KP_MAXB = 2048;   // keys (one LDS counter each in the partition passes)
constexpr int KP_SLOT_BITS = 11;
constexpr uint32_t KP_SLOT_MASK = (1u << KP_SLOT_BITS) - 1u;
constexpr int KP_RADIX_BITS = 7, KP_RADIX = 1 << KP_RADIX_BITS;
constexpr int KP_MAX_OCC = 24;           // fullest rank bucket the bucket sort accepts before the key is radix-sorted instead
constexpr int KP_TOT_STRIDE = 16;        // words between two keys' claim counters: a 64-byte sector each (the claims of
                                         // 1024 keys are returning atomics on 1024 different sectors, not on 32 lines)
constexpr int KP_IDX_BITS = 21;          // message indices and ranks: m < 2^21
constexpr uint32_t KP_IDX_MASK = (1u << KP_IDX_BITS) - 1u;
constexpr uint32_t KP_INVALID = 0xffffffffu;  // sort word of a command the replica takes no part in (sorts last)

template <int N> struct KpTile {
  // words per record.  n <= 5: {i | header << 21, number, ranks packed 21 bits each} -- 16 B (n = 3), 24 B (n = 5);
  // n = 7: {i, number, flags, ranks packed} -- 32 B.  Round 3's record was {i, number, flags, rank[n], padding}: 32 / 48 B
  static constexpr int NI = N <= 3 ? 4 : N <= 5 ? 6 : 8;
  static constexpr int TC = N <= 3 ? 1536 : N <= 5 ? 1152 : 640;   // commands of one key held on chip (<= 2^11)
  static constexpr int W = N <= 3 ? 4 : N <= 5 ? 3 : 2;            // wavefronts per replica
  static constexpr int THREADS = 64 * N * W;
  static constexpr int MAXC = (TC + 63) / 64;
  static constexpr int CPW = (MAXC + W - 1) / W;                   // 64-command chunks per wavefront
  static constexpr int NBK = N <= 5 ? 1024 : 512;                  // rank buckets of the bucket sort
  static constexpr int RQ = (TC + THREADS - 1) / THREADS;          // records per thread
  // LDS: i, number, flags [3][TC] | region R | part totals | misc.  R holds the two sort buffers and the rank buckets
  // (the radix sort's counters and cursors, when a key's ranks clump) while a key is sorted, then the conflict rows
  // [TC][N - 1][N]: the n-2 counted answers and the leader's own
  static constexpr size_t META_BYTES = (size_t)3 * TC * 4;
  static constexpr size_t SORT_BYTES = (size_t)2 * N * TC * 4 + (size_t)N * NBK * 4;
  static constexpr int RSTR = (N - 1) * N + 1;  // ints between two commands' conflict rows: odd, so that random commands
                                                // spread over all LDS banks ((N - 1) N = 20 reaches 16 of 64: 40 % of the
                                                // kernel's LDS cycles were bank conflicts)
  static constexpr size_t ROWS_BYTES = (size_t)TC * RSTR * 4;
  static constexpr size_t R_BYTES = SORT_BYTES > ROWS_BYTES ? SORT_BYTES : ROWS_BYTES;
  static constexpr size_t BYTES = META_BYTES + R_BYTES + (size_t)N * W * 2 * N * 4 + 256 + (size_t)N * 2 * N * 4;
  static_assert(BYTES <= 160 * 1024, "LDS of one CU");
  static_assert((size_t)2 * N * W * KP_RADIX * 4 <= (size_t)N * NBK * 4, "the radix sort's counters take the bucket table's place");
  static_assert(TC <= (1 << KP_SLOT_BITS), "slots share the sort word with the rank");
};

// flags of a command as the key kernel keeps them: leader | is_set << 3 | resp_mask << 8 | seen_mask << 16
__device__ __forceinline__ int kp_flags(int L, int is_set, unsigned resp, unsigned seen) {
  return L | (is_set ? 8 : 0) | (int)(resp << 8) | (int)(seen << 16);
}

// The record.  resp_mask is n - 2 of the n - 1 replicas that are not the leader: the header names the ONE that is left
// out (its index among the non-leaders), which with leader, is_set and seen_mask fits beside a 21-bit message index for
// n <= 5.  Ranks are < m < 2^21: two ranks' words carry a third rank's halves in their upper 11 bits.
template <int N>
__device__ __forceinline__ void kp_pack(uint32_t* w, int i, int x, int L, int is_set, unsigned resp, unsigned seen, const int* rk) {
  const unsigned full = (1u << N) - 1u;
  const int q = __ffs((int)(full & ~resp & ~(1u << L))) - 1;  // the non-leader whose answer is not waited for
  const unsigned e = (unsigned)(q - (q > L ? 1 : 0));
  const unsigned hdr = (unsigned)L | (is_set ? 8u : 0u) | (seen << 4) | (e << (4 + N));
  auto pair_lo = [](int a, int c) { return (uint32_t)a | (((uint32_t)c & 0x7ffu) << KP_IDX_BITS); };
  auto pair_hi = [](int a, int c) { return (uint32_t)a | (((uint32_t)c >> 11) << KP_IDX_BITS); };
  if constexpr (N <= 3) {
    w[0] = (uint32_t)i | (hdr << KP_IDX_BITS), w[1] = (uint32_t)x;
    w[2] = pair_lo(rk[0], rk[2]), w[3] = pair_hi(rk[1], rk[2]);
  } else if constexpr (N <= 5) {
    w[0] = (uint32_t)i | (hdr << KP_IDX_BITS), w[1] = (uint32_t)x;
    w[2] = pair_lo(rk[0], rk[4]), w[3] = pair_hi(rk[1], rk[4]), w[4] = (uint32_t)rk[2], w[5] = (uint32_t)rk[3];
  } else {
    w[0] = (uint32_t)i, w[1] = (uint32_t)x, w[2] = (uint32_t)kp_flags(L, is_set, resp, seen);
    w[3] = pair_lo(rk[0], rk[4]), w[4] = pair_hi(rk[1], rk[4]);
    w[5] = pair_lo(rk[2], rk[5]), w[6] = pair_hi(rk[3], rk[5]), w[7] = (uint32_t)rk[6];
  }
}

template <int N>
__device__ __forceinline__ void kp_unpack(const uint32_t* w, int* i, int* x, int* flags, int* rk) {
  auto third = [](uint32_t a, uint32_t c) { return (int)((a >> KP_IDX_BITS) | ((c >> KP_IDX_BITS) << 11)); };
  if constexpr (N <= 5) {
    *i = (int)(w[0] & KP_IDX_MASK), *x = (int)w[1];
    const unsigned hdr = w[0] >> KP_IDX_BITS, full = (1u << N) - 1u;
    const int L = (int)(hdr & 7u);
    const unsigned seen = (hdr >> 4) & full, e = hdr >> (4 + N);
    const unsigned q = e + (e >= (unsigned)L ? 1u : 0u);
    *flags = kp_flags(L, (int)((hdr >> 3) & 1u), full & ~(1u << L) & ~(1u << q), seen);
    rk[0] = (int)(w[2] & KP_IDX_MASK), rk[1] = (int)(w[3] & KP_IDX_MASK);
    if constexpr (N <= 3) {
      rk[2] = third(w[2], w[3]);
    } else {
      rk[2] = (int)w[4], rk[3] = (int)w[5], rk[4] = third(w[2], w[3]);
    }
  } else {
    *i = (int)w[0], *x = (int)w[1], *flags = (int)w[2];
    rk[0] = (int)(w[3] & KP_IDX_MASK), rk[1] = (int)(w[4] & KP_IDX_MASK), rk[4] = third(w[3], w[4]);
    rk[2] = (int)(w[5] & KP_IDX_MASK), rk[3] = (int)(w[6] & KP_IDX_MASK), rk[5] = third(w[5], w[6]);
    rk[6] = (int)w[7];
  }
}

struct KpArgs {
  int m, tiles, B;                  // B = num_keys
  int groups;                       // workgroups of k_kp_hist (KP_HG tiles each)
  uint16_t* hist;                   // [tiles][B] where the tile's records of the key start within the key's segment
  uint32_t* tot;                    // [B * KP_TOT_STRIDE] records claimed in the key's segment; zero between ticks (the key
                                    // kernel clears what it read)
  uint32_t* big;                    // [groups] 1 = a claim of this group of tiles ran past its key's segment
  uint32_t* ctl;                    // [1] groups with such a claim (written by k_kp_scatter, read by the key kernel)
  uint32_t* bad;                    // [2] {seq of the last tick a partition kernel rejected, index of the message}: the
                                    // partition of tick t + 1 may run beside the key kernel of tick t, so its verdict
                                    // must not land in the context's status words before tick t + 1's own turn
  unsigned long long* fp;           // [2 (N + 1)] additive fingerprints: of the indices 0..m-1, then of every rank row
  uint32_t* recs;                   // [B][tc][NI]: a key's segment has room for what the on-chip tables take
  volatile uint32_t* host_flag;     // page-locked: [1] = ctl[1], then [0] = seq
  uint32_t seq;
  int tc;                           // KpTile<N>::TC
  int32_t* packed;                  // [m][stride] or null
  int stride;
  int vec;                          // every input array starts at a 16-byte boundary and m % 4 == 0: a thread's four (sixteen) messages
                                    // are neighbours and arrive as 16-byte loads
};

// the partition kernels' verdict on the tick: first reporter wins (seq grows from tick to tick)
__device__ __forceinline__ void kp_reject(const KpArgs& a, int index) {
  if (atomicMax(&a.bad[0], a.seq) < a.seq) a.bad[1] = (uint32_t)index;
}

// two independent 32-bit mixes of one value (murmur3's finaliser on differently salted inputs): the additive
// fingerprints are sums of these in 64-bit accumulators.  (Round 3 summed two splitmix64 finalisers: four 64-bit
// multiplies per value, a third of the scatter kernel's issue slots.)
__device__ __forceinline__ uint32_t kp_mix32(uint32_t h) {
  h ^= h >> 16, h *= 0x85EBCA6Bu;
  h ^= h >> 13, h *= 0xC2B2AE35u;
  return h ^ (h >> 16);
}
__device__ __forceinline__ void kp_fp_add(unsigned long long* f, uint32_t v) {
  f[0] += kp_mix32(v + 0x9E3779B9u), f[1] += kp_mix32(v ^ 0x7F4A7C15u) ^ 0x5BD1E995u;
}

// wave64 exclusive sum of one value per lane, on the DPP network (four row_shr steps inside each row of 16, row_bcast:15 /
// row_bcast:31 carry the row totals on; lanes without a source add the 0 of `old`).  Six __shfl_up steps were six dependent
// trips through the LDS crossbar: the one wavefront per replica that turns bucket counts into bucket starts waited 0.25 us
// of the phase's 0.76 for them.
__device__ __forceinline__ uint32_t kp_wave_excl_sum(uint32_t v) {
  int inc = (int)v;
  inc += dpp0<0x111, 0xF>(inc);
  inc += dpp0<0x112, 0xF>(inc);
  inc += dpp0<0x114, 0xF>(inc);
  inc += dpp0<0x118, 0xF>(inc);
  inc += dpp0<0x142, 0xA>(inc);
  inc += dpp0<0x143, 0xC>(inc);
  return (uint32_t)inc - v;
}

// wave64 sum of one value per lane on the same network: the total arrives in lane 63
__device__ __forceinline__ uint32_t kp_wave_sum_to_lane63(uint32_t v) {
  int inc = (int)v;
  inc += dpp0<0x111, 0xF>(inc);
  inc += dpp0<0x112, 0xF>(inc);
  inc += dpp0<0x114, 0xF>(inc);
  inc += dpp0<0x118, 0xF>(inc);
  inc += dpp0<0x142, 0xA>(inc);
  inc += dpp0<0x143, 0xC>(inc);
  return (uint32_t)inc;
}

// KP_HG tiles per workgroup (128 threads each).  After the count, thread j owns key j: the key's commands in the
// workgroup's tiles are ONE claim in the key's segment (so the records of 8 neighbouring tiles are neighbours there, and
// the workgroups of one XCD take neighbouring tiles in k_kp_scatter: their 24-byte stores meet in one L2 and leave it as
// whole lines), split among the tiles in tile order.
constexpr int KP_HG = 8;
__global__ void __launch_bounds__(128 * KP_HG) k_kp_hist(const EpxState st, const EpxBatch b, const KpArgs a) {
  extern __shared__ uint32_t kp_h[];  // [KP_HG][B]
  if (blockIdx.x == 0 && threadIdx.x < 2 * (st.n + 1)) a.fp[threadIdx.x] = 0ull;
  for (int j = threadIdx.x; j < KP_HG * a.B; j += 128 * KP_HG) kp_h[j] = 0;
  __syncthreads();
  const int sub = threadIdx.x >> 7, t = threadIdx.x & 127;
  const int tile = blockIdx.x * KP_HG + sub;
  uint32_t* h = kp_h + sub * a.B;
  const int first = tile * KP_TILE;
  constexpr int KT = KP_TILE / 128;  // a thread's keys: neighbours in the tick
  const int i0 = first + KT * t;
  int k[KT];
  if (a.vec && tile < a.tiles && i0 < a.m) {
    static_assert(KT % 4 == 0, "16-byte loads");
#pragma unroll
    for (int j = 0; j < KT; j += 4) {
      const bool in = i0 + j < a.m;  // m % 4 == 0: four at a time are inside or outside
      const int4 v = in ? *reinterpret_cast<const int4*>(b.key + i0 + j) : make_int4(-1, -1, -1, -1);
      k[j] = v.x, k[j + 1] = v.y, k[j + 2] = v.z, k[j + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < KT; ++j) k[j] = (tile < a.tiles && i0 + j < a.m) ? b.key[i0 + j] : -1;
  }
#pragma unroll
  for (int j = 0; j < KT; ++j) {
    const int i = i0 + j;
    if (tile < a.tiles && i < a.m) {
      if (k[j] < 0 || k[j] >= a.B) kp_reject(a, i);
      else atomicAdd(&h[k[j]], 1u);
    }
  }
  __syncthreads();
  const int t0 = blockIdx.x * KP_HG, nt = min(KP_HG, a.tiles - t0);
  int over = 0;
  for (int j = threadIdx.x; j < a.B; j += 128 * KP_HG) {
    uint32_t c[KP_HG], total = 0;
#pragma unroll
    for (int q = 0; q < KP_HG; ++q) c[q] = kp_h[q * a.B + j], total += c[q];
    uint32_t run = total ? atomicAdd(&a.tot[(size_t)j * KP_TOT_STRIDE], total) : 0u;
    if (run + total > (uint32_t)a.tc) over = 1;
#pragma unroll
    for (int q = 0; q < KP_HG; ++q)
      if (q < nt) a.hist[(size_t)(t0 + q) * a.B + j] = (uint16_t)min(run, 0xffffu), run += c[q];
  }
  over = __syncthreads_or(over);
  if (threadIdx.x == 0) a.big[blockIdx.x] = over ? 1u : 0u;
}

// The scatter workgroup: KpScat<N>::TILES neighbouring tiles of one k_kp_hist group (whose claimed runs are neighbours in
// every key's segment, in tile order: the workgroup's commands of a key are ONE run starting where its first tile's
// does).  Validation of one message exactly as k_epx_keys; the records are STAGED in LDS in key order and leave as
// 8-byte words from consecutive lanes, so a key's run crosses the memory pipeline as requests of whole sectors instead of
// a 16- and an 8-byte store per record at 2 M random places (as rounds 3 - 4 had it: 55 MB written for 25 MB of records;
// staged: 27.5 MB).  One tile per workgroup (two of them fit a CU): with two tiles the runs are twice as long, but ONE
// 1024-thread workgroup per CU loads, counts, stages and stores in lockstep with all the others -- measured 3 us slower
// per tick (profiles/r04_k5.md).
template <int N> struct KpScat {
  static constexpr int TILES = 1;
  static constexpr int SW = TILES * KP_TILE;           // records of one workgroup
  static constexpr int THREADS = SW / 4;               // four messages per thread, all their loads in flight at once
  static constexpr int NI = KpTile<N>::NI;
  static constexpr size_t BYTES = ((size_t)2 * KP_MAXB + SW + (size_t)SW * NI) * 4;
  static_assert(KP_HG % TILES == 0, "a workgroup's tiles belong to one k_kp_hist group");
  static_assert(KP_MAXB % THREADS == 0 || THREADS % KP_MAXB == 0, "keys per thread in the scan");
};

template <int N>
__global__ void __launch_bounds__(KpScat<N>::THREADS) k_kp_scatter(const EpxState st, const EpxBatch b, const KpArgs a) {
  using T = KpTile<N>;
  using S = KpScat<N>;
  extern __shared__ __align__(16) uint32_t kp_sc[];
  uint32_t* cnt = kp_sc;                    // [KP_MAXB] the workgroup's commands of the key; after the scan: where its run starts in `stage`
  uint32_t* goff = cnt + KP_MAXB;           // [KP_MAXB] the run's first record, in records from the start of a.recs
  uint32_t* gpos = goff + KP_MAXB;          // [SW] where the staged record goes (~0: nowhere, a claim ran past its key's segment)
  uint32_t* stage = gpos + S::SW;           // [SW][NI]
  __shared__ unsigned long long fsum[S::THREADS / 64][2 * (N + 1)];
  __shared__ uint32_t wsum[S::THREADS / 64];
  // workgroups go round-robin over the 8 XCDs: the ones of one XCD take consecutive tiles, whose records are neighbours
  // in every key's segment -- their stores meet in the same L2
  const int units = (a.tiles + S::TILES - 1) / S::TILES;
  const int ups = (units + 7) / 8, unit = ((int)blockIdx.x % 8) * ups + (int)blockIdx.x / 8;
  if (blockIdx.x == 0) {
    // did a claim run past a key's segment?  the key kernel reads ctl[0], the host the page-locked word
    uint32_t nbig = 0;
    for (int j = threadIdx.x; j < a.groups; j += S::THREADS) nbig += a.big[j];
    nbig = (uint32_t)__syncthreads_count(nbig != 0);
    if (threadIdx.x == 0) {
      a.ctl[0] = nbig;
      if (a.host_flag) {
        a.host_flag[1] = nbig;
        __threadfence_system();
        a.host_flag[0] = a.seq;
      }
    }
  }
  if ((int)blockIdx.x / 8 >= ups || unit >= units) return;
  const int tile = unit * S::TILES;
  for (int k = threadIdx.x; k < a.B; k += S::THREADS) cnt[k] = 0, goff[k] = (uint32_t)k * (uint32_t)a.tc + a.hist[(size_t)tile * a.B + k];
  for (int k = a.B + threadIdx.x; k < KP_MAXB; k += S::THREADS) cnt[k] = 0;
  __syncthreads();
  unsigned long long f[2 * (N + 1)];
#pragma unroll
  for (int q = 0; q < 2 * (N + 1); ++q) f[q] = 0ull;
  const int first = tile * KP_TILE;
  constexpr int MB = S::SW / S::THREADS;  // messages of one thread, neighbours in the tick: all their loads are in flight together
  static_assert(MB == 4, "a thread's messages are one 16-byte load per array");
  const int i0 = first + MB * (int)threadIdx.x;
  int Lq[MB], kq[MB], xq[MB], rkq[MB][N];
  unsigned mq[MB], sq[MB], tq[MB];
  if (a.vec && i0 < a.m) {  // (m % 4 == 0: all four are inside)
    auto ld4 = [&](const int32_t* p, int* o) {
      const int4 v = *reinterpret_cast<const int4*>(p + i0);
      o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w;
    };
    auto ld4b = [&](const uint8_t* p, unsigned* o) {
      const uint32_t v = *reinterpret_cast<const uint32_t*>(p + i0);
      o[0] = v & 0xffu, o[1] = (v >> 8) & 0xffu, o[2] = (v >> 16) & 0xffu, o[3] = v >> 24;
    };
    ld4(b.leader, Lq), ld4(b.key, kq), ld4(b.number, xq), ld4b(b.resp_mask, mq), ld4b(b.is_set, tq);
    if (b.seen_mask) ld4b(b.seen_mask, sq);
    else {
#pragma unroll
      for (int u = 0; u < MB; ++u) sq[u] = mq[u];
    }
#pragma unroll
    for (int r = 0; r < N; ++r) {
      int v[MB];
      ld4(b.rank + (size_t)r * a.m, v);
#pragma unroll
      for (int u = 0; u < MB; ++u) rkq[u][r] = v[u];
    }
  } else {
#pragma unroll
    for (int u = 0; u < MB; ++u) {
      const int i = i0 + u;
      const bool in = i < a.m;
      Lq[u] = in ? b.leader[i] : 0, kq[u] = in ? b.key[i] : 0, xq[u] = in ? b.number[i] : 0;
      mq[u] = in ? b.resp_mask[i] : 0u, tq[u] = in ? b.is_set[i] : 0u;
      sq[u] = in ? (b.seen_mask ? b.seen_mask[i] : mq[u]) : 0u;
#pragma unroll
      for (int r = 0; r < N; ++r) rkq[u][r] = in ? b.rank[(size_t)r * a.m + i] : 0;
    }
  }
  uint32_t at[MB];  // the command's number among the workgroup's commands of its key; ~0: no record
#pragma unroll
  for (int u = 0; u < MB; ++u) {
    const int i = i0 + u;
    at[u] = ~0u;
    if (i >= a.m) continue;
    const int L = Lq[u], k = kq[u], x = xq[u];
    const unsigned mask = mq[u], seen = sq[u];
    bool ok = L >= 0 && L < N && x >= 0 && k >= 0 && k < a.B;
    ok = ok && !((mask >> (ok ? L : 0)) & 1u) && (mask >> N) == 0 && (int)__popc(mask) == N - 2;
    ok = ok && (mask & ~seen) == 0 && !((seen >> (ok ? L : 0)) & 1u) && (seen >> N) == 0;
    kp_fp_add(f, (uint32_t)i);
#pragma unroll
    for (int r = 0; r < N; ++r) {
      ok = ok && rkq[u][r] >= 0 && rkq[u][r] < a.m;
      kp_fp_add(f + 2 + 2 * r, (uint32_t)rkq[u][r]);
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
      kp_reject(a, i);
      continue;
    }
    at[u] = atomicAdd(&cnt[k], 1u);
  }
  __syncthreads();
  // the runs' starts in the staging buffer: exclusive sums of the counts, KP_MAXB / THREADS keys per thread
  {
    constexpr int E = KP_MAXB >= S::THREADS ? KP_MAXB / S::THREADS : 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool mine = (int)threadIdx.x * E < KP_MAXB;
    uint32_t c[E], sum = 0;
#pragma unroll
    for (int j = 0; j < E; ++j) c[j] = mine ? cnt[threadIdx.x * E + j] : 0u, sum += c[j];
    uint32_t run = kp_wave_excl_sum(sum);
    if (lane == 63) wsum[wv] = run + sum;
    __syncthreads();
    for (int q = 0; q < wv; ++q) run += wsum[q];
#pragma unroll
    for (int j = 0; j < E; ++j)
      if (mine) cnt[threadIdx.x * E + j] = run, run += c[j];
  }
  __syncthreads();
  int staged = 0;
  for (int q = 0; q < S::THREADS / 64; ++q) staged += (int)wsum[q];
#pragma unroll
  for (int u = 0; u < MB; ++u) {
    if (at[u] == ~0u) continue;
    const int i = i0 + u;
    const int k = kq[u];
    const uint32_t p = cnt[k] + at[u], pos = goff[k] + at[u];
    gpos[p] = pos < (uint32_t)(k + 1) * (uint32_t)a.tc ? pos : ~0u;  // past the key's segment: the tick goes the first form's way
    uint32_t w[T::NI];
    kp_pack<N>(w, i, xq[u], Lq[u], tq[u] ? 1 : 0, mq[u], sq[u], rkq[u]);
    uint2* rec = reinterpret_cast<uint2*>(stage + (size_t)p * T::NI);
#pragma unroll
    for (int h = 0; h < T::NI / 2; ++h) rec[h] = make_uint2(w[2 * h], w[2 * h + 1]);
  }
  __syncthreads();
  {
    constexpr int H = T::NI / 2;  // 8-byte words of a record
    const uint2* src = reinterpret_cast<const uint2*>(stage);
    uint2* dst = reinterpret_cast<uint2*>(a.recs);
    for (int q = threadIdx.x; q < staged * H; q += S::THREADS) {
      const int p = q / H, h = q - p * H;
      const uint32_t g = gpos[p];
      if (g != ~0u) dst[(size_t)g * H + h] = src[q];
    }
  }
  // the fingerprints: wavefront sums, then one 64-bit atomic per workgroup and word.  A thread's word is the sum of at
  // most MB 32-bit values (< 2^34): its low 20 and its high 14 bits are added up separately on the DPP network (wavefront
  // sums < 2^26 and < 2^20) -- 12 instructions per word where a butterfly of 64-bit shuffles was 72 dependent trips through
  // the LDS crossbar per workgroup: 4.5 us at the end of every scatter workgroup (profiles/r04_k5.md, timelines)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  static_assert(MB <= 4, "a thread's fingerprint word fits 34 bits");
#pragma unroll
  for (int q = 0; q < 2 * (N + 1); ++q) {
    const uint32_t lo = kp_wave_sum_to_lane63((uint32_t)f[q] & 0xFFFFFu), hi = kp_wave_sum_to_lane63((uint32_t)(f[q] >> 20));
    if (lane == 63) fsum[wv][q] = (unsigned long long)lo + ((unsigned long long)hi << 20);
  }
  __syncthreads();
  if (threadIdx.x < 2 * (N + 1)) {
    unsigned long long v = 0;
    for (int q = 0; q < S::THREADS / 64; ++q) v += fsum[q][threadIdx.x];
    atomicAdd(&a.fp[threadIdx.x], v);
  }
}

// LOG: the command log is kept (its own instantiation: the extra pass costs registers the plain tick must not pay for)
template <int N, bool LOG>
__global__ void __launch_bounds__(KpTile<N>::THREADS) k_epx_key2(const EpxState st, const EpxBatch b, const KpArgs a) {
  using T = KpTile<N>;
  extern __shared__ __align__(16) unsigned char kp_smem[];
  int* recs = reinterpret_cast<int*>(kp_smem);                                 // [3][TC] field-major: i, number, flags
#define RF(f, sl) recs[(f) * T::TC + (sl)]
  unsigned char* region = kp_smem + T::META_BYTES;
  uint32_t* sortA = reinterpret_cast<uint32_t*>(region);                        // [N][TC]
  uint32_t* sortB = sortA + (size_t)N * T::TC;                                  // [N][TC]
  uint32_t* bk = sortB + (size_t)N * T::TC;                                     // [N][NBK] rank buckets
  int* rows = reinterpret_cast<int*>(region);                                   // [TC][N - 1][N] (after the sort)
  uint32_t* rcnt = bk;                                                          // [N][W][128] (radix sort: the buckets are spent)
  uint32_t* rcur = rcnt + N * T::W * KP_RADIX;                                  // [N][W][128]
  int* tot = reinterpret_cast<int*>(region + T::R_BYTES);                       // [N][W][2N] puts of a wavefront's part: gets, sets
  int* cntr = tot + N * T::W * 2 * N;                                           // [N] participants of replica r
  int* rmin = cntr + N;                                                         // [N] smallest / largest rank among them
  int* rmax = rmin + N;
  int* degenerate = rmax + N;                                                   // a rank bucket is too full: radix sort
  int* base = degenerate + 1;                                                   // [N][2N] the replicas' TopOne vectors of the key
  // a tick that is not applied (an error, a key beyond the tables, ranks that are no permutation) still has to leave
  // the claim counters at zero for the next one
  auto give_up = [&]() {
    for (int k = blockIdx.x * T::THREADS + threadIdx.x; k < a.B; k += gridDim.x * T::THREADS) a.tot[(size_t)k * KP_TOT_STRIDE] = 0;
  };
  if (st.status[0] != 0 || a.bad[0] == a.seq || a.ctl[0] != 0) {  // ctl[0]: a key does not fit the tables, the host sends the whole tick the first form's way
    if (st.status[0] == 0 && a.bad[0] == a.seq && threadIdx.x == 0 && blockIdx.x == 0) epx_report(st.status, FPX_EINVAL, (int)a.bad[1]);
    give_up();
    return;
  }
  // a rank row is a permutation of 0..m-1 iff (values are in range, checked by k_kp_scatter, and) its multiset of
  // values is that of the indices: compared through two additive 64-bit fingerprints of independently mixed values
  {
    bool bad = false;
    for (int r = 0; r < N; ++r) bad = bad || a.fp[2 + 2 * r] != a.fp[0] || a.fp[3 + 2 * r] != a.fp[1];
    if (bad) {
      if (threadIdx.x == 0 && blockIdx.x == 0) epx_report(st.status, FPX_EINVAL, -1);
      give_up();
      return;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = wave / T::W, w = wave - r * T::W;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int rank_bits = 1;
  while ((1 << rank_bits) <= a.m) ++rank_bits;  // 2^rank_bits > m: no rank is all ones in the bits the radix passes look at
  const int passes = (rank_bits + KP_RADIX_BITS - 1) / KP_RADIX_BITS;

  struct Next {
    int len;
    uint32_t q[T::RQ][T::NI];
  };
  auto fetch = [&](int k, Next& s) {
    s.len = min((int)a.tot[(size_t)k * KP_TOT_STRIDE], T::TC);
    const uint32_t* src = a.recs + (size_t)k * T::TC * T::NI;
#pragma unroll
    for (int j = 0; j < T::RQ; ++j) {
      const int sl = j * T::THREADS + threadIdx.x;
      if (sl < s.len) {
        if constexpr (T::NI == 6) {
#pragma unroll
          for (int h = 0; h < 3; ++h) {
            const uint2 v = reinterpret_cast<const uint2*>(src + (size_t)sl * 6)[h];
            s.q[j][2 * h] = v.x, s.q[j][2 * h + 1] = v.y;
          }
        } else {
#pragma unroll
          for (int h = 0; h < T::NI / 4; ++h) {
            const uint4 v = reinterpret_cast<const uint4*>(src + (size_t)sl * T::NI)[h];
            s.q[j][4 * h] = v.x, s.q[j][4 * h + 1] = v.y, s.q[j][4 * h + 2] = v.z, s.q[j][4 * h + 3] = v.w;
          }
        }
      }
    }
  };

  Next cur, nxt;
  int k = blockIdx.x;
  if (k >= st.num_keys) return;
  fetch(k, cur);
  __syncthreads();
  for (; k < st.num_keys; k += gridDim.x) {
    const int c = cur.len;  // <= TC
    const int cpad = (c + 63) & ~63;
    int carry_in = 0;
    // ---- the records: i, number, flags field-major; per replica the command's sort word rank << 11 | slot at the
    // command's slot, ~0 where the replica takes no part (they sort last and are not counted)
    {
#pragma unroll
      for (int j = 0; j < T::RQ; ++j) {
        const int sl = j * T::THREADS + threadIdx.x;
        if (sl < c) {
          int i, x, fl, rk[N];
          kp_unpack<N>(cur.q[j], &i, &x, &fl, rk);
          RF(0, sl) = i, RF(1, sl) = x, RF(2, sl) = fl;
          const unsigned part = (((unsigned)fl >> 16) & 0xffu) | (1u << (fl & 7));
#pragma unroll
          for (int q = 0; q < N; ++q) {
            const bool in = (part >> q) & 1u;
            sortA[(size_t)q * T::TC + sl] = in ? (((uint32_t)rk[q] << KP_SLOT_BITS) | (uint32_t)sl) : KP_INVALID;
          }
        } else if (sl < cpad) {
#pragma unroll
          for (int q = 0; q < N; ++q) sortA[(size_t)q * T::TC + sl] = KP_INVALID;
        }
      }
      for (int j = threadIdx.x; j < N * T::W * 2 * N; j += T::THREADS) tot[j] = 0;
      for (int j = threadIdx.x; j < N * T::NBK; j += T::THREADS) bk[j] = 0;
      if (threadIdx.x == 0) *degenerate = 0;
      if (threadIdx.x < N) rmin[threadIdx.x] = 0x7fffffff, rmax[threadIdx.x] = 0;  // (only a second sorting attempt uses them)
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
    // ---- the replica's words in rank order.  A wavefront owns a run of slots.  Bucket sort: the ranks of one key
    // spread over the tick's [0, m) (or, failing that, the key's own [rmin, rmax]); NBK buckets of equal width hold
    // about one word each, a word's place is its bucket's
    // start + the words of the bucket below it (found by looking at them: buckets are tiny) -- 4 LDS round trips per
    // word where an LSD radix sort of the 21 rank bits takes 3 passes of ~90 instructions per 64 words.  Ranks that
    // clump (a bucket with more than KP_MAX_OCC words) send the key through the radix sort below instead: any order of
    // ranks is sorted correctly.
    const int per = ((cpad / 64 + T::W - 1) / T::W) * 64;  // slots of one wavefront's run
    const int p0 = min(cpad, w * per), p1 = min(cpad, p0 + per);
    uint32_t* src = sortA + (size_t)r * T::TC;
    uint32_t* dst = sortB + (size_t)r * T::TC;
    uint32_t* mycnt = rcnt + (r * T::W + w) * KP_RADIX;
    uint32_t* mycur = rcur + (r * T::W + w) * KP_RADIX;
    int sort_passes = passes;
    // First attempt: buckets over the whole tick's ranks [0, m) -- a rank row is a permutation of them, and a key whose
    // commands arrive all through the tick spreads evenly there (no reduction over the key's ranks is needed: the
    // per-key minimum / maximum cost 6.7 us of the kernel's 76).  A key whose commands arrive in a burst fills a few
    // buckets only: second attempt over the key's own [rmin, rmax]; ranks that clump inside that as well go to the
    // radix sort.
#pragma nounroll
    for (int attempt = 0; attempt < 2 && sort_passes != 0; ++attempt) {
      uint32_t* bkr = bk + r * T::NBK;
      int lo = 0;
      unsigned span1 = (unsigned)max(a.m - 1, 0);  // span - 1
      if (attempt == 1) {
        // (everybody has seen *degenerate == 1 behind the barrier below; the bucket table holds the first attempt's starts)
        int mn = 0x7fffffff, mx = 0;
        for (int p = p0 + lane; p < p1; p += 64) {
          const uint32_t wd = src[p];
          if (wd != KP_INVALID) mn = min(mn, (int)(wd >> KP_SLOT_BITS)), mx = imax(mx, (int)(wd >> KP_SLOT_BITS));
        }
        mx = __builtin_amdgcn_readlane(wave_incl_max(mx), 63);
        mn = 0x7fffffff - __builtin_amdgcn_readlane(wave_incl_max(0x7fffffff - mn), 63);
        if (lane == 0 && mn <= mx) atomicMin(&rmin[r], mn), atomicMax(&rmax[r], mx);
        for (int j = threadIdx.x; j < N * T::NBK; j += T::THREADS) bk[j] = 0;
        __syncthreads();
        if (threadIdx.x == 0) *degenerate = 0;
        lo = rmin[r];
        span1 = rmax[r] >= lo ? (unsigned)(rmax[r] - lo) : 0u;
      }
      constexpr int LOG_NBK = T::NBK == 1024 ? 10 : 9;
      const int sh = max(0, (32 - __clz((int)span1 | 1)) - LOG_NBK);   // (span - 1) >> sh < NBK
      uint32_t ec[T::CPW], eq[T::CPW], ea[T::CPW];
#pragma unroll
      for (int cc = 0; cc < T::CPW; ++cc) {
        const int p = p0 + cc * 64 + lane;
        ec[cc] = p < p1 ? src[p] : KP_INVALID, eq[cc] = 0, ea[cc] = 0;
        if (ec[cc] != KP_INVALID) {
          eq[cc] = (uint32_t)((int)(ec[cc] >> KP_SLOT_BITS) - lo) >> sh;
          ea[cc] = atomicAdd(&bkr[eq[cc]], 1u);
        }
      }
      if (attempt == 0 && w == 0 && lane < 2 * N) base[r * 2 * N + lane] = carry_in;
      __syncthreads();
      if (w == 0) {  // bucket counts -> bucket starts, by one wavefront per replica: NBK / 64 consecutive buckets per lane
        constexpr int PL = T::NBK / 64;
        uint32_t v[PL], sum = 0, big = 0;
#pragma unroll
        for (int j = 0; j < PL; ++j) v[j] = bkr[lane * PL + j], sum += v[j], big = max(big, v[j]);
        uint32_t run = kp_wave_excl_sum(sum);
        if (lane == 63) cntr[r] = (int)(run + sum);
#pragma unroll
        for (int j = 0; j < PL; ++j) bkr[lane * PL + j] = run, run += v[j];
        if (big > (uint32_t)KP_MAX_OCC) *degenerate = 1;
      }
      __syncthreads();
      if (*degenerate == 0) {
        const uint32_t crr = (uint32_t)cntr[r];
        const int perq = max(64, (((int)crr + T::W * 64 - 1) / (T::W * 64)) * 64);  // words of one wavefront's run in the scans
        // bucket start and end are read in front of the barrier, with the word's first placing (the table does not change
        // any more); kept as start | members << 16 (both < 2^11).  Lanes without a word read bucket 0 and write nothing.
        uint32_t sn[T::CPW];
#pragma unroll
        for (int cc = 0; cc < T::CPW; ++cc) {
          const bool valid = ec[cc] != KP_INVALID;
          const uint32_t q = eq[cc];  // (0 without a word)
          const uint32_t s0 = bkr[q], s1 = bkr[min(q + 1u, (uint32_t)T::NBK - 1u)];
          sn[cc] = s0 | (valid ? ((q + 1u < (uint32_t)T::NBK ? s1 : crr) - s0) << 16 : 0u);
          if (valid) dst[s0 + ea[cc]] = ec[cc];
        }
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < T::CPW; ++cc)
          if (ec[cc] != KP_INVALID) {
            // a bucket holds about one word, one in 300 more than four: its first four members in flight at once (what
            // lies behind a short bucket is somebody else's word or the bucket table: read, not counted), a loop only
            // for fuller buckets (six in flight cost 1 us per tick more than the loop they spared; the plain loop was
            // compiled into three nested divergent loops with an LDS round trip each; all chunks' members in flight at
            // once spilled registers and was 1 us slower than chunk by chunk)
            const uint32_t s0 = sn[cc] & 0xffffu, nb = sn[cc] >> 16;
            const int sl = (int)(ec[cc] & KP_SLOT_MASK);
            const int fl = RF(2, sl), id1 = RF(1, sl) + 1;
            uint32_t below = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
              const uint32_t v = dst[s0 + j];
              below += ((j < nb) & (v < ec[cc])) ? 1u : 0u;
            }
#pragma nounroll
            for (uint32_t j = s0 + 4; j < s0 + nb; ++j) below += dst[j] < ec[cc] ? 1u : 0u;
            const uint32_t at = s0 + below;
            src[at] = ec[cc];
            // the word's place is known: its put (per column the largest id + 1) is part of the carry of the runs
            // behind the one it lands in (the run's number by comparisons: the division by perq was ~25 instructions)
            int run = 0;
#pragma unroll
            for (int t = 1; t < T::W; ++t) run += at >= (uint32_t)(t * perq) ? 1 : 0;
            atomicMax(&tot[(r * T::W + run) * 2 * N + ((fl >> 3) & 1) * N + (fl & 7)], id1);
          }
        __syncthreads();
        sort_passes = 0;  // sorted, in sortA
      }
    }
    // ---- LSD radix sort of the replica's words on the rank bits, in LDS (the words the replica has no part in are
    // all ones: they stay behind the others)
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
    // ---- the scans.  The replica's cr sorted words are split into runs of whole chunks, one per wavefront; the puts of
    // the runs before a wavefront's own are its carry (learnt where the sort placed the words; after the radix sort, here)
    const int cr = cntr[r];
    const int perq = max(64, ((cr + T::W * 64 - 1) / (T::W * 64)) * 64);
    const int q0 = w * perq, q1 = min(cr, q0 + perq);
    uint32_t code[T::CPW];
#pragma unroll
    for (int cc = 0; cc < T::CPW; ++cc) {
      const int p = q0 + cc * 64 + lane;
      code[cc] = p < q1 ? src[p] : KP_INVALID;
    }
    if (sort_passes > 0) {
      int* mytot = tot + (r * T::W + w) * 2 * N;
#pragma unroll
      for (int cc = 0; cc < T::CPW; ++cc)
        if (code[cc] != KP_INVALID) {
          const int sl = (int)(code[cc] & KP_SLOT_MASK);
          const int fl = RF(2, sl);
          atomicMax(&mytot[((fl >> 3) & 1) * N + (fl & 7)], RF(1, sl) + 1);
        }
      __syncthreads();
    }
    // (the sort buffers are spent once every wavefront holds its words: their place takes the conflict rows.)  Row n-2
    // of a command is its leader's own conflicts D (the PreAccept's dependencies), rows 0 .. n-3 those of the replicas
    // whose answers the leader counts, in replica order
    {
      int cg[N], cs[N], ng[N], ns[N];
#pragma unroll
      for (int l = 0; l < N; ++l) cg[l] = base[r * 2 * N + l], cs[l] = base[r * 2 * N + N + l], ng[l] = 0, ns[l] = 0;
      for (int w2 = 0; w2 < w; ++w2) {
        const int* o = tot + (r * T::W + w2) * 2 * N;
#pragma unroll
        for (int l = 0; l < N; ++l) cg[l] = imax(cg[l], o[l]), cs[l] = imax(cs[l], o[N + l]);
      }
      bool first = true;
#pragma unroll
      for (int cc = 0; cc < T::CPW; ++cc) {
        const bool valid = code[cc] != KP_INVALID;
        const int sl = valid ? (int)(code[cc] & KP_SLOT_MASK) : 0;
        const int fl = valid ? RF(2, sl) : 0;
        const int id1 = valid ? RF(1, sl) + 1 : 0;  // TopOne.put: max(.., id + 1), util/TopOne.scala:12-15
        const int L = fl & 7;
        int dep[N];
        if (q0 + cc * 64 < q1) scan_chunk<N>(valid, (fl >> 3) & 1, L, id1, cg, cs, ng, ns, dep);
        if (first) {  // the rows go where the sorted words were: every wavefront of the workgroup must hold its own first
          __syncthreads();
          first = false;
        }
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
    // ---- commit -> updateConflictIndex at every replica (Replica.scala:815-828): the key's watermarks learn every
    // instance of the tick (each command was scanned by its leader's replica: the max over replicas is the tick).  The
    // puts are complete once the words are sorted, so this runs here, on each replica's last wavefront (the first ones
    // decide two commands per thread below), and not as a one-wavefront tail in front of the key's last barrier
    if (w == T::W - 1 && lane < 2 * N) {
      int v = 0;
#pragma unroll
      for (int j = 0; j < N * T::W; ++j) v = imax(v, tot[j * 2 * N + lane]);
      const size_t ib = ((size_t)r * st.num_keys + k) * N;
      int32_t* p = lane < N ? &st.gets[ib + lane] : &st.sets[ib + lane - N];
      if (v > base[r * 2 * N + lane]) *p = v;
    }
    if (threadIdx.x == T::THREADS - 1) a.tot[(size_t)k * KP_TOT_STRIDE] = 0;  // the next tick claims from zero
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
#if FPX_K5_NT_OUT
        typedef int kp_int4v __attribute__((ext_vector_type(4)));
        const kp_int4v v = {o[0], o[1], o[2], o[3]};
        __builtin_nontemporal_store(v, reinterpret_cast<kp_int4v*>(a.packed + (size_t)RF(0, sl) * a.stride) + q);
#else
        reinterpret_cast<int4*>(a.packed + (size_t)RF(0, sl) * a.stride)[q] = make_int4(o[0], o[1], o[2], o[3]);
#endif
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
          const bool valid = code[cc] != KP_INVALID;
          const int sl = valid ? (int)(code[cc] & KP_SLOT_MASK) : 0;
          const int fl = valid ? RF(2, sl) : 0;
          const int x = valid ? RF(1, sl) : 0;
          const int L = fl & 7;
          int dep2[N];
          if (q0 + cc * 64 < q1) scan_chunk<N>(valid, (fl >> 3) & 1, L, x + 1, cg, cs, ng, ns, dep2);
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
              const int v = imax(dep2[l], l == L ? o[2 * N + 7] : o[N + l]);
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
    __syncthreads();  // the tables are reused by the next key
    cur = nxt;
  }
}
#undef RF
