// fpx_kernels.hpp -- hand-written HIP kernels for gfx950 (MI355X / CDNA4), wave64.
//
// The hot path of SURVEY.md section 8 as data-parallel kernels over a (log-slot x acceptor)
// struct-of-arrays resident in HBM:
//
//   vote_round[S][R], vote_value[S][R] (+ ballot[S][R] in FPX_BALLOT_PER_SLOT mode), int32,
//   slot-major: the R cells of one slot are contiguous (1 KiB at R = 256), so one wavefront moves
//   one slot row with a single 16-byte-per-lane access.
//
// Kernels (names follow SURVEY.md section 2.1):
//   k_validate   run-contract check of a device batch (slot range, slot uniqueness, one round per
//                acceptor group)
//   k_phase2     K1 (acceptor vote: a1/a2) and K3 (fused open + vote + tally: a6 + a1 + a3/a4/a5)
//   k_open       a6  ProxyLeader.handlePhase2a bookkeeping
//   k_tally      K2  ProxyLeader.handlePhase2b (a3/a4) using the K2q predicates (a5)
//   k_finalize   folds the whole-group shards and the claimed per-block rows of maxima into the
//                per-acceptor scalars round / maxVotedSlot
//   k_phase1a_*  Acceptor.handlePhase1a
//   k_quorum_eval  a5 standalone
//
// This path is integer compare / bit reduce at ~0.1 op/byte: HBM-bound, no MFMA.  What matters is
// full-line coalesced traffic, enough loads in flight per wave, and never re-reading a row: the vote
// bitmap of a slot is built in registers (one nibble per lane, OR-reduced across the lanes of the
// slot with cross-lane shuffles), tallied on the spot with popcount / mask tests, and only the
// 8-byte chosen record leaves the chip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// tuning knobs (compile time)
// One slot row in flight per wavefront: batching 2 / 4 rows per wave cost occupancy and measured
// slower (profiles/r01_tuning_sweep.txt); requesting the next row's ballots one step ahead measured
// no better than not doing it (profiles/r01_tuning_sweep2.txt) -- the other 5-6 waves of the SIMD
// already cover the latency.
#ifndef FPX_CHUNK
#define FPX_CHUNK 32   // messages per wavefront chunk at R in (128, 256]: 64 / 32 / 16 / 8 measured,
                       // 32 and 16 are best (profiles/r01_tuning_sweep4.txt): finer units balance the XCDs
#endif
#ifndef FPX_NT_LOAD
#define FPX_NT_LOAD 0  // ballot rows are read once: nontemporal loads
#endif
#ifndef FPX_PREFETCH
#define FPX_PREFETCH 0
#endif
#ifndef FPX_NT
#define FPX_NT 1  // vote rows are written once and not re-read soon: nontemporal stores
#endif

namespace fpx {

typedef int int4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

// tally key word: 0 = empty, else (round + 1) | flags.  KEY_RANGE marks the per-slot shadow of a
// Mencius noop range of length one (mencius SlotRound(slot, slot + 1, round) collides with the
// single-slot key, mencius/ProxyLeader.scala:86-90).
enum : uint32_t { KEY_DONE = 0x80000000u, KEY_RANGE = 0x40000000u, KEY_ROUND_MASK = 0x3fffffffu };
constexpr int MAX_ROUND = 0x3ffffffe;
constexpr int PART_ALL_STRIDE = 32;  // ints: one 128-byte line per shard of the whole-group maxima

// status word layout in HBM (int32[8])
// ST_ABORT: set by k_validate only (FPX_EINVAL / FPX_EORDER on a _dev batch): every later kernel up to the next
// fpx_sync applies nothing.  Errors raised while a batch is being applied (FPX_ECAPACITY,
// FPX_EFATAL_UNKNOWN_SLOTROUND) concern single messages: they set the code and everything else goes on.
enum { ST_CODE = 0, ST_INDEX = 1, ST_SLOT = 2, ST_ROUND = 3, ST_ABORT = 4 };

struct Geom {
  int32_t S, R;
  int32_t lg_rows;                                 // > 0: rows are stored leader-group-major -- slot s lives in row
                                                   // (s % L) * lg_rows + s / L (lg_rows = S / L): the slots of one leader group are
                                                   // neighbours in memory, so what one group's leader proposes or skips covers whole
                                                   // lines whatever the other groups do (0: row = slot)
  int32_t VS;                                      // row stride of vote_round / vote_value: RS, or 2 RS when the two rows of
                                                   // a slot are interleaved (R <= 4: [round x 4 | value x 4] is ONE 32-byte sector;
                                                   // as two arrays every small-group vote wrote two half sectors)
  int32_t RS;                                      // row stride of the cell arrays: R rounded up to a multiple of 4,
                                                   // so that every lane moves one aligned int4 whatever R is
  int32_t num_groups, num_leader_groups, ngroups;  // ngroups = num_leader_groups * num_groups
  int32_t qkind, qsize;                            // qsize: threshold for THRESHOLD / MAJORITY / UNANIMOUS
  int32_t grid_rows, grid_cols;
  int32_t per_slot;                                // ballot mode
  int32_t ways, wp;                                // tally ways, padded row length (4 or 8)
  int32_t base, total;                             // replica_base, replicas_total
  uint64_t member[4];                              // bits [0, total)
  int32_t part_rows;                               // rows of State::part (= the largest launch grid)
  // slot / L and slot / A without a division (fast_div): a 32-bit divide by a kernel argument is ~30 VALU instructions,
  // and the small-group vote kernel -- VALU-bound: 3900 VALU instructions per wavefront on BASELINE.json configs[4],
  // profiles/r05_cfg5.md -- met four of them per slot.  0 = divisor 1
  uint32_t l_magic, a_magic;
  int32_t l_shift, a_shift;
};

// s / d for 0 <= s < 2^31 and the (magic, shift) of d >= 2 that make_geom computes: shift = ceil(log2 d) - 1,
// magic = floor(2^(32 + shift) / d) + 1 < 2^32 (the error term s / 2^(32 + shift) stays below 1 / d because d <= 2^(shift + 1))
__device__ __forceinline__ int fast_div(int s, uint32_t magic, int shift) {
  return (int)(__umulhi((uint32_t)s, magic) >> shift);
}
// slot -> (leader group, row inside the leader group)
__device__ __forceinline__ void split_slot(const Geom& g, int s, int* lg, int* row) {
  const int q = g.l_magic ? fast_div(s, g.l_magic, g.l_shift) : s;
  *row = q, *lg = s - q * g.num_leader_groups;
}
// where slot s lives (phys_slot) and the acceptor group that votes in it (group_of_slot), from one split
__device__ __forceinline__ void place_of_slot(const Geom& g, int s, int* phys, int* grp) {
  int lg, row;
  split_slot(g, s, &lg, &row);
  *phys = g.lg_rows ? lg * g.lg_rows + row : s;
  const int ag = g.a_magic ? row - fast_div(row, g.a_magic, g.a_shift) * g.num_groups : 0;
  *grp = g.ngroups == 1 ? 0 : lg * g.num_groups + ag;
}

struct State {
  int32_t* promised;    // [ngroups][R]   Acceptor.round
  int32_t* max_voted;   // [ngroups][R]   Acceptor.maxVotedSlot
  int32_t* vote_round;  // [S][VS]  (cell (s, r) at s * VS + r)
  int32_t* vote_value;  // [S][VS]  (= vote_round + RS when the rows are interleaved)
  int32_t* ballot;      // [S][R] or null
  uint32_t* pl_key;     // [S][wp]        0 = empty, else (round + 1) | KEY_DONE
  int32_t* pl_value;    // [S][wp]
  uint64_t* pl_bits;    // [S][wp][4]
  // FPX_BALLOT_PER_SLOT: a Phase1a that nothing is ahead of is recorded per acceptor instead of being written into
  // every cell (k_p1a_*): the effective ballot of cell (s, a) is max(ballot[s][a], s >= lz_from[a] ? lz_round[a] : -1)
  int32_t* lz_round;    // [ngroups][R]   -1 = none
  int32_t* lz_from;     // [ngroups][R]   first slot the lazy promise covers (the Phase1a's chosenWatermark)
  int32_t* max_ballot;  // [ngroups][R]   an upper bound of the acceptor's effective ballots
  int32_t* p1;          // [4][R] + 1     scratch of one Phase1a: mode / a / b / c per acceptor, then the "sweep needed" flag
  uint8_t* row_voted;   // [S]            0 = no acceptor of the slot's group has ever voted in it (its cells
                        //                are all -1): partially voted cells can then be written whole without a read
  uint32_t* stamp;      // [S]            run id of the last run that touched the slot
  int32_t* run_round;   // [ngroups]      the single round of the current run per group (-1 = none)
  int32_t* status;      // [8]
  int32_t* part;        // [grid][2][ngroups*R] per-workgroup maxima rows (accepted round, voted slot)
  uint32_t* part_stamp; // [grid] the launch (Batch::launch_seq) that wrote row b of `part`: a workgroup that used the
                        //        tables writes row blockIdx.x -- no claiming counter (8192 same-address atomics were 65 us)
  int32_t* part_all;    // [2][64][PART_ALL_STRIDE] whole-group maxima (round, slot), 64 lines, per launch parity
  int32_t* log_value;   // [S]  the replica's log (BufferMap), -1 where absent
  uint8_t* log_present; // [S]
  int32_t* log_scalars; // [8]  LG_*: executedWatermark, numChosen, largestKey, scan result
};

struct Batch {
  int32_t n;
  const int32_t* slot;
  const int32_t* round;
  const int32_t* value;
  const uint64_t* target;  // n x 4 or null
  uint64_t* vote_bits;     // K1
  uint64_t* nack_bits;     // K1
  int32_t* nack_round;     // K1 / K3
  uint8_t* chosen;         // K3 (K2: newly_chosen)
  int32_t* chosen_round;
  int32_t* chosen_value;
  uint8_t* is_new;         // k_open
  const uint8_t* mask;     // k_log_ingest: which messages are Chosen (null = all)
  uint32_t run_id;
  int32_t parity;          // K1 / K3 launch counter & 1: which half of part_all this launch uses
  uint32_t launch_seq;     // K1 / K3 launch counter (never 0): stamps the rows of `part` this launch writes
  int32_t check_round;     // validate: enforce one round per group (ACCEPTOR ballot mode)
  int32_t chunk;           // K1 / K3 at G = 64: messages per wavefront (4 .. FPX_CHUNK)
  int32_t index_base;      // added to the message index an error reports (host batches launched in pieces)
  int32_t solo;            // K1 / K3: the launch is ONE workgroup, which applies its maxima itself (no k_finalize follows)
  int32_t sc_lds;          // K1 / K3 on leader-group-major rows: byte offset of 4 x 3 KiB of LDS for the column quads (0 = none)
  int32_t th_lds;          // K1 / K3, FPX_BALLOT_ACCEPTOR with several acceptor groups: byte offset of the staged rounds [ngroups * R]
  uint8_t* run_done;       // K1 / K3 with target masks at G = 64 run as two launches: the packed walk (MODE 3) takes the chunks
                           // whose messages all go to runs of acceptors and says so here, one byte per chunk; the
                           // row-at-a-time walk behind it takes the others
};

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ int group_of_slot(const Geom& g, int slot) {
  // multipaxos/ProxyLeader.scala:190 ; mencius/ProxyLeader.scala:169-176,231-234
  if (g.ngroups == 1) return 0;
  int lg, row;
  split_slot(g, slot, &lg, &row);
  const int ag = g.a_magic ? row - fast_div(row, g.a_magic, g.a_shift) * g.num_groups : 0;
  return lg * g.num_groups + ag;
}

// where slot s lives in the cell arrays and the tally tables (vote_round, vote_value, ballot, pl_key, pl_value, pl_bits),
// row_voted), and which slot a row holds; stamp and the replica log are indexed by the slot itself
__device__ __forceinline__ int phys_slot(const Geom& g, int s) {
  if (!g.lg_rows) return s;
  int lg, row;
  split_slot(g, s, &lg, &row);
  return lg * g.lg_rows + row;
}
// k_phase2: leader-group-major rows exist only for groups of at most 32 acceptors (make_geom), so the instantiations
// for bigger groups -- the headline's among them -- carry no trace of them
template <int G>
__device__ __forceinline__ int phys_slot_g(const Geom& g, int s) {
  if constexpr (G > 8) return s;
  else return phys_slot(g, s);
}
__device__ __forceinline__ int slot_of_row(const Geom& g, int p) {
  return g.lg_rows ? (p % g.lg_rows) * g.num_leader_groups + p / g.lg_rows : p;
}

__device__ __forceinline__ void report(const State& st, int code, int index, int slot, int round) {
  if (atomicCAS(&st.status[ST_CODE], 0, code) == 0) {
    st.status[ST_INDEX] = index;
    st.status[ST_SLOT] = slot;
    st.status[ST_ROUND] = round;
  }
}

__device__ __forceinline__ void row_store(int4v v, int4v* p) {
#if FPX_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

__device__ __forceinline__ void report_abort(const State& st, int code, int index, int slot, int round) {
  st.status[ST_ABORT] = 1;
  report(st, code, index, slot, round);
}

__device__ __forceinline__ int popc256(const uint64_t x[4]) {
  return __popcll(x[0]) + __popcll(x[1]) + __popcll(x[2]) + __popcll(x[3]);
}

// bits [lo, lo + n) of the 256-bit set, restricted to word w
__device__ __forceinline__ uint64_t range_mask(int lo, int n, int w) {
  const int wlo = w * 64;
  const int a = lo > wlo ? lo : wlo;
  const int b = (lo + n) < (wlo + 64) ? (lo + n) : (wlo + 64);
  if (a >= b) return 0ull;
  const int len = b - a;
  const uint64_t m = len == 64 ? ~0ull : ((1ull << len) - 1ull);
  return m << (a - wlo);
}

// K2q: isWriteQuorum on a member-masked 256-bit acceptor set.
//   THRESHOLD        ProxyLeader.scala:238      |X| >= f + 1
//   SIMPLE_MAJORITY  SimpleMajority.scala:30,46 |X| >= n / 2 + 1
//   UNANIMOUS        UnanimousWrites.scala:50   X == members
//   GRID             Grid.scala:43-50           every row has a member of X
__device__ __forceinline__ bool is_write_quorum(const Geom& g, const uint64_t x[4]) {
  if (g.qkind != 2) return popc256(x) >= g.qsize;
  bool ok = true;
  for (int r = 0; r < g.grid_rows; ++r) {
    const int lo = r * g.grid_cols;
    uint64_t hit = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) hit |= x[w] & range_mask(lo, g.grid_cols, w);
    ok = ok && (hit != 0);
  }
  return ok;
}

//   THRESHOLD        n - f (the sets that intersect every (f+1)-subset)
//   SIMPLE_MAJORITY  SimpleMajority.scala:41-47
//   UNANIMOUS        UnanimousWrites.scala:36-42  non-empty
//   GRID             Grid.scala:36-41             some row fully contained
__device__ __forceinline__ bool is_read_quorum(const Geom& g, const uint64_t x[4]) {
  const int c = popc256(x);
  switch (g.qkind) {
    case 0: return c >= g.total - (g.qsize - 1);
    case 1: return c >= g.qsize;
    case 3: return c >= 1;
    default: break;
  }
  bool any = false;
  for (int r = 0; r < g.grid_rows; ++r) {
    const int lo = r * g.grid_cols;
    bool all = true;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint64_t m = range_mask(lo, g.grid_cols, w);
      all = all && ((x[w] & m) == m);
    }
    any = any || all;
  }
  return any;
}

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
  return (uint64_t)__shfl((unsigned long long)v, src);
}
__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int m) {
  return (uint64_t)__shfl_xor((unsigned long long)v, m);
}

// 256-bit left shift by sh (0 <= sh < 256, multiple of 4)
__device__ __forceinline__ void shl256(uint64_t x[4], int sh) {
  if (sh == 0) return;
  const int ws = sh >> 6, bs = sh & 63;
  // whole-word moves with static register indices (a runtime-indexed array would go to scratch)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (ws > i) {
      x[3] = x[2];
      x[2] = x[1];
      x[1] = x[0];
      x[0] = 0ull;
    }
  }
  if (bs != 0) {
    x[3] = (x[3] << bs) | (x[2] >> (64 - bs));
    x[2] = (x[2] << bs) | (x[1] >> (64 - bs));
    x[1] = (x[1] << bs) | (x[0] >> (64 - bs));
    x[0] = x[0] << bs;
  }
}

// Is the 256-bit set m (bits >= R clear) a cyclic RUN of positions [start, start + len) mod 256 with start a multiple of 16
// (a 64-byte sector of a row) and len <= 128 -- up to the positions R .. 255, which no acceptor owns (a run that passes
// over them holds them too)?  What a thrifty proxy leader that sends every Phase2a to f + 1 NEIGHBOURING acceptors
// produces (any f + 1 of the group will do: multipaxos/ProxyLeader.scala:190-191).  cell0 = first 16-byte cell of the
// run, ncell = cells of the sectors it touches (a multiple of 4, at most 32).
__device__ __forceinline__ bool classify_run(const uint64_t m[4], int R, int* cell0, int* ncell) {
  uint64_t w[4] = {m[0], m[1], m[2], m[3]};
  if (R < 256 && ((w[3] >> (R - 193)) & 1ull) && (w[0] & 1ull)) w[3] |= ~0ull << (R - 192);  // 192 < R: the run wraps over the gap
  uint64_t first[4];
  first[0] = w[0] & ~((w[0] << 1) | (w[3] >> 63));
  first[1] = w[1] & ~((w[1] << 1) | (w[0] >> 63));
  first[2] = w[2] & ~((w[2] << 1) | (w[1] >> 63));
  first[3] = w[3] & ~((w[3] << 1) | (w[2] >> 63));
  int start = -1;
#pragma unroll
  for (int j = 3; j >= 0; --j)
    if (first[j]) start = 64 * j + (int)__ffsll((unsigned long long)first[j]) - 1;
  const int len = popc256(w);
  if (start < 0 || (start & 15) || len > 128) return false;
  const int over = start + len - 256;
  bool same = true;
#pragma unroll
  for (int j = 0; j < 4; ++j) same = same && w[j] == (range_mask(start, len, j) | (over > 0 ? range_mask(0, over, j) : 0ull));
  *cell0 = start >> 2, *ncell = ((len + 15) >> 4) << 2;
  return same;
}

// Assemble the per-slot bitmap from per-lane nibbles.  A slot is handled by G consecutive lanes
// (lane gi of the group owns local acceptors 4*gi .. 4*gi+3).  Every lane of the group returns the
// full bitmap (bit position = global acceptor index = base + local index).
template <int G>
__device__ __forceinline__ void assemble_bits(uint32_t nibble, int lane, int base, uint64_t out[4]) {
  const int gi = lane & (G - 1);
  uint64_t v = (uint64_t)nibble << (4 * (gi & 15));
  constexpr int ROW = G < 16 ? G : 16;
#pragma unroll
  for (int m = 1; m < ROW; m <<= 1) v |= shfl_xor64(v, m);
  // every lane of a 16-lane row now holds that row's 64-bit word
  const int gbase = lane & ~(G - 1);
  if (G == 64) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, w * 16);
      const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), w * 16);
      out[w] = ((uint64_t)hi << 32) | lo;
    }
  } else {
    out[0] = (G > 16) ? shfl64(v, gbase) : v;
    out[1] = (G > 16) ? shfl64(v, gbase + 16) : 0ull;
    out[2] = 0ull;
    out[3] = 0ull;
  }
  shl256(out, base);
}

// max over the G lanes of a slot group
template <int G>
__device__ __forceinline__ int group_max(int v) {
#pragma unroll
  for (int m = 1; m < G; m <<= 1) {
    const int o = __shfl_xor(v, m);
    v = o > v ? o : v;
  }
  return v;
}

// max over the 64 lanes of values >= 0 on the DPP network (4 row_shr steps inside each row of 16, row_bcast:15 / :31
// across the rows; lanes without a source read 0): the total arrives in lane 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or0(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ int wave_max_to_lane63(int v) {
  int o;
  o = dpp_or0<0x111, 0xF>(v), v = o > v ? o : v;  // row_shr:1
  o = dpp_or0<0x112, 0xF>(v), v = o > v ? o : v;  // row_shr:2
  o = dpp_or0<0x114, 0xF>(v), v = o > v ? o : v;  // row_shr:4
  o = dpp_or0<0x118, 0xF>(v), v = o > v ? o : v;  // row_shr:8
  o = dpp_or0<0x142, 0xA>(v), v = o > v ? o : v;  // row_bcast:15 -> rows 1, 3
  o = dpp_or0<0x143, 0xC>(v), v = o > v ? o : v;  // row_bcast:31 -> rows 2, 3
  return v;
}

// wave-level ordering of LDS traffic between lanes of one wavefront (DS ops of a wave execute in
// order; this keeps the compiler from moving them)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// small device fills on the context's stream (hipMemsetAsync of a few bytes measured ~170 us per call between
// kernels on this stack, a kernel launch ~5 us: profiles/r02_adversarial.txt)
__global__ void __launch_bounds__(256) k_fill32(int32_t* p, int32_t v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// fpx_proxy_forget / fpx_recycle_slots when rows are leader-group-major (the slots of a range are L rows apart): one
// thread per slot -- its tally keys emptied, with votes != 0 its cells back to "no vote"
__global__ void __launch_bounds__(256) k_clear_slots(const Geom g, const State st, int first, int count, int votes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const size_t ps = (size_t)phys_slot(g, first + i);
  for (int w = 0; w < g.wp; ++w) st.pl_key[ps * g.wp + w] = 0;
  if (votes) {
    for (int r = 0; r < g.RS; ++r) st.vote_round[ps * g.VS + r] = -1, st.vote_value[ps * g.VS + r] = -1;
    if (st.row_voted) st.row_voted[ps] = 0;
  }
}

// ------------------------------------------------------------------------------------------------
// k_validate: run contract of a device batch.  One thread per message.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_validate(const Geom g, const State st, const Batch b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.n) return;
  const int s = b.slot[i], r = b.round ? b.round[i] : 0;
  if (s < 0 || s >= g.S || r < 0 || r > MAX_ROUND) {
    report_abort(st, 1 /*FPX_EINVAL*/, i + b.index_base, s, r);
    return;
  }
  // (1) slots pairwise distinct within the run
  const uint32_t old = atomicExch(&st.stamp[s], b.run_id);
  if (old == b.run_id) report_abort(st, 6 /*FPX_EORDER*/, i + b.index_base, s, r);
  // (2) one round per acceptor group within the run
  if (b.check_round) {
    // every message of a group looks at the same word, and same-address requests serialise in one L2 channel
    // (170 us for a 20 k-message epoch when every thread asked).  One acceptor group: the wavefront agrees on its
    // round among itself and sends ONE lane.
    int* rr = &st.run_round[group_of_slot(g, s)];
    bool ask = true;
    if (g.ngroups == 1) {
      const int r0 = __builtin_amdgcn_readfirstlane(r);
      if (r != r0) report_abort(st, 6, i + b.index_base, s, r);
      ask = __builtin_amdgcn_readfirstlane(i) == i;  // the first active lane
    }
    if (ask) {
      int cur = __hip_atomic_load(rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == -1) {
        cur = atomicCAS(rr, -1, r);
        if (cur == -1) cur = r;
      }
      if (cur != r) report_abort(st, 6, i + b.index_base, s, r);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_phase2<G, RMW, PERSLOT, FUSED>
//   G       lanes per slot (power of two, 4*G >= R): 64 for R in (128, 256], ... 1 for R <= 4
//   RMW     partially voted 16-byte cells are written by load / blend / store instead of 4-byte stores
//           (FPX_F_SCATTERED_TARGETS launches with target masks; rows are always moved as int4's, they are
//           padded to a multiple of 4 cells)
//   PERSLOT ballot[S][R] in HBM instead of the per-acceptor scalar
//   FUSED   K3 (open + vote + tally) instead of K1 (vote, bitmaps out)
// A wavefront owns a chunk of consecutive messages (FPX_CHUNK = 32 at G = 64, else 64): it stages
// their (slot, round, value) in registers with one coalesced load each, runs the proxy leader's open
// step for all of them at once (K3), then walks them Q = 64/G at a time.
// LDS: the workgroup's maxima (whole-group scalars, or [2][ntab] tables after a partial vote), then
// per-wave staging of the outputs so that they leave the CU as full coalesced lines.
// ------------------------------------------------------------------------------------------------
template <bool FUSED>
struct WaveOut;
template <>
struct WaveOut<false> {    // K1: per-wave LDS staging of the per-slot bitmaps, indexed by message-in-chunk
  uint64_t votes[64][4];
  uint64_t nacks[64][4];
  int32_t nack_round[64];
};
template <>
struct WaveOut<true> {     // K3: only the chosen flags are staged; bitmaps stay in registers
  int32_t nack_round[64];
  int32_t chosen[64];
};

template <int G, int MODE, int PS, bool FUSED>
__global__ void __launch_bounds__(256)
    k_phase2(const Geom g, const State st, const Batch b) {
  // PS 0: the acceptor's round is a scalar (FPX_BALLOT_ACCEPTOR); 1: ballot[S][R] in HBM; 2: the same with lazy
  // Phase1a promises to honour (only while some are outstanding: 8 VGPRs the steady state does not pay)
  constexpr bool PERSLOT = PS != 0;
  constexpr bool LAZY = PS == 2;
  // MODE 0: no target masks (dense delivery) -- the lean kernel of the steady state; 1: target masks; 2: target
  // masks + FPX_F_SCATTERED_TARGETS.  The target-mask code (LDS staging, fresh-row blend) costs 6-12 VGPRs = one
  // wave per SIMD, which the dense stream would pay for nothing.
  constexpr bool TGT = MODE != 0;
  constexpr bool RMW = MODE == 2;  // (3: target masks that are all runs of neighbouring acceptors, see PACK below)
  constexpr bool VEC = true;  // 16-byte row accesses
  constexpr int Q = 64 / G;           // slots per step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  if (st.status[ST_ABORT] != 0) return;  // a failed validation applies nothing

  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  if constexpr (G == 64 && (MODE == 1 || MODE == 2)) {
    // behind a packed walk: a workgroup all of whose chunks that walk took leaves before it sets anything up (8192
    // workgroups that only found out in their unit loop were 24 us of a 260 us step)
    if (b.run_done) {
      const int nunits = (b.n + b.chunk - 1) / b.chunk;
      bool mine = false;
      for (int unit = blockIdx.x * 4 + wib; unit < nunits; unit += gridDim.x * 4) mine = mine || b.run_done[unit] == 0;
      if (!__syncthreads_or(mine)) return;
    }
  }
  const int gi = lane & (G - 1);
  const int q = lane / G;
  const int ntab = g.ngroups * g.R;
  int32_t* tab_pr = reinterpret_cast<int32_t*>(smem);
  int32_t* tab_mv = tab_pr + ntab;
  // [0] this workgroup used the tables, [1] its partial-table row, [2] [3] whole-group maxima (round, slot)
  int32_t* blk_flag = tab_pr + 2 * ntab;
  WaveOut<FUSED>* wo = reinterpret_cast<WaveOut<FUSED>*>(smem + (((size_t)ntab * 8 + 16 + 15) & ~(size_t)15)) + wib;
  // launches that carry target masks: the chunk's masks (64 x 32 B per wave) are staged here with one coalesced
  // load, so that the walk has no global load of its own in the ACCEPTOR model (a dependent load per row made
  // thrifty delivery latency-bound: 0.88 ms per 2^20 rows against 0.42 ms dense, profiles/r02_thrifty.txt)
  uint64_t* wt = reinterpret_cast<uint64_t*>(smem + (((size_t)ntab * 8 + 16 + 15) & ~(size_t)15) + 4 * sizeof(WaveOut<FUSED>)) +
                 (size_t)wib * (256 + 16);

  for (int i = threadIdx.x; i < 2 * ntab + 4; i += blockDim.x) tab_pr[i] = (i < 2 * ntab || i >= 2 * ntab + 2) ? -1 : 0;
  // FPX_BALLOT_ACCEPTOR with several acceptor groups: every step needs the rounds of ITS group's acceptors.  They are
  // staged in LDS once per workgroup: as global loads inside the walk they put an s_waitcnt vmcnt(0) into every step,
  // and on gfx9 vmcnt counts STORES too -- each step then waited until the rows of the step before had reached memory
  // (a full store round trip per row and wavefront)
  int32_t* tab_th = reinterpret_cast<int32_t*>(smem + b.th_lds);
  if (!PERSLOT && g.ngroups != 1)
    for (int i = threadIdx.x; i < ntab; i += blockDim.x) tab_th[i] = st.promised[i];
  __syncthreads();
  int w_slot = -1, w_round = -1;  // maxima over the steps in which the whole group voted (wave-uniform)
  bool table_used = false;

  const bool one_group = g.ngroups == 1;
  const int r0 = 4 * gi;  // first local acceptor of this lane
  uint32_t own = 0;       // which of my 4 acceptors exist
#pragma unroll
  for (int k = 0; k < 4; ++k) own |= (r0 + k < g.R) ? (1u << k) : 0u;
  const int bitpos = g.base + r0;  // global bit of my first acceptor

  // per-lane maxima when there is one acceptor group (registers; folded into LDS at the end)
  int acc_pr[4] = {-1, -1, -1, -1}, acc_mv[4] = {-1, -1, -1, -1};
  int4v init_thr = {-1, -1, -1, -1};
  if (!PERSLOT && one_group) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (own >> k & 1) init_thr[k] = st.promised[r0 + k];
  }

  // the packed walk (below): lane (half, l32) owns the cells l32 and 32 + l32 of every row it meets
  // MODE 3 (G = 64 only): the packed walk alone, for the chunks all of whose messages go to runs; the others are left
  // to a MODE 1 / 2 launch behind this one (Batch::run_done) -- with both walks in one kernel the register count went
  // from 85 to 141
  constexpr bool PACK = G == 64 && MODE == 3;
  static_assert(MODE != 3 || (G == 64 && PS != 2), "the packed walk: 256-cell rows, no lazy promises");
  int pk_pr[8] = {-1, -1, -1, -1, -1, -1, -1, -1}, pk_mv[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
  bool pk_used = false;

  int lzr[4] = {-1, -1, -1, -1}, lzf[4] = {0, 0, 0, 0};
  if (LAZY && one_group) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (own >> k & 1) lzr[k] = st.lz_round[r0 + k], lzf[k] = st.lz_from[r0 + k];
  }

  // messages per wavefront chunk: 64, or at G = 64 the launch's choice (32 for big batches, fewer when the batch
  // would not otherwise fill the chip: a wave walks its chunk one row at a time)
  const int CH = (G == 64) ? b.chunk : 64;
  // Leader-group-major rows (Geom::lg_rows) want the 64 messages of a wavefront to be 64 consecutive slots of ONE
  // leader group.  A batch that comes as the leader groups' batches back to back is that already.  A batch in slot
  // order across P proposing leader groups (message i + P is the next slot of message i's group) is walked COLUMN BY
  // COLUMN instead: in tiles of 64 P messages, a wavefront takes every P-th message of its tile.  P is read off the
  // batch (the first message after message 0 with message 0's leader group); any P gives a permutation of the
  // messages, so a batch without that regularity only loses the speed, and outputs stay indexed by message.  The
  // workgroups of one XCD take neighbouring columns (they share the tile's input and output lines in that L2).
  // Measured on BASELINE.json configs[4] (profiles/r03_cfg5.md): 241 -> 92 us for the slot-ordered batch, against 50 us
  // for the same messages grouped by leader group -- inputs and outputs are one L2 request per message and array here
  // (a workgroup taking 32 neighbouring columns itself, for the L1's sake: 105 us, fewer workgroups in flight); with the
  // column quads below the step is 0.118 ms against 0.109 ms for the grouped batch.
  int period = 1, bx = blockIdx.x;
  // (small groups only: at R > 32 a slot's row is 128 bytes or more by itself, and the G = 64 kernel of the headline
  // must not pay registers for this)
  constexpr bool COLS = G <= 8;
  if (COLS && g.lg_rows && CH == 64 && b.n >= 128) {
    const int L = g.num_leader_groups, lg0 = b.slot[0] % L, lim = b.n < 2 * L ? b.n : 2 * L;
    for (int k0 = 1; k0 < lim; k0 += 64) {
      const int k = k0 + lane;
      const uint64_t hit = __ballot(k < lim && b.slot[k] % L == lg0);
      if (hit) {
        period = k0 + (int)__ffsll((unsigned long long)hit) - 1;
        break;
      }
    }
    if (period > 1 && (gridDim.x & 7) == 0) bx = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  }
  // a period that is a multiple of 4: a wavefront takes FOUR neighbouring columns in one go -- lane j reads the 16
  // bytes (row j, columns c0 .. c0 + 3) of each input array (a quarter of the L2 requests of four strided dword loads)
  // into its own four words of LDS, walks the columns one after the other, parks each column's result there and stores
  // the four results of its row as 16 bytes per output array
  const int kc = (COLS && period > 1 && (period & 3) == 0 && b.sc_lds) ? 4 : 1;
  const int tile = 64 * period, tiled = period > 1 ? (b.n / tile) * tile : 0;
  const int u1 = tiled / (64 * kc), per_tile = period / kc;  // units of kc columns x 64 rows; units per tile
  const int units = u1 + (b.n - tiled + CH - 1) / CH;        // + the plain chunks behind the tiles
  int32_t* sc = reinterpret_cast<int32_t*>(smem + b.sc_lds) + wib * (3 * 4 * 64);  // [3][4][64] per wavefront
  for (int unit = bx * 4 + wib; unit < units; unit += gridDim.x * 4) {
    if constexpr (G == 64 && (MODE == 1 || MODE == 2)) {
      if (b.run_done && b.run_done[unit]) continue;  // the packed walk of the launch before this one took the chunk
    }
    const bool strided = COLS && unit < u1;
    const bool quad = COLS && strided && kc == 4;
    const int v0 = tiled + (unit - u1) * CH;  // plain chunk: its first message
    const int mbase = strided ? (unit / per_tile) * tile + lane * period + (unit % per_tile) * kc : v0 + lane;
    if (quad) {
      const int4 a = *reinterpret_cast<const int4*>(b.slot + mbase), c = *reinterpret_cast<const int4*>(b.round + mbase),
                 d = *reinterpret_cast<const int4*>(b.value + mbase);
      sc[0 * 256 + 0 * 64 + lane] = a.x, sc[0 * 256 + 1 * 64 + lane] = a.y, sc[0 * 256 + 2 * 64 + lane] = a.z, sc[0 * 256 + 3 * 64 + lane] = a.w;
      sc[1 * 256 + 0 * 64 + lane] = c.x, sc[1 * 256 + 1 * 64 + lane] = c.y, sc[1 * 256 + 2 * 64 + lane] = c.z, sc[1 * 256 + 3 * 64 + lane] = c.w;
      sc[2 * 256 + 0 * 64 + lane] = d.x, sc[2 * 256 + 1 * 64 + lane] = d.y, sc[2 * 256 + 2 * 64 + lane] = d.z, sc[2 * 256 + 3 * 64 + lane] = d.w;
    }
    for (int cc = 0; cc < (quad ? 4 : 1); ++cc) {
    // ---- stage the chunk: lane i owns message i (or every period-th message of the unit's tile) ----
    const int m = mbase + cc;
    const bool mv = strided || (lane < CH && m < b.n);
    const int myslot = quad ? sc[0 * 256 + cc * 64 + lane] : (mv ? b.slot[m] : -1);
    const int myround = quad ? sc[1 * 256 + cc * 64 + lane] : (mv ? b.round[m] : 0);
    const int myvalue = quad ? sc[2 * 256 + cc * 64 + lane] : (mv ? b.value[m] : 0);
    // a slot nobody has voted in yet (the common case: a first proposal) holds -1 in every cell, so a partial
    // vote -- thrifty delivery to a random f+1 of the group (ProxyLeader.scala:190-191), or some acceptors
    // Nacking -- can be stored as whole 16-byte cells blended with -1: full-line traffic, nothing read
    // the slot's row and acceptor group, ONCE per message: the walk takes them from this lane (small groups only -- at
    // G > 8 a row is the slot itself and there is one acceptor group per leader group)
    int myphys = myslot, mygrp = 0;
    if constexpr (G <= 8) {
      if (g.lg_rows || !one_group) place_of_slot(g, myslot, &myphys, &mygrp);
    }
    bool myfresh = false;
    if constexpr (TGT) myfresh = mv && st.row_voted[myphys] == 0;
    // MODE 3: is this chunk the packed walk's?  Decided from the messages' masks before anything else of the chunk is
    // loaded (a chunk that is not costs this launch its masks and row_voted bytes, nothing more)
    int mycell0 = 0, myncell = 0;
    if constexpr (PACK) {
      bool okm = true;
      if (mv) {
        const uint64_t* tp = b.target + (size_t)m * 4;
        const uint64_t tm[4] = {tp[0] & g.member[0], tp[1] & g.member[1], tp[2] & g.member[2], tp[3] & g.member[3]};
        okm = myfresh && classify_run(tm, g.R, &mycell0, &myncell);
      }
      const bool packed = __all(okm);
      if (lane == 0) b.run_done[unit] = packed ? 1 : 0;
      if (!packed) continue;  // nothing of this chunk has been touched: the launch behind this one walks it row by row
    }
    // ProxyLeader.handlePhase2a for 64 messages at once (ProxyLeader.scala:176-184): lane i reads the
    // tally-key row of its slot (one gathered 16-byte access per lane), detects a known (slot, round)
    // and picks the free way.  The key word is written back after the walk, one lane per message.
    int myway = -1;
    bool mydeliver = mv;
    if (FUSED && mv) {
      const uint32_t* kr = st.pl_key + (size_t)myphys * g.wp;
      const uint4v k0 = *reinterpret_cast<const uint4v*>(kr);
      uint4v k1 = uint4v{0, 0, 0, 0};
      if (g.wp == 8) k1 = *reinterpret_cast<const uint4v*>(kr + 4);
      const uint32_t want = (uint32_t)myround + 1u;
      const uint32_t keys[8] = {k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
      bool dup = false;
#pragma unroll
      for (int w = 7; w >= 0; --w) {
        if (w < g.ways) {
          dup = dup || ((keys[w] & KEY_ROUND_MASK) == want);
          if (keys[w] == 0) myway = w;
        }
      }
      mydeliver = !dup && myway >= 0;  // a known (slot, round) is ignored and NOT forwarded
      if (!dup && myway < 0) report(st, 5 /*FPX_ECAPACITY*/, m + b.index_base, myslot, myround);
    }

    if constexpr (TGT) {
      if (strided) {
#pragma unroll
        for (int w = 0; w < 4; ++w) wt[lane * 4 + w] = b.target[(size_t)m * 4 + w];
      } else {
        const size_t w0 = (size_t)v0 * 4, wend = (size_t)b.n * 4;
        for (int w = lane; w < CH * 4; w += 64)
          if (w0 + w < wend) wt[w] = b.target[w0 + w];
      }
      wave_lds_sync();
    }

    // ---- walk the chunk, Q slots per step; the ballot row of the next step is already in flight ----
    // message src of the chunk as seen from this lane (at G = 1 a lane walks its own message: nothing to fetch)
    auto pick = [&](int v, int src) -> int {
      if constexpr (G == 1) return v;
      else return __shfl(v, src);
    };
    auto load_thr = [&](int step, int& s_out, int& grp_out, int& phys_out) -> int4v {
      const int src = step * Q + q;
      const int s = pick(myslot, src);
      s_out = s;
      grp_out = 0;
      phys_out = s;
      // (fetched by every lane, outside the branch below: a lane that sits the branch out could not be read from)
      if constexpr (G <= 8) phys_out = pick(myphys, src), grp_out = pick(mygrp, src);
      int4v thr = init_thr;
      if (s >= 0) {
        const size_t row = (size_t)phys_out * (size_t)g.RS + (size_t)r0;
        if constexpr (G > 8) {
          if (!one_group) grp_out = group_of_slot(g, s);
        }
        if (PERSLOT) {
          if (VEC) {
#if FPX_NT_LOAD
            if (own) thr = __builtin_nontemporal_load(reinterpret_cast<const int4v*>(st.ballot + row));
#else
            if (own) thr = *reinterpret_cast<const int4v*>(st.ballot + row);
#endif
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (own >> k & 1) thr[k] = st.ballot[row + k];
          }
        } else if (!one_group) {
          const int32_t* pr = tab_th + grp_out * g.R + r0;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (own >> k & 1) thr[k] = pr[k];
        }
        if constexpr (LAZY) {
          if (one_group) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int lz = s >= lzf[k] ? lzr[k] : -1;
              thr[k] = lz > thr[k] ? lz : thr[k];
            }
          } else {
            const size_t e = (size_t)grp_out * g.R + r0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (own >> k & 1) {
                const int lz = s >= st.lz_from[e + k] ? st.lz_round[e + k] : -1;
                thr[k] = lz > thr[k] ? lz : thr[k];
              }
            }
          }
        }
      }
      return thr;
    };

    // ---- thrifty delivery to RUNS of acceptors: two rows per step -------------------------------------------------
    // The walk below costs ~180 vector instructions per row whatever the row moves: 0.31 ms per 2^20 rows, the time of
    // a dense step, for half the bytes (profiles/r04_thrifty.md).  When every message of the chunk goes to a sector-aligned run of
    // at most 128 acceptors of a fresh row (classify_run), a wavefront takes TWO rows per step, 32 lanes each.  A run
    // is a window of 32 consecutive cells (mod 64), which holds exactly one of the cells l32 and 32 + l32 for every
    // l32: lane (half, l32) takes that one, so it only ever meets two fixed cells -- its acceptors' rounds and maxima
    // stay in registers.  With no Nack in the step the votes of a row ARE its target set: the bitmap is not assembled.
    if constexpr (PACK) {
      pk_used = true;
      __builtin_amdgcn_s_waitcnt(0x0F70);  // the chunk's own loads (see below)
      const int half = lane >> 5, l32 = lane & 31;
      uint32_t own_hi = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) own_hi |= (128 + 4 * l32 + k < g.R) ? (1u << k) : 0u;
      int4v th_lo = {-1, -1, -1, -1}, th_hi = {-1, -1, -1, -1};
      if (!PERSLOT) {
        th_lo = *reinterpret_cast<const int4v*>(st.promised + 4 * l32);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (own_hi >> k & 1) th_hi[k] = st.promised[128 + 4 * l32 + k];
        __builtin_amdgcn_s_waitcnt(0x0F70);
      }
      uint64_t* scr = wt + 256 + half * 8;  // [2][4]: votes, Nacks of my row when somebody Nacks
      for (int t = 0; t < CH / 2; ++t) {
        const int src = 2 * t + half;
        const int s = __shfl(myslot, src);
        const int rnd = __shfl(myround, src);
        const int val = __shfl(myvalue, src);
        const bool deliver = __shfl((int)mydeliver, src) != 0 && s >= 0;
        const int c0 = __shfl(mycell0, src), nc = __shfl(myncell, src);
        const int j = (l32 - c0) & 31;   // my place in the run's window
        const int c = (c0 + j) & 63;     // my cell of this row
        const bool hi = c >= 32, active = j < nc && deliver;
        const int bp = 4 * c;
        const uint32_t own_c = hi ? own_hi : 0xFu;
        const uint64_t tw = active ? wt[src * 4 + (bp >> 6)] : 0ull;
        const uint32_t tn = own_c & (uint32_t)((tw >> (bp & 63)) & 0xFull);
        const size_t row = (size_t)s * (size_t)g.RS + (size_t)bp, vrow = (size_t)s * (size_t)g.VS + (size_t)bp;
        int4v thr = hi ? th_hi : th_lo;
        if (PERSLOT && active) thr = *reinterpret_cast<const int4v*>(st.ballot + row);
        uint32_t acc = 0, nck = 0;
        int nr = -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool tk = (tn >> k) & 1u;
          const bool ok = tk && (rnd >= thr[k]);  // Acceptor.scala:192
          acc |= ok ? (1u << k) : 0u;
          if (tk && !ok) nck |= 1u << k, nr = thr[k] > nr ? thr[k] : nr;
        }
        if (active) {  // a fresh row: every cell of the sectors the run touches is written whole, -1 where nobody votes
          int4v rr, vv, nb = thr;
          bool ballot_moves = false;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bool a = (acc >> k) & 1u;
            rr[k] = a ? rnd : -1, vv[k] = a ? val : -1;
            if (a) ballot_moves = ballot_moves || thr[k] != rnd, nb[k] = rnd;
          }
          row_store(rr, reinterpret_cast<int4v*>(st.vote_round + vrow));
          row_store(vv, reinterpret_cast<int4v*>(st.vote_value + vrow));
          if (PERSLOT && ballot_moves) row_store(nb, reinterpret_cast<int4v*>(st.ballot + row));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool a = (acc >> k) & 1u;
          const bool al = a && !hi, ah = a && hi;
          pk_mv[k] = (al && s > pk_mv[k]) ? s : pk_mv[k], pk_pr[k] = (al && rnd > pk_pr[k]) ? rnd : pk_pr[k];
          pk_mv[4 + k] = (ah && s > pk_mv[4 + k]) ? s : pk_mv[4 + k], pk_pr[4 + k] = (ah && rnd > pk_pr[4 + k]) ? rnd : pk_pr[4 + k];
        }
        uint64_t vb[4], nbits[4] = {0, 0, 0, 0};
        int nrm = -1;
        if (!__any(nck != 0u)) {  // everybody asked voted: the votes are the targets
#pragma unroll
          for (int w = 0; w < 4; ++w) vb[w] = deliver ? (wt[src * 4 + w] & g.member[w]) : 0ull;
        } else {
          if (l32 < 8) scr[l32] = 0ull;
          wave_lds_sync();
          if (acc) atomicOr(reinterpret_cast<unsigned long long*>(&scr[bp >> 6]), (unsigned long long)acc << (bp & 63));
          if (nck) atomicOr(reinterpret_cast<unsigned long long*>(&scr[4 + (bp >> 6)]), (unsigned long long)nck << (bp & 63));
          wave_lds_sync();
#pragma unroll
          for (int w = 0; w < 4; ++w) vb[w] = scr[w], nbits[w] = scr[4 + w];
          nrm = group_max<32>(nr);
          wave_lds_sync();
        }
        if constexpr (!FUSED) {
          if (l32 == 0) {
#pragma unroll
            for (int w = 0; w < 4; ++w) wo->votes[src][w] = vb[w], wo->nacks[src][w] = b.nack_bits ? nbits[w] : 0ull;
            wo->nack_round[src] = nrm;
          }
        } else {
          bool ch = false;  // ProxyLeader.scala:235-256
          const int way = __shfl(myway, src);
          if (deliver) {
            uint64_t x[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) x[w] = vb[w] & g.member[w];
            ch = is_write_quorum(g, x);
            if (!ch && l32 == 0) {
              const size_t e = (size_t)s * g.wp + way;
#pragma unroll
              for (int w = 0; w < 4; ++w) st.pl_bits[e * 4 + w] = x[w];
            }
          }
          if (l32 == 0) wo->chosen[src] = ch ? 1 : 0, wo->nack_round[src] = nrm;
        }
      }
    }
    if constexpr (!PACK) {
    // Everything the chunk loaded (messages, tally keys, row_voted, masks, and at kernel start the acceptors' rounds) is
    // waited for HERE: left to the compiler, the s_waitcnt vmcnt(0) of a first use inside the walk sits in the loop
    // body, where -- vmcnt counts stores on gfx9 -- it makes every step wait for the rows the step before it wrote.
    // In the FPX_BALLOT_ACCEPTOR model the walk then has no vector-memory wait at all: rows stream out back to back.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt and lgkmcnt untouched
    int s_cur, grp_cur, phys_cur;
    int4v thr_cur = load_thr(0, s_cur, grp_cur, phys_cur);
    for (int t = 0; t < G * CH / 64; ++t) {
#if FPX_PREFETCH
      int s_nxt = -1, grp_nxt = 0, phys_nxt = 0;
      int4v thr_nxt = init_thr;
      if (t + 1 < G) thr_nxt = load_thr(t + 1, s_nxt, grp_nxt, phys_nxt);
#endif
      const int src = t * Q + q;
      const int s = s_cur;
      const int rnd = pick(myround, src);
      const int val = pick(myvalue, src);
      const bool deliver = pick((int)mydeliver, src) != 0 && s >= 0;
      bool fresh = false;
      if constexpr (TGT) fresh = pick((int)myfresh, src) != 0;
      const int4v thr = thr_cur;
      uint64_t tw = ~0ull;
      if constexpr (TGT) {
        if (own && s >= 0) tw = wt[src * 4 + (bitpos >> 6)];
      }

      // Acceptor.scala:192: phase2a.round < round -> Nack ; else vote
      const uint32_t tn = deliver ? (own & (uint32_t)((tw >> (bitpos & 63)) & 0xFull)) : 0u;
      uint32_t acc = 0, nck = 0;
      int nr = -1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool tk = (tn >> k) & 1u;
        const bool ok = tk && (rnd >= thr[k]);
        acc |= ok ? (1u << k) : 0u;
        if (tk && !ok) {
          nck |= 1u << k;
          nr = thr[k] > nr ? thr[k] : nr;
        }
      }
      // Acceptor.scala:204-208: round = phase2a.round; states(slot) = State(round, value)
      const bool full_cell = (acc | (~own & 0xFu)) == 0xFu;
      // does any acceptor of my 64-byte sector (4 lanes x 16 B) vote?  A fresh sector nobody votes in stays as it is
      // (contiguous target runs would otherwise write twice the bytes); one somebody votes in is written whole
      uint32_t sector_acc = 0;
      if constexpr (TGT) {
        sector_acc = acc | (uint32_t)__shfl_xor((int)acc, 1);
        sector_acc |= (uint32_t)__shfl_xor((int)sector_acc, 2);
      }
      if (TGT && fresh && deliver && own && sector_acc) {
        // first votes of the slot: cells whose acceptor does not vote hold -1 / -1.  ONE store path for the whole
        // row (fully voted cells included): two half-masked store instructions per array cost the issue
        // slots of two full ones
        const size_t ps = (size_t)phys_cur, row = ps * (size_t)g.RS + (size_t)r0, vrow = ps * (size_t)g.VS + (size_t)r0;
        int4v rr, vv, nb = thr;
        bool ballot_moves = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool a = (acc >> k) & 1u;
          rr[k] = a ? rnd : -1, vv[k] = a ? val : -1;
          if (a) ballot_moves = ballot_moves || thr[k] != rnd, nb[k] = rnd;
        }
        row_store(rr, reinterpret_cast<int4v*>(st.vote_round + vrow));
        row_store(vv, reinterpret_cast<int4v*>(st.vote_value + vrow));
        if (PERSLOT && ballot_moves) row_store(nb, reinterpret_cast<int4v*>(st.ballot + row));
      } else if (acc) {
        const size_t ps = (size_t)phys_cur, row = ps * (size_t)g.RS + (size_t)r0, vrow = ps * (size_t)g.VS + (size_t)r0;
        // whole-lane fast path: every acceptor this lane owns voted (padding cells may be overwritten)
        if (VEC && full_cell) {
          const int4v rr = {rnd, rnd, rnd, rnd};
          const int4v vv = {val, val, val, val};
          row_store(rr, reinterpret_cast<int4v*>(st.vote_round + vrow));
          row_store(vv, reinterpret_cast<int4v*>(st.vote_value + vrow));
          if (PERSLOT) {
            if (((own & 1u) && thr[0] != rnd) || ((own & 2u) && thr[1] != rnd) || ((own & 4u) && thr[2] != rnd) ||
                ((own & 8u) && thr[3] != rnd))
              row_store(rr, reinterpret_cast<int4v*>(st.ballot + row));
          }
        } else if (RMW) {
          // some of the lane's acceptors voted (thrifty delivery to a random f+1): read-modify-write of whole
          // 16-byte cells keeps the traffic full-line -- 4-byte stores leave the L2 with partially written
          // sectors that cost a DRAM read-modify-write each at eviction (+15 % on random f+1 of 255,
          // profiles/r01_thrifty.txt).  Its own instantiation: the load in the step costs registers (one
          // wave less per SIMD) and a wait that contiguous or dense targets do not want.
          int4v* pr = reinterpret_cast<int4v*>(st.vote_round + vrow);
          int4v* pv = reinterpret_cast<int4v*>(st.vote_value + vrow);
          int4v orr = *pr, ovv = *pv;
          int4v nb = {thr[0], thr[1], thr[2], thr[3]};
          bool ballot_moves = false;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (acc >> k & 1u) {
              orr[k] = rnd, ovv[k] = val;
              ballot_moves = ballot_moves || thr[k] != rnd;
              nb[k] = rnd;
            }
          }
          row_store(orr, pr);
          row_store(ovv, pv);
          if (PERSLOT && ballot_moves) row_store(nb, reinterpret_cast<int4v*>(st.ballot + row));
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (acc >> k & 1u) {
              st.vote_round[vrow + k] = rnd;
              st.vote_value[vrow + k] = val;
              if (PERSLOT && thr[k] != rnd) st.ballot[row + k] = rnd;
            }
          }
        }
      }
      // maxVotedSlot (Acceptor.scala:209) and the acceptor's new round.  When the WHOLE group voted
      // (the steady state) the maxima are the same for every acceptor: two wave-uniform scalars.
      const bool whole_group = (G == 64) && one_group && __all(acc == own);
      // G = 1 with several acceptor groups: a step whose 64 slots belong to ONE group (a leader group's batch; with
      // leader-group-major rows the batches of a launch are best sent that way) would send 64 lanes to the same LDS
      // words, 2 R serialized atomics each.  Fold such a step in registers; lane 63 alone touches the table.
      bool folded = false;
      if constexpr (G == 1) {
        if (!one_group) {
          const uint64_t have = __ballot(acc != 0);
          if (have) {
            const int g0 = __shfl(grp_cur, (int)__ffsll((unsigned long long)have) - 1);
            if (__all(acc == 0 || grp_cur == g0)) {
              folded = true;
              table_used = true;
              const int e = g0 * g.R;
              if (__all(acc == 0 || acc == own)) {  // every acceptor of the group voted wherever one did
                // (lane k raises acceptor k's words: one pass instead of R passes of lane 63 alone -- every pass costs
                // the wavefront its issue slots whoever is active)
                const int ms = __builtin_amdgcn_readlane(wave_max_to_lane63(acc ? s + 1 : 0), 63);
                const int mr = __builtin_amdgcn_readlane(wave_max_to_lane63(acc ? rnd + 1 : 0), 63);
                if (lane < g.R) atomicMax(&tab_mv[e + lane], ms - 1), atomicMax(&tab_pr[e + lane], mr - 1);
              } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  const bool a = (acc >> k) & 1u;
                  const int ms = wave_max_to_lane63(a ? s + 1 : 0), mr = wave_max_to_lane63(a ? rnd + 1 : 0);
                  if (lane == 63 && ms > 0) atomicMax(&tab_mv[e + k], ms - 1), atomicMax(&tab_pr[e + k], mr - 1);
                }
              }
            }
          }
        }
      }
      if (whole_group) {
        w_slot = s > w_slot ? s : w_slot;
        w_round = rnd > w_round ? rnd : w_round;
      } else if (acc && !folded) {
        if (one_group) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (acc >> k & 1u) {
              acc_mv[k] = s > acc_mv[k] ? s : acc_mv[k];
              acc_pr[k] = rnd > acc_pr[k] ? rnd : acc_pr[k];
            }
          }
        } else {
          table_used = true;
          const int e = grp_cur * g.R + r0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (acc >> k & 1u) {
              atomicMax(&tab_mv[e + k], s);
              atomicMax(&tab_pr[e + k], rnd);
            }
          }
        }
      }

      // ---- the slot's vote bitmap, in registers -----------------------------------------------
      uint64_t vb[4];
      assemble_bits<G>(acc, lane, g.base, vb);
      int nrm = -1;
      const bool any_nack = __any(nck != 0u);  // wave-uniform: the Nack reductions are rare
      if (any_nack) nrm = group_max<G>(nr);

      if constexpr (!FUSED) {
        uint64_t nb[4] = {0, 0, 0, 0};
        if (b.nack_bits && any_nack) assemble_bits<G>(nck, lane, g.base, nb);
        if (gi == 0) {
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            wo->votes[src][w] = vb[w];
            wo->nacks[src][w] = nb[w];
          }
          wo->nack_round[src] = nrm;
        }
      } else {
        // ProxyLeader.scala:235-256: record votes, test the quorum, Chosen exactly once
        bool ch = false;
        const int way = pick(myway, src);
        if (deliver) {
          uint64_t x[4];
#pragma unroll
          for (int w = 0; w < 4; ++w) x[w] = vb[w] & g.member[w];
          ch = is_write_quorum(g, x);
          if (!ch && gi == 0) {  // stays Pending: keep the votes (the key word is written below)
            const size_t e = (size_t)phys_cur * g.wp + way;
#pragma unroll
            for (int w = 0; w < 4; ++w) st.pl_bits[e * 4 + w] = x[w];
          }
        }
        if (gi == 0) {
          wo->chosen[src] = ch ? 1 : 0;
          wo->nack_round[src] = nrm;
        }
      }
#if FPX_PREFETCH
      s_cur = s_nxt, grp_cur = grp_nxt, phys_cur = phys_nxt, thr_cur = thr_nxt;
#else
      if (t + 1 < G) thr_cur = load_thr(t + 1, s_cur, grp_cur, phys_cur);
#endif
    }
    }  // (the walk, one row per step)

    // ---- outputs of the 64 messages leave as coalesced lines ------------------------------------
    wave_lds_sync();
    // the slot's row is no longer known to be all -1 (marked even if every acceptor Nacked: that only costs the shortcut)
    if (TGT ? (myfresh && mydeliver) : (mv && mydeliver)) st.row_voted[myphys] = 1;
    if (mv) {
      if constexpr (!FUSED) {
        if (b.vote_bits) {
          uint64_t* o = b.vote_bits + (size_t)m * 4;
#pragma unroll
          for (int w = 0; w < 4; ++w) o[w] = wo->votes[lane][w];
        }
        if (b.nack_bits) {
          uint64_t* o = b.nack_bits + (size_t)m * 4;
#pragma unroll
          for (int w = 0; w < 4; ++w) o[w] = wo->nacks[lane][w];
        }
      } else {
        const bool ch = wo->chosen[lane] != 0;
        if (mydeliver) {
          // states(slotround) = Done (ProxyLeader.scala:256) or Pending(phase2a, votes) (:213)
          const size_t e = (size_t)myphys * g.wp + myway;
          st.pl_key[e] = ((uint32_t)myround + 1u) | (ch ? KEY_DONE : 0u);
          if (!ch) st.pl_value[e] = myvalue;
        }
        if (quad) {  // parked: chosen flag and nack_round; the round and the value are still there
          sc[0 * 256 + cc * 64 + lane] = ((wo->nack_round[lane] + 1) << 1) | (ch ? 1 : 0);
        } else {
          if (b.chosen) b.chosen[m] = ch ? 1 : 0;
          if (b.chosen_round) b.chosen_round[m] = ch ? myround : -1;
          if (b.chosen_value) b.chosen_value[m] = ch ? myvalue : -1;
        }
      }
      if (b.nack_round && !(FUSED && quad)) b.nack_round[m] = wo->nack_round[lane];
    }
    wave_lds_sync();
    }  // columns of the unit
    if (FUSED && quad) {  // the four results of my row, 16 bytes per output array
      int4 cr, cv, nr;
      uint32_t chb = 0;
      int* crp = &cr.x;
      int* cvp = &cv.x;
      int* nrp = &nr.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int w = sc[0 * 256 + k * 64 + lane];
        const bool ch = w & 1;
        chb |= (ch ? 1u : 0u) << (8 * k);
        crp[k] = ch ? sc[1 * 256 + k * 64 + lane] : -1;
        cvp[k] = ch ? sc[2 * 256 + k * 64 + lane] : -1;
        nrp[k] = (w >> 1) - 1;
      }
      if (b.chosen) *reinterpret_cast<uint32_t*>(b.chosen + mbase) = chb;
      if (b.chosen_round) *reinterpret_cast<int4*>(b.chosen_round + mbase) = cr;
      if (b.chosen_value) *reinterpret_cast<int4*>(b.chosen_value + mbase) = cv;
      if (b.nack_round) *reinterpret_cast<int4*>(b.nack_round + mbase) = nr;
      wave_lds_sync();
    }
  }

  // ---- fold maxima --------------------------------------------------------------------------------
  // (a) steps in which the WHOLE group voted only raised two wave-uniform scalars: one pair of
  //     atomics per wavefront into a 64-way sharded table (the common case: nothing else to do);
  // (b) everything else went through registers / LDS tables: the workgroup appends ONE row to the
  //     partial table, claimed with a counter, only if it saw such a step.
  const int par = b.parity;
  if (lane == 0 && w_slot >= 0) {  // wave -> workgroup (LDS)
    atomicMax(&blk_flag[2], w_round);
    atomicMax(&blk_flag[3], w_slot);
  }
  if constexpr (PACK) {
    if (__any(pk_used)) {  // cell gi's maxima sit in the lanes gi & 31 of both halves, in their low or high set
      const int la = lane & 31, lb = 32 + (lane & 31);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int m0 = __shfl(pk_mv[k], la), m1 = __shfl(pk_mv[k], lb), m2 = __shfl(pk_mv[4 + k], la), m3 = __shfl(pk_mv[4 + k], lb);
        const int p0 = __shfl(pk_pr[k], la), p1 = __shfl(pk_pr[k], lb), p2 = __shfl(pk_pr[4 + k], la), p3 = __shfl(pk_pr[4 + k], lb);
        const int mm = lane < 32 ? (m0 > m1 ? m0 : m1) : (m2 > m3 ? m2 : m3), pp = lane < 32 ? (p0 > p1 ? p0 : p1) : (p2 > p3 ? p2 : p3);
        acc_mv[k] = mm > acc_mv[k] ? mm : acc_mv[k], acc_pr[k] = pp > acc_pr[k] ? pp : acc_pr[k];
      }
    }
  }
  bool any_table = false;
  if (one_group) {
    // the 64 / G lanes with the same gi hold maxima of the same acceptors: fold them across the wavefront first, one
    // lane per acceptor quad touches the LDS table (at G = 1 all 256 lanes of the workgroup would otherwise queue on
    // the same 2 R words: 12 us of a 19 us kernel on a 65 536 x 3 step, profiles/r03_small_n.txt)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (G == 1) {
        acc_mv[k] = wave_max_to_lane63(acc_mv[k] + 1) - 1, acc_pr[k] = wave_max_to_lane63(acc_pr[k] + 1) - 1;
      } else if constexpr (G < 64) {
#pragma unroll
        for (int m = G; m < 64; m <<= 1) {
          const int a = __shfl_xor(acc_mv[k], m), c = __shfl_xor(acc_pr[k], m);
          acc_mv[k] = a > acc_mv[k] ? a : acc_mv[k], acc_pr[k] = c > acc_pr[k] ? c : acc_pr[k];
        }
      }
    }
    const bool folder = G == 1 ? lane == 63 : q == 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (folder && (own >> k & 1u)) {
        if (acc_mv[k] >= 0) atomicMax(&tab_mv[r0 + k], acc_mv[k]), any_table = true;
        if (acc_pr[k] >= 0) atomicMax(&tab_pr[r0 + k], acc_pr[k]), any_table = true;
      }
    }
  } else {
    any_table = table_used;
  }
  if (any_table) blk_flag[0] = 1;
  __syncthreads();
  if (b.solo) {
    // a small batch (a tick of a few hundred messages is the reference's everyday case): the launch is this one
    // workgroup, so its LDS tables ARE the launch's maxima -- what k_finalize would fold out of the partial rows.
    // One launch instead of two: ~16 -> ~9 us per fused step (profiles/r03_small_n.txt)
    const int wr = blk_flag[2], ws = blk_flag[3];
    int32_t* top = g.per_slot ? st.max_ballot : st.promised;
    for (int e = threadIdx.x; e < ntab; e += blockDim.x) {
      int pr = tab_pr[e], mv = tab_mv[e];
      if (g.ngroups == 1) pr = wr > pr ? wr : pr, mv = ws > mv ? ws : mv;
      if (pr > top[e]) top[e] = pr;
      if (mv > st.max_voted[e]) st.max_voted[e] = mv;
    }
    return;
  }
  if (threadIdx.x == 0 && blk_flag[3] >= 0) {  // workgroup -> one of 64 cache lines (2 atomics per workgroup)
    int32_t* pa = st.part_all + ((size_t)par * 64 + (blockIdx.x & 63)) * PART_ALL_STRIDE;
    atomicMax(&pa[0], blk_flag[2]);
    atomicMax(&pa[1], blk_flag[3]);
  }
  if (blk_flag[0] && (int)blockIdx.x < g.part_rows) {  // (the grid never exceeds part_rows: never write out of bounds)
    int32_t* prow = st.part + (size_t)blockIdx.x * 2 * ntab;
    for (int i = threadIdx.x; i < 2 * ntab; i += blockDim.x) prow[i] = tab_pr[i];
    if (threadIdx.x == 0) st.part_stamp[blockIdx.x] = b.launch_seq;
  }
}

// ------------------------------------------------------------------------------------------------
// k_finalize: promised[e] = max(promised[e], max_b part[b][0][e]); max_voted likewise.
// grid.x = ceil(ntab / 64); block = 256 threads = 64 entries x 4 slices of the block axis.
// ------------------------------------------------------------------------------------------------
// grid = (ceil(ntab / 64), FINALIZE_SLICES): blockIdx.y strides over the rows of the partial table,
// the 4 waves of a block stride within that; one atomicMax per (entry, blockIdx.y) that improves.
constexpr int FINALIZE_SLICES = 8;  // at least; the launch uses more for big grids (about 8 rows per wavefront)
__global__ void __launch_bounds__(256) k_finalize(const Geom g, const State st, int par, int grid, uint32_t seq) {
  __shared__ int32_t red[2][4][64];
  // the buffers of the OTHER parity are used by the next launch: clear them here, whatever happens
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    if (threadIdx.x < 128)
      st.part_all[((size_t)(par ^ 1) * 64 + (threadIdx.x >> 1)) * PART_ALL_STRIDE + (threadIdx.x & 1)] = -1;
  }
  if (st.status[ST_ABORT] != 0) return;
  const int ntab = g.ngroups * g.R;
  const int nblocks = grid < g.part_rows ? grid : g.part_rows;
  const int e = blockIdx.x * 64 + (threadIdx.x & 63);
  const int slice = threadIdx.x >> 6;
  int pr = -1, mvs = -1;
  if (blockIdx.y == 0 && slice == 0 && g.ngroups == 1) {
    // steps in which the whole group voted (k_phase2 (a)): the same maxima for every acceptor
    const int32_t* pa = st.part_all + (size_t)par * 64 * PART_ALL_STRIDE;
    for (int i = 0; i < 64; ++i) {
      pr = pa[i * PART_ALL_STRIDE] > pr ? pa[i * PART_ALL_STRIDE] : pr;
      mvs = pa[i * PART_ALL_STRIDE + 1] > mvs ? pa[i * PART_ALL_STRIDE + 1] : mvs;
    }
  }
  if (e < ntab) {
#pragma unroll 4
    for (int bl = blockIdx.y * 4 + slice; bl < nblocks; bl += 4 * (int)gridDim.y) {
      // stamp and row are loaded together (no dependent chain); a row of another launch is simply not used
      const bool mine = st.part_stamp[bl] == seq;  // that workgroup of THIS launch used the tables
      const int32_t* prow = st.part + (size_t)bl * 2 * ntab;
      const int a = prow[e], c = prow[ntab + e];
      pr = (mine && a > pr) ? a : pr;
      mvs = (mine && c > mvs) ? c : mvs;
    }
  }
  red[0][slice][threadIdx.x & 63] = pr;
  red[1][slice][threadIdx.x & 63] = mvs;
  __syncthreads();
  if (slice == 0 && e < ntab) {
    for (int k = 1; k < 4; ++k) {
      pr = red[0][k][threadIdx.x] > pr ? red[0][k][threadIdx.x] : pr;
      mvs = red[1][k][threadIdx.x] > mvs ? red[1][k][threadIdx.x] : mvs;
    }
    // the accepted rounds raise Acceptor.round -- or, with a ballot per cell, the bound on the acceptor's ballots
    int32_t* top = g.per_slot ? st.max_ballot : st.promised;
    if (pr > top[e]) atomicMax(&top[e], pr);
    if (mvs > st.max_voted[e]) atomicMax(&st.max_voted[e], mvs);
  }
}

// ------------------------------------------------------------------------------------------------
// k_open: ProxyLeader.handlePhase2a bookkeeping (ProxyLeader.scala:175-184, 213). Thread / message.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_open(const Geom g, const State st, const Batch b) {
  if (st.status[ST_ABORT] != 0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.n) return;
  const int s = b.slot[i], rnd = b.round[i];
  const size_t ps = (size_t)phys_slot(g, s);
  uint32_t* kr = st.pl_key + ps * g.wp;
  const uint32_t want = (uint32_t)rnd + 1u;
  int way = -1;
  bool dup = false;
  for (int w = g.ways - 1; w >= 0; --w) {
    const uint32_t k = kr[w];
    dup = dup || ((k & KEY_ROUND_MASK) == want);
    if (k == 0) way = w;
  }
  uint8_t fresh = 0;
  if (!dup) {
    if (way < 0) {
      report(st, 5, i, s, rnd);
    } else {
      const size_t e = ps * g.wp + way;
      st.pl_key[e] = want;
      st.pl_value[e] = b.value[i];
#pragma unroll
      for (int w = 0; w < 4; ++w) st.pl_bits[e * 4 + w] = 0ull;
      fresh = 1;
    }
  }
  if (b.is_new) b.is_new[i] = fresh;
}

// ------------------------------------------------------------------------------------------------
// k_tally (K2): ProxyLeader.handlePhase2b (ProxyLeader.scala:217-258). Thread / message; the slots
// of a run are distinct so a tally entry has exactly one writer.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_tally(const Geom g, const State st, const Batch b) {
  if (st.status[ST_ABORT] != 0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.n) return;
  const int s = b.slot[i], rnd = b.round[i];
  uint64_t in[4];
  const uint64_t* vin = b.vote_bits + (size_t)i * 4;
#pragma unroll
  for (int w = 0; w < 4; ++w) in[w] = vin[w] & g.member[w];
  uint8_t ch = 0;
  int cr = -1, cv = -1;
  if ((in[0] | in[1] | in[2] | in[3]) != 0) {
    const size_t ps = (size_t)phys_slot(g, s);
    const uint32_t* kr = st.pl_key + ps * g.wp;
    const uint32_t want = (uint32_t)rnd + 1u;
    int way = -1;
    uint32_t key = 0;
    for (int w = 0; w < g.ways; ++w) {
      const uint32_t k = kr[w];
      if ((k & KEY_ROUND_MASK) == want) way = w, key = k;
    }
    if (way < 0) {
      report(st, 2 /*FPX_EFATAL_UNKNOWN_SLOTROUND*/, i, s, rnd);  // :220-225
    } else if (!(key & (KEY_DONE | KEY_RANGE))) {  // Done -> ignored, :227-232; a pending noop range
                                                   // under the same key -> ignored, mencius :327-333
      const size_t e = ps * g.wp + way;
      uint64_t x[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) x[w] = st.pl_bits[e * 4 + w] | in[w];  // :237
      if (is_write_quorum(g, x)) {                                        // :238-243
        ch = 1;
        cr = rnd;
        cv = st.pl_value[e];  // :246-253  Chosen(slot, pending.phase2a.value)
        st.pl_key[e] = key | KEY_DONE;  // :256
      } else {
#pragma unroll
        for (int w = 0; w < 4; ++w) st.pl_bits[e * 4 + w] = x[w];
      }
    }
  }
  if (b.chosen) b.chosen[i] = ch;
  if (b.chosen_round) b.chosen_round[i] = cr;
  if (b.chosen_value) b.chosen_value[i] = cv;
}

// ------------------------------------------------------------------------------------------------
// Phase1a (Acceptor.scala:148-182)
// ------------------------------------------------------------------------------------------------
// ACCEPTOR mode: ONE block, one thread per acceptor of the group (R <= 256).  out[0..3] promised bits,
// out[4..7] nack bits, assembled in LDS and written whole (no zeroing pass before the launch)
__global__ void __launch_bounds__(256) k_phase1a_scalar(const Geom g, const State st, int group, int round,
                                                        const uint64_t* target, uint64_t* out) {
  __shared__ unsigned long long bits[8];
  const int r = threadIdx.x;
  if (r < 8) bits[r] = 0ull;
  __syncthreads();
  if (r < g.R) {
    const int bit = g.base + r;
    if (!target || ((target[bit >> 6] >> (bit & 63)) & 1ull)) {
      int* pr = &st.promised[(size_t)group * g.R + r];
      if (round < *pr) {  // :155 Nack
        atomicOr(&bits[4 + (bit >> 6)], 1ull << (bit & 63));
      } else {            // :166 round = phase1a.round
        *pr = round;
        atomicOr(&bits[bit >> 6], 1ull << (bit & 63));
      }
    }
  }
  __syncthreads();
  if (r < 8) out[r] = bits[r];
}

// PER_SLOT mode, Acceptor.handlePhase1a generalised to a ballot per cell: every cell of the group at or above the
// watermark becomes max(old, round), and the acceptor Nacks iff one of them was ahead (old > round).
// k_p1a_decide (one thread per acceptor): if nothing of this acceptor can be ahead (max_ballot <= round) the whole
// Phase1a is ONE lazy record (round, watermark) -- O(R) instead of a sweep over S x R cells; an older record whose
// range starts below the new watermark is first written into the cells it alone covers (mode 1).  Otherwise
// (a stale Phase1a) the acceptor's column is checked cell by cell (mode 2).  k_p1a_sweep does that work and
// returns at once when no acceptor asked for any.
enum { P1_MODE = 0, P1_A = 1, P1_B = 2, P1_C = 3 };
// ONE block (R <= 256).  Writes the reply bitmaps itself for every acceptor that needs no sweep: out[0..3] promised
// bits, out[4..7] nack bits (zero here; a stale Phase1a's Nacks and promises come from the tail of k_p1a_sweep).
__global__ void __launch_bounds__(256) k_p1a_decide(const Geom g, const State st, int group, int round, int watermark,
                                                    const uint64_t* target, uint64_t* out) {
  __shared__ unsigned long long bits[4];
  const int r = threadIdx.x;
  if (r < 4) bits[r] = 0ull;
  __syncthreads();
  if (r < g.R) {
    int32_t* mode = st.p1 + P1_MODE * g.R;
    mode[r] = 0;
    const int bit = g.base + r;
    if (!target || ((target[bit >> 6] >> (bit & 63)) & 1ull)) {
      const size_t e = (size_t)group * g.R + r;
      const int wm = watermark < 0 ? 0 : watermark;
      const int lr = st.lz_round[e], lf = st.lz_from[e];
      if (st.max_ballot[e] <= round) {
        if (lr >= 0 && wm > lf) {  // the cells of [lf, wm) keep the older promise: make it explicit there
          mode[r] = 1;
          st.p1[P1_A * g.R + r] = lf, st.p1[P1_B * g.R + r] = wm, st.p1[P1_C * g.R + r] = lr;
          st.p1[4 * g.R] = 1;
        }
        st.lz_round[e] = round, st.lz_from[e] = wm;
        st.max_ballot[e] = round;
        atomicOr(&bits[bit >> 6], 1ull << (bit & 63));  // promised: nothing of this acceptor was ahead
      } else {
        mode[r] = 2;  // possibly stale: its column is checked cell by cell
        st.p1[4 * g.R] = 1;
      }
    }
  }
  __syncthreads();
  if (r < 4) out[r] = bits[r], out[4 + r] = 0ull;
}

__global__ void __launch_bounds__(256) k_p1a_sweep(const Geom g, const State st, int group, int round, int watermark,
                                                   uint64_t* out) {
  if (st.p1[4 * g.R] == 0) return;  // the common case: nothing to sweep
  const int32_t* mode = st.p1 + P1_MODE * g.R;
  const int wm = watermark < 0 ? 0 : watermark;
  const size_t ncell = (size_t)g.S * g.RS;
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncell; c += (size_t)gridDim.x * blockDim.x) {
    const int s = slot_of_row(g, (int)(c / g.RS)), r = (int)(c % g.RS);
    if (r >= g.R) continue;
    const int m = mode[r];
    if (m == 0 || group_of_slot(g, s) != group) continue;
    const int cur = st.ballot[c];
    if (m == 1) {
      if (s >= st.p1[P1_A * g.R + r] && s < st.p1[P1_B * g.R + r] && st.p1[P1_C * g.R + r] > cur)
        st.ballot[c] = st.p1[P1_C * g.R + r];
    } else if (s >= wm) {
      const size_t e = (size_t)group * g.R + r;
      const int lz = s >= st.lz_from[e] ? st.lz_round[e] : -1;
      const int eff = lz > cur ? lz : cur;
      if (eff > round) {
        // one bit per acceptor: test before the atomic, or a stale Phase1a that every cell Nacks would queue
        // S x R same-address atomics
        const int bit = g.base + r;
        unsigned long long* word = (unsigned long long*)&out[4 + (bit >> 6)];
        if (!((__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (bit & 63)) & 1ull))
          atomicOr(word, 1ull << (bit & 63));
      } else if (cur != round) {
        st.ballot[c] = round;
      }
    }
  }
  // the block that finishes last: the swept acceptors promise unless one of their cells was ahead; re-arm the flag
  __shared__ int last;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(&st.p1[4 * g.R + 1], 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  const int r = threadIdx.x;
  if (r < g.R && mode[r] == 2) {
    const int bit = g.base + r;
    const unsigned long long nack = __hip_atomic_load((unsigned long long*)&out[4 + (bit >> 6)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!((nack >> (bit & 63)) & 1ull)) atomicOr((unsigned long long*)&out[bit >> 6], 1ull << (bit & 63));
  }
  if (r == 0) st.p1[4 * g.R] = 0, st.p1[4 * g.R + 1] = 0;
}

// every outstanding lazy promise written into the cells it covers, the records cleared (readback / digests /
// fpx_acceptor_flush_promises)
__global__ void __launch_bounds__(256) k_lazy_flush(const Geom g, const State st) {
  const size_t ncell = (size_t)g.S * g.RS;
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncell; c += (size_t)gridDim.x * blockDim.x) {
    const int s = slot_of_row(g, (int)(c / g.RS)), r = (int)(c % g.RS);
    if (r >= g.R) continue;
    const size_t e = (size_t)group_of_slot(g, s) * g.R + r;
    const int lr = st.lz_round[e];
    if (lr >= 0 && s >= st.lz_from[e] && lr > st.ballot[c]) st.ballot[c] = lr;
  }
}
__global__ void k_lazy_clear(const Geom g, const State st) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < g.ngroups * g.R) st.lz_round[e] = -1, st.lz_from[e] = 0;
}

// ------------------------------------------------------------------------------------------------
// a5 standalone: n node sets -> isWriteQuorum / isReadQuorum (strict: foreign bit => status EINVAL)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_quorum_eval(const Geom g, int n, const uint64_t* nodes, int strict,
                                                     int read, uint8_t* out, int32_t* status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x[4];
  bool foreign = false;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const uint64_t v = nodes[(size_t)i * 4 + w];
    foreign = foreign || ((v & ~g.member[w]) != 0);
    x[w] = v & g.member[w];
  }
  if (strict && foreign) {
    if (atomicCAS(&status[ST_CODE], 0, 1) == 0) status[ST_INDEX] = i;
    out[i] = 0;
    return;
  }
  out[i] = (read ? is_read_quorum(g, x) : is_write_quorum(g, x)) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// readback helper: the log of one acceptor (a strided column of the SoA)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gather_acceptor(const Geom g, const State st, int group, int replica,
                                                         int32_t* vr, int32_t* vv, int32_t* bl) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= g.S) return;
  const bool mine = group_of_slot(g, s) == group;
  const size_t ps = (size_t)phys_slot(g, s), c = ps * g.RS + replica, vc = ps * g.VS + replica;
  vr[s] = mine ? st.vote_round[vc] : -1;
  vv[s] = mine ? st.vote_value[vc] : -1;
  bl[s] = (mine && st.ballot) ? st.ballot[c] : -1;
}

// Acceptor.maxVotedSlot restricted to a window of the log (the read path, multipaxos/Acceptor.scala:222-254): the largest
// slot of [first, first + count) of the acceptor's group in which it holds a vote, -1 if none.  The scalar max_voted is
// that maximum over ALL rows; a caller that maps an unbounded log onto the rows (row = slot % S, jni/Native.scala) needs
// it over the rows of one lap.  One strided column read; off the steady path.
__global__ void __launch_bounds__(256) k_max_voted_in(const Geom g, const State st, int group, int replica, int first, int count,
                                                      int32_t* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int best = -1;
  if (i < count) {
    const int s = first + i;
    if (group_of_slot(g, s) == group && st.vote_round[(size_t)phys_slot(g, s) * g.VS + replica] != -1) best = s;
  }
#pragma unroll
  for (int k = 1; k < 64; k <<= 1) {
    const int o = __shfl_xor(best, k);
    best = o > best ? o : best;
  }
  if ((threadIdx.x & 63) == 0 && best >= 0) atomicMax(out, best);
}

// ------------------------------------------------------------------------------------------------
// f1: the replica's log.  Replica.handleChosen (multipaxos/Replica.scala:572-590): a slot that is
// already in the log is ignored, otherwise log.put + numChosen += 1; executeLog (:394-404) advances
// executedWatermark over the contiguous prefix.
// ------------------------------------------------------------------------------------------------
// one atomicMin per wavefront (and none when it cannot lower the target): a log full of holes would otherwise
// send one same-address atomic per thread
__device__ __forceinline__ void wave_atomic_min(int32_t* target, int v) {
#pragma unroll
  for (int k = 1; k < 64; k <<= 1) {
    const int o = __shfl_xor(v, k);
    v = o < v ? o : v;
  }
  if ((threadIdx.x & 63) == 0 && v < __hip_atomic_load(target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMin(target, v);
}

enum { LG_WATERMARK = 0, LG_NUM_CHOSEN = 1, LG_LARGEST = 2, LG_FIRST_MISSING = 3, LG_RANGE_FIRST = 4 };

__global__ void __launch_bounds__(256) k_log_ingest(const Geom g, const State st, const Batch b) {
  if (st.status[ST_ABORT] != 0) return;
  // grid-stride, counts and the largest key accumulated per thread, then per workgroup: the two scalars see
  // one atomic pair per workgroup (a pair per wavefront serialised 2^15 same-address atomics per 2^20 records)
  __shared__ int w_cnt[4], w_top[4];
  int cnt = 0, top = -1;  // BufferMap.largestKey
  const int step = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < b.n; i += step) {
    if (b.mask && !b.mask[i]) continue;
    const int s = b.slot[i];
    if (!st.log_present[s]) {  // BufferMap.get == None
      st.log_value[s] = b.value[i];
      st.log_present[s] = 1;
      ++cnt;
      top = s > top ? s : top;
    }
  }
#pragma unroll
  for (int k = 1; k < 64; k <<= 1) {
    cnt += __shfl_xor(cnt, k);
    const int o = __shfl_xor(top, k);
    top = o > top ? o : top;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) w_cnt[w] = cnt, w_top[w] = top;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    int c = 0, t = -1;
    for (int j = 0; j < nw; ++j) {
      c += w_cnt[j];
      t = w_top[j] > t ? w_top[j] : t;
    }
    if (c) {
      atomicAdd(&st.log_scalars[LG_NUM_CHOSEN], c);
      atomicMax(&st.log_scalars[LG_LARGEST], t);
    }
  }
}

// scan range = [executedWatermark, min(S, largestKey + 1)): the slot after largestKey is absent by
// definition, so LG_FIRST_MISSING starts at the end of the range and only ever decreases
__global__ void k_log_prep(const Geom g, const State st) {
  if (st.status[ST_ABORT] != 0) return;
  const int hi = st.log_scalars[LG_LARGEST] + 1;
  st.log_scalars[LG_FIRST_MISSING] = hi < g.S ? hi : g.S;
}

__global__ void __launch_bounds__(256) k_log_scan(const Geom g, const State st) {
  if (st.status[ST_ABORT] != 0) return;
  const int lo = st.log_scalars[LG_WATERMARK];
  const int hi0 = st.log_scalars[LG_LARGEST] + 1;
  const int hi = hi0 < g.S ? hi0 : g.S;
  const int stride = gridDim.x * blockDim.x;
  int mine = 0x7fffffff;
  for (int s = lo + blockIdx.x * blockDim.x + threadIdx.x; s < hi; s += stride) {
    // nothing at or above the smallest hole found so far matters any more
    if (s >= __hip_atomic_load(&st.log_scalars[LG_FIRST_MISSING], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    if (!st.log_present[s]) {
      mine = s;
      break;  // later slots of this thread are larger
    }
  }
  wave_atomic_min(&st.log_scalars[LG_FIRST_MISSING], mine);
}

__global__ void k_log_commit(const State st) {
  if (st.status[ST_ABORT] != 0) return;
  const int fm = st.log_scalars[LG_FIRST_MISSING];
  if (fm > st.log_scalars[LG_WATERMARK]) st.log_scalars[LG_WATERMARK] = fm;
}

// mencius Replica.handleChosenNoopRange (mencius/Replica.scala:464-485): the slots start, start + L, ...
// below `end` get Noop in order UNTIL the first one that is already in the log -- there the reference
// handler returns (the rest of the range is dropped and executeLog is not run).  Position k <-> slot
// start + k * stride.  k_log_range_first: the smallest position already present (LG_RANGE_FIRST starts
// at count); k_log_range_fill: put Noop at the positions before it.
__global__ void __launch_bounds__(256) k_log_range_first(const State st, int start, int stride, int count) {
  if (st.status[ST_ABORT] != 0) return;
  const int step = gridDim.x * blockDim.x;
  int mine = 0x7fffffff;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += step) {
    if (k >= __hip_atomic_load(&st.log_scalars[LG_RANGE_FIRST], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    if (st.log_present[(size_t)start + (size_t)k * stride]) {
      mine = k;
      break;  // later positions of this thread are larger
    }
  }
  wave_atomic_min(&st.log_scalars[LG_RANGE_FIRST], mine);
}

__global__ void __launch_bounds__(256) k_log_range_fill(const State st, int start, int stride) {
  if (st.status[ST_ABORT] != 0) return;
  const int first = st.log_scalars[LG_RANGE_FIRST];
  const int step = gridDim.x * blockDim.x;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < first; k += step) {
    const size_t s = (size_t)start + (size_t)k * stride;
    st.log_value[s] = -1;  // Noop
    st.log_present[s] = 1;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && first > 0) {
    st.log_scalars[LG_NUM_CHOSEN] += first;
    const int top = start + (first - 1) * stride;  // BufferMap.largestKey
    if (top > st.log_scalars[LG_LARGEST]) st.log_scalars[LG_LARGEST] = top;
  }
}

// ------------------------------------------------------------------------------------------------
// f2: Leader.handlePhase1b recovery scan (multipaxos/Leader.scala:306-329, 543-566).
// k_quorum_max_slot: maxSlot over the quorum's acceptors; k_phase1b_scan<G>: per slot the arg-max
// voteRound over the quorum's acceptors of the slot's group (ties: lowest acceptor index).
// ------------------------------------------------------------------------------------------------
__global__ void k_quorum_max_slot(const Geom g, const State st, const uint64_t* qmask, int watermark, int32_t* out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= g.ngroups * g.R) return;
  const int grp = e / g.R, bit = g.base + e % g.R;
  if (!((qmask[(size_t)grp * 4 + (bit >> 6)] >> (bit & 63)) & 1ull)) return;
  const int mv = st.max_voted[e];
  if (mv >= watermark) atomicMax(out, mv);  // maxPhase1bSlot over info from chosenWatermark
}

template <int G>
__global__ void __launch_bounds__(256)
    k_phase1b_scan(const Geom g, const State st, const uint64_t* qmask, int watermark, int count, int vec,
                   int32_t* safe_round, int32_t* safe_value) {
  constexpr int Q = 64 / G;
  const int lane = threadIdx.x & 63;
  const int gi = lane & (G - 1), q = lane / G;
  const int r0 = 4 * gi;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int base = wave * Q; base < count; base += nwaves * Q) {
    const int idx = base + q;
    const bool live = idx < count;
    const int s = watermark + idx;
    int best_round = -1, best_val = -1, best_idx = 1 << 30;
    if (live && r0 < g.R) {
      const int grp = group_of_slot(g, s);
      const size_t row = (size_t)phys_slot(g, s) * g.VS + r0;
      int vr[4] = {-1, -1, -1, -1}, vv[4] = {-1, -1, -1, -1};
      if (vec) {
        const int4v a = *reinterpret_cast<const int4v*>(st.vote_round + row);
        const int4v c = *reinterpret_cast<const int4v*>(st.vote_value + row);
#pragma unroll
        for (int k = 0; k < 4; ++k) vr[k] = a[k], vv[k] = c[k];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (r0 + k < g.R) vr[k] = st.vote_round[row + k], vv[k] = st.vote_value[row + k];
      }
      const int bit = g.base + r0;
      const uint32_t qn = (uint32_t)((qmask[(size_t)grp * 4 + (bit >> 6)] >> (bit & 63)) & 0xFull);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // phase1b.info.find(_.slot == slot): only acceptors of the quorum that voted in the slot
        if ((qn >> k & 1u) && r0 + k < g.R && vr[k] > best_round) best_round = vr[k], best_val = vv[k], best_idx = r0 + k;
      }
    }
    // slotInfos.maxBy(_.voteRound) across the lanes of the slot
#pragma unroll
    for (int m = 1; m < G; m <<= 1) {
      const int orr = __shfl_xor(best_round, m), ov = __shfl_xor(best_val, m), oi = __shfl_xor(best_idx, m);
      if (orr > best_round || (orr == best_round && oi < best_idx)) best_round = orr, best_val = ov, best_idx = oi;
    }
    if (live && gi == 0) {
      safe_round[idx] = best_round;
      safe_value[idx] = best_round >= 0 ? best_val : -1;  // Noop when nobody voted (Leader.scala:323-325)
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_probe: the hot kernel's HBM access pattern and nothing else (one wavefront per 32 consecutive rows: read the
// ballot row, write the two vote rows, 16 B per lane), run by fpx_create on a freshly allocated slab BEFORE it is
// initialised.  The same kernel on the same box runs 7 % apart depending on where the slab happened to be
// placed (profiles/r02_placement.txt); create times a few placements and keeps the fastest.
// ------------------------------------------------------------------------------------------------
// rows = number of rows touched, in `pieces` equal runs spread evenly over the S rows of the arrays
__global__ void __launch_bounds__(256) k_probe(int32_t* a_read, int32_t* b_write, int32_t* c_write, int rows, int q4,
                                               int pieces, long long piece_stride) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int first = wave * 32;
  const int per_piece = rows / pieces;
  for (int i = first; i < first + 32 && i < rows; ++i) {
    if (lane >= q4) continue;
    const long long r = (long long)(i / per_piece) * piece_stride + i % per_piece;
    const size_t o = ((size_t)r * q4 + lane) * 4;
    int4v v = {i, i, i, i};
    if (a_read) v = *reinterpret_cast<const int4v*>(a_read + o);
    row_store(v, reinterpret_cast<int4v*>(b_write + o));
    if (c_write) row_store(v, reinterpret_cast<int4v*>(c_write + o));
  }
}

// ------------------------------------------------------------------------------------------------
// State digests (parity at full size): order-independent 64-bit sums of per-element hashes, so that
// the whole acceptor / proxy-leader / replica state of a 2^20 x 256 context can be compared with another
// implementation's (the formula is part of the ABI, include/fpx.h) without moving 3 GiB through PCIe.
// Reads every cell once: HBM-bound.
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finalizer
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t digest_term(uint64_t idx, int32_t v) {
  return mix64(idx * 0x9E3779B97F4A7C15ull + (uint64_t)(uint32_t)v + 1ull);
}

__device__ __forceinline__ void block_add_u64(uint64_t v, uint64_t* out) {
  __shared__ uint64_t part[4];
#pragma unroll
  for (int k = 1; k < 64; k <<= 1) v += shfl_xor64(v, k);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint64_t t = part[0] + part[1] + part[2] + part[3];
    if (t) atomicAdd(reinterpret_cast<unsigned long long*>(out), (unsigned long long)t);
  }
}

// a cell array [S][RS]: element (s, r), r < R, contributes digest_term(s * R + r, a[s][r])
__global__ void __launch_bounds__(256) k_digest_cells(const Geom g, const int32_t* a, int stride, uint64_t* out) {
  const size_t q = (size_t)(g.RS >> 2);  // int4's per row; row s starts at a + s * stride
  const size_t n4 = (size_t)g.S * q;
  uint64_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t prow = i / q, s = (size_t)slot_of_row(g, (int)prow);
    const int r0 = (int)(i - prow * q) * 4;
    const int4v v = *reinterpret_cast<const int4v*>(a + prow * (size_t)stride + (size_t)r0);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (r0 + k < g.R) acc += digest_term(s * (size_t)g.R + (size_t)(r0 + k), v[k]);
  }
  block_add_u64(acc, out);
}

__global__ void __launch_bounds__(256) k_digest_1d(const int32_t* a, int n, uint64_t* out) {
  uint64_t acc = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += digest_term((uint64_t)i, a[i]);
  block_add_u64(acc, out);
}

// ProxyLeader.states: every single-slot tally (slot, round) -> Done | Pending(value, votes); key order free
__global__ void __launch_bounds__(256) k_digest_tally(const Geom g, const State st, uint64_t* out) {
  uint64_t acc = 0;
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < g.S; s += gridDim.x * blockDim.x) {
    for (int w = 0; w < g.ways; ++w) {
      const size_t e = (size_t)phys_slot(g, s) * g.wp + w;
      const uint32_t k = st.pl_key[e];
      if (k == 0 || (k & KEY_RANGE)) continue;
      const uint32_t round = (k & KEY_ROUND_MASK) - 1u;
      const uint64_t done = (k & KEY_DONE) ? 1ull : 0ull;
      uint64_t t = mix64((((uint64_t)(uint32_t)s << 32) | round) * 0x9E3779B97F4A7C15ull + done);
      if (!done) {  // a Done entry has dropped its payload (ProxyLeader.scala:256)
        t = mix64(t ^ (uint64_t)(uint32_t)st.pl_value[e]);
#pragma unroll
        for (int j = 0; j < 4; ++j) t = mix64(t ^ st.pl_bits[e * 4 + j]);
      }
      acc += t;
    }
  }
  block_add_u64(acc, out);
}

// the replica's log: present entries (slot, value) + executedWatermark + numChosen
__global__ void __launch_bounds__(256) k_digest_log(const Geom g, const State st, uint64_t* out) {
  uint64_t acc = 0;
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < g.S; s += gridDim.x * blockDim.x)
    if (st.log_present[s]) acc += digest_term((uint64_t)s, st.log_value[s]);
  if (blockIdx.x == 0 && threadIdx.x == 0)
    acc += mix64((uint64_t)(uint32_t)st.log_scalars[0] * 0x9E3779B97F4A7C15ull + 7ull) +
           mix64((uint64_t)(uint32_t)st.log_scalars[1] * 0x9E3779B97F4A7C15ull + 11ull);
  block_add_u64(acc, out);
}

}  // namespace fpx
