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
#ifndef FPX_EARLY_THR
#define FPX_EARLY_THR 1  // the first step's ballot row is requested with the chunk's other loads (fpx_phase2_body.inc)
#endif
#ifndef FPX_NT
#define FPX_NT 1  // vote rows are written once and not re-read soon: nontemporal stores
#endif

#include "fpx_fastdiv.hpp"

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
  // slot / L and row / A by multiplication (fpx_fastdiv.hpp); magic 0 = divisor 1
  uint32_t l_magic, a_magic;
  int32_t l_shift, a_shift;
};

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
  int32_t* p1;          // [4][R] + 8     scratch of one Phase1a: mode / a / b / c per acceptor, then its control words (P1_SWEEP ..)
  uint8_t* row_voted;   // [S]            0 = no acceptor of the slot's group has ever voted in it (its cells
                        //                are all -1): partially voted cells can then be written whole without a read
  uint32_t* stamp;      // [S]            run id of the last run that touched the slot
  int32_t* run_round;   // [ngroups]      the single round of the current run per group (-1 = none)
  int32_t* status;      // [8]
  int32_t* part;        // [grid][2][ngroups*R] per-workgroup maxima rows (accepted round, voted slot)
  uint32_t* part_stamp; // [grid] the launch (Batch::launch_seq) that wrote row b of `part`: a workgroup that used the
                        //        tables writes row blockIdx.x -- no claiming counter (8192 same-address atomics were 65 us)
  int32_t* part_all;    // [3][64][PART_ALL_STRIDE] whole-group maxima (round, slot) + the largest round of any message, 64 lines, per launch counter mod 3
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
  int32_t parity;          // K1 / K3 launch counter mod 3: which third of part_all this launch uses
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

#define FPX_P2_NAME k_phase2
#define FPX_P2_EXTRA_PARAMS
#define FPX_P2_NBLK gridDim.x
#define FPX_P2_BID blockIdx.x
#define FPX_P2_PROLOGUE
#include "fpx_phase2_body.inc"
#undef FPX_P2_NAME
#undef FPX_P2_EXTRA_PARAMS
#undef FPX_P2_NBLK
#undef FPX_P2_BID
#undef FPX_P2_PROLOGUE

// the device's status words as a host-path call left them, into the call's page-locked words (what
// fpx_phase2_fused_wait reports): a 32-byte posted write instead of a copy-engine packet on the compute stream
__global__ void __launch_bounds__(64) k_status_snap(const int32_t* status, int32_t* out) {
  if (threadIdx.x < 8) out[threadIdx.x] = status[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// k_finalize: promised[e] = max(promised[e], max_b part[b][0][e]); max_voted likewise.
// grid.x = ceil(ntab / 64); block = 256 threads = 64 entries x 4 slices of the block axis.
// ------------------------------------------------------------------------------------------------
// grid = (ceil(ntab / 64), FINALIZE_SLICES): blockIdx.y strides over the rows of the partial table,
// the 4 waves of a block stride within that; one atomicMax per (entry, blockIdx.y) that improves.
constexpr int FINALIZE_SLICES = 8;  // at least; the launch uses more for big grids (about 8 rows per wavefront)
// (the body by workgroup coordinates: k_finalize is its own grid, k_ranges_fill_lg_fin -- fpx_ranges.hpp -- appends
// these workgroups to a Mencius band's fill)
// part_all has THREE buffers (launch counter mod 3).  The fold of launch k may run as late as INSIDE launch k + 1 (round 6: the
// deferred fold, k_phase2_fin below), which writes buffer k + 1; so the buffer a fold clears for reuse is k + 2's (last
// folded by launch k - 1's fold, which was complete before launch k + 1 began).  A fold that runs at once behind its
// launch clears the same one: launch k + 1's was cleared by the fold of launch k - 1.
// carried: the fold rides in the NEXT launch, whose validation may have set ST_ABORT -- that is no statement about the launch
// being folded (which, had IT been aborted, left nothing to fold: its shards read -1, none of its rows is stamped).
__device__ __forceinline__ void finalize_body(const Geom& g, const State& st, int par, int grid, uint32_t seq, int fx, int fy, int slices,
                                              bool carried = false) {
  __shared__ int32_t red[2][4][64];
  // the shards of the launch after next: clear them here, whatever happens
  if (fx == 0 && fy == 0) {
    const int clr = par >= 1 ? par - 1 : 2;  // (par + 2) % 3
    if (threadIdx.x < 128)
      st.part_all[((size_t)clr * 64 + (threadIdx.x >> 1)) * PART_ALL_STRIDE + (threadIdx.x & 1)] = -1;
    else if (threadIdx.x < 192)  // (word 2: the largest round a partial row of the launch holds, what k_p1a_fast reads)
      st.part_all[((size_t)clr * 64 + (threadIdx.x - 128)) * PART_ALL_STRIDE + 2] = -1;
  }
  if (!carried && st.status[ST_ABORT] != 0) return;
  const int ntab = g.ngroups * g.R;
  const int nblocks = grid < g.part_rows ? grid : g.part_rows;
  const int e = fx * 64 + (threadIdx.x & 63);
  const int slice = threadIdx.x >> 6;
  int pr = -1, mvs = -1;
  if (fy == 0 && slice == 0 && g.ngroups == 1) {
    // steps in which the whole group voted (k_phase2 (a)): the same maxima for every acceptor
    const int32_t* pa = st.part_all + (size_t)par * 64 * PART_ALL_STRIDE;
    for (int i = 0; i < 64; ++i) {
      pr = pa[i * PART_ALL_STRIDE] > pr ? pa[i * PART_ALL_STRIDE] : pr;
      mvs = pa[i * PART_ALL_STRIDE + 1] > mvs ? pa[i * PART_ALL_STRIDE + 1] : mvs;
    }
  }
  if (e < ntab) {
#pragma unroll 4
    for (int bl = fy * 4 + slice; bl < nblocks; bl += 4 * slices) {
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
__global__ void __launch_bounds__(256) k_finalize(const Geom g, const State st, int par, int grid, uint32_t seq) {
  finalize_body(g, st, par, grid, seq, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}

// ---- the fold of the launch BEFORE, as the first workgroups of this launch (round 6) --------------------------------------
// Every K1 / K3 launch used to be followed by k_finalize: a dependent launch of ~5 us and the gap in front of it -- 12 us
// of the headline's 560 us step, 9 of config 2's 13.  With a ballot per cell (FPX_BALLOT_PER_SLOT) no vote kernel reads what
// the fold writes (max_ballot, max_voted: Phase1a and the read-backs do), so the fold of launch k may ride in launch k + 1:
// k_phase2_fin = the vote kernel whose first fj.nblk workgroups fold the launch before (its shards of part_all, its half of
// the partial rows -- the two halves of `part` / `part_stamp` alternate from launch to launch, the State a launch is
// handed points at its own).  Anything else that touches the context first folds what is pending (DeviceGuard), so the
// deferral is not observable through the ABI.  FPX_NO_DEFER_FINALIZE=1 switches it off.
struct FinJob {
  int nblk;              // workgroups that fold (fgx * slices; 0: nothing pending)
  int fgx, slices, par, grid;
  uint32_t seq;
  int32_t* part;         // the folded launch's half of State::part / part_stamp
  uint32_t* part_stamp;
};
#define FPX_P2_NAME k_phase2_fin
#define FPX_P2_EXTRA_PARAMS , const FinJob fj
#define FPX_P2_NBLK (gridDim.x - fj.nblk)
#define FPX_P2_BID (blockIdx.x - fj.nblk)
#define FPX_P2_PROLOGUE                                                                                                   \
  if ((int)blockIdx.x < fj.nblk) {                                                                                        \
    State fs = st;                                                                                                        \
    fs.part = fj.part, fs.part_stamp = fj.part_stamp;                                                                     \
    finalize_body(g, fs, fj.par, fj.grid, fj.seq, (int)blockIdx.x % fj.fgx, (int)blockIdx.x / fj.fgx, fj.slices, true); \
    return;                                                                                                               \
  }
#include "fpx_phase2_body.inc"
#undef FPX_P2_NAME
#undef FPX_P2_EXTRA_PARAMS
#undef FPX_P2_NBLK
#undef FPX_P2_BID
#undef FPX_P2_PROLOGUE


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
// ACCEPTOR mode: ONE block, one thread per acceptor of the group (R <= 256).  outp[0..3] promised bits,
// outn[0..3] nack bits, assembled in LDS and written whole (no zeroing pass before the launch)
__global__ void __launch_bounds__(256) k_phase1a_scalar(const Geom g, const State st, int group, int round,
                                                        const uint64_t* target, uint64_t* outp, uint64_t* outn) {
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
  if (r < 4) outp[r] = bits[r], outn[r] = bits[4 + r];
}

// PER_SLOT mode, Acceptor.handlePhase1a generalised to a ballot per cell: every cell of the group at or above the
// watermark becomes max(old, round), and the acceptor Nacks iff one of them was ahead (old > round).
// k_p1a_decide (one thread per acceptor): if nothing of this acceptor can be ahead (max_ballot <= round) the whole
// Phase1a is ONE lazy record (round, watermark) -- O(R) instead of a sweep over S x R cells; an older record whose
// range starts below the new watermark is first written into the cells it alone covers (mode 1).  Otherwise
// (a stale Phase1a) the acceptor's column is checked cell by cell (mode 2).  k_p1a_sweep does that work and
// returns at once when no acceptor asked for any.
enum { P1_MODE = 0, P1_A = 1, P1_B = 2, P1_C = 3 };
// st.p1 + 4 R: the control words of a Phase1a
enum { P1_SWEEP = 0,      // a sweep is needed (set by the deciding workgroup, cleared by the sweep's last workgroup)
       P1_SWEPT = 1,      // sweep workgroups that are done
       P1_NACKS = 8 };    // k_p1a_fast: [4] 64-bit words, the Nack bits of a sweep while it runs
// ONE workgroup of 256 threads (R <= 256).  Writes the reply bitmaps itself for every acceptor that needs no sweep: outp[0..3]
// promised bits, outn[0..3] nack bits (zero here; a stale Phase1a's Nacks and promises come from the tail of the sweep).
__device__ __forceinline__ void p1a_decide_body(const Geom& g, const State& st, int group, int round, int watermark,
                                                const uint64_t* target, uint64_t* outp, uint64_t* outn) {
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
          st.p1[4 * g.R + P1_SWEEP] = 1;
        }
        st.lz_round[e] = round, st.lz_from[e] = wm;
        st.max_ballot[e] = round;
        atomicOr(&bits[bit >> 6], 1ull << (bit & 63));  // promised: nothing of this acceptor was ahead
      } else {
        mode[r] = 2;  // possibly stale: its column is checked cell by cell
        st.p1[4 * g.R + P1_SWEEP] = 1;
      }
    }
  }
  __syncthreads();
  if (r < 4) outp[r] = bits[r], outn[r] = 0ull;
}

// workgroup `bid` of `nblk` (256 threads each) of the sweep
__device__ __forceinline__ void p1a_sweep_body(const Geom& g, const State& st, int group, int round, int watermark, uint64_t* outp,
                                               uint64_t* outn, int bid, int nblk) {
  int32_t* ctl = st.p1 + 4 * g.R;
  if (ctl[P1_SWEEP] == 0) return;  // the common case: nothing to sweep
  const int32_t* mode = st.p1 + P1_MODE * g.R;
  const int wm = watermark < 0 ? 0 : watermark;
  const size_t ncell = (size_t)g.S * g.RS;
  for (size_t c = (size_t)bid * blockDim.x + threadIdx.x; c < ncell; c += (size_t)nblk * blockDim.x) {
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
        unsigned long long* word = (unsigned long long*)&outn[bit >> 6];
        if (!((__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (bit & 63)) & 1ull))
          atomicOr(word, 1ull << (bit & 63));
      } else if (cur != round) {
        st.ballot[c] = round;
      }
    }
  }
  // the workgroup that finishes last: the swept acceptors promise unless one of their cells was ahead; re-arm the flag
  __shared__ int last;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(&ctl[P1_SWEPT], 1) == nblk - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  const int r = threadIdx.x;
  if (r < g.R && mode[r] == 2) {
    const int bit = g.base + r;
    const unsigned long long nack = __hip_atomic_load((unsigned long long*)&outn[bit >> 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!((nack >> (bit & 63)) & 1ull)) atomicOr((unsigned long long*)&outp[bit >> 6], 1ull << (bit & 63));
  }
  if (r == 0) ctl[P1_SWEEP] = 0, ctl[P1_SWEPT] = 0;
}

// the three steps as launches of their own (FPX_P1A_SPLIT=1: rounds 2 - 5, with the fold of a pending launch in front)
__global__ void __launch_bounds__(256) k_p1a_decide(const Geom g, const State st, int group, int round, int watermark,
                                                    const uint64_t* target, uint64_t* outp, uint64_t* outn) {
  p1a_decide_body(g, st, group, round, watermark, target, outp, outn);
}
__global__ void __launch_bounds__(256) k_p1a_sweep(const Geom g, const State st, int group, int round, int watermark,
                                                   uint64_t* outp, uint64_t* outn) {
  p1a_sweep_body(g, st, group, round, watermark, outp, outn, (int)blockIdx.x, (int)gridDim.x);
}

// ---- Phase1a as ONE launch that does not wait for a pending fold (round 6) ------------------------------------------------
// A Phase1a on a context with a ballot per cell was the fold of the vote launch before it (the decision reads max_ballot,
// which that fold raises), k_p1a_decide (one workgroup) and k_p1a_sweep (which returns at its first instruction unless the
// Phase1a is stale for an acceptor or moves the watermark past an older lazy promise): three dependent launches of ~5 us,
// 28 times per pass of SURVEY 8(d)'s adversarial stream -- a fifth of its time (profiles/r06_phase1a.md).  Here:
//  * the decision does not need the fold: a vote launch leaves the largest round of any of its messages in its shards of
//    part_all (word 2, beside the whole-group maxima; k_phase2's epilogue), which bounds whatever its fold will raise --
//    max(max_ballot[e], that bound) > round sends the acceptor to the cell-by-cell check exactly as a folded max_ballot
//    would, or (the bound being loose) a little more often, and the check is exact.  So the pending fold stays pending and
//    rides in the NEXT vote launch as before (what it later does to max_ballot commutes with what the decision wrote);
//  * every workgroup takes the decision for all R acceptors itself (R + 128 words of reads) instead of waiting for one
//    that does: the test for "stale" reads only words the deciding workgroup never changes the verdict of (it overwrites
//    max_ballot[e] by `round` where max_ballot[e] <= round held), so there is nothing to wait for; workgroup 0 writes the
//    lazy records, and the reply bits when nobody is stale; when somebody is, all workgroups check those acceptors'
//    columns and the last one to finish writes the reply bits.
// The host takes this form only when the watermark does not pass an older lazy promise's start (it knows every watermark
// it ever handed in: fpx_ctx::lz_min_from) -- that case (P1 mode 1) needs the older records as they were and goes
// through the three launches.  (Also measured: the three steps in one launch behind each other, the sweep's workgroups
// waiting for a ticket of the deciding one -- 2.7 - 5.6 ms per pass against 1.9: polls across the XCDs' L2s see the ticket
// late, acquire fences per poll are cache invalidations; and fold + decision in one launch by a last-workgroup counter:
// no faster than two launches.)
__global__ void __launch_bounds__(256) k_p1a_fast(const Geom g, const State st, int par_pending, int group, int round, int watermark,
                                                  const uint64_t* target, uint64_t* outp, uint64_t* outn) {
  __shared__ unsigned long long prom[4], stale[4];
  __shared__ int bound_s, last;
  const int r = threadIdx.x;
  if (r < 4) prom[r] = 0ull, stale[r] = 0ull;
  if (r == 0) bound_s = -1;
  __syncthreads();
  if (par_pending >= 0 && r < 64) {  // what the vote launch whose fold is still pending can have raised, at most
    const int32_t* pa = st.part_all + ((size_t)par_pending * 64 + r) * PART_ALL_STRIDE;
    const int a = pa[0], c = pa[2];
    const int v = group_max<64>(a > c ? a : c);
    if (r == 0) bound_s = v;
  }
  __syncthreads();
  const int wm = watermark < 0 ? 0 : watermark;
  const int bit = g.base + r;
  const size_t e = (size_t)group * g.R + (r < g.R ? r : 0);
  bool tg = false, m2 = false;
  if (r < g.R && (!target || ((target[bit >> 6] >> (bit & 63)) & 1ull))) {
    tg = true;
    const int mb = __hip_atomic_load(&st.max_ballot[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    m2 = (mb > bound_s ? mb : bound_s) > round;
    atomicOr(m2 ? &stale[bit >> 6] : &prom[bit >> 6], 1ull << (bit & 63));
  }
  __syncthreads();
  const bool need = (stale[0] | stale[1] | stale[2] | stale[3]) != 0ull;
  if (blockIdx.x == 0) {
    if (tg && !m2) st.lz_round[e] = round, st.lz_from[e] = wm, st.max_ballot[e] = round;
    if (!need && r < 4) outp[r] = prom[r], outn[r] = 0ull;  // promised: nothing of these acceptors was ahead
  }
  if (!need) return;
  // a stale Phase1a for some acceptor: its column cell by cell (p1a_sweep_body's mode 2)
  int32_t* ctl = st.p1 + 4 * g.R;
  unsigned long long* nacks = reinterpret_cast<unsigned long long*>(ctl + P1_NACKS);  // [4], zero between Phase1a's
  const size_t ncell = (size_t)g.S * g.RS;
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncell; c += (size_t)gridDim.x * blockDim.x) {
    const int s = slot_of_row(g, (int)(c / g.RS)), rr = (int)(c % g.RS);
    if (rr >= g.R) continue;
    const int b2 = g.base + rr;
    if (!((stale[b2 >> 6] >> (b2 & 63)) & 1ull) || group_of_slot(g, s) != group || s < wm) continue;
    const size_t e2 = (size_t)group * g.R + rr;
    const int cur = st.ballot[c];
    const int lz = s >= st.lz_from[e2] ? st.lz_round[e2] : -1;
    const int eff = lz > cur ? lz : cur;
    if (eff > round) {
      if (!((__hip_atomic_load(&nacks[b2 >> 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (b2 & 63)) & 1ull))
        atomicOr(&nacks[b2 >> 6], 1ull << (b2 & 63));
    } else if (cur != round) {
      st.ballot[c] = round;
    }
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(&ctl[P1_SWEPT], 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (r < 4) {  // the swept acceptors promise unless one of their cells was ahead
    const unsigned long long nk = __hip_atomic_load(&nacks[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    outp[r] = prom[r] | (stale[r] & ~nk), outn[r] = nk;
    nacks[r] = 0ull;
  }
  if (r == 0) ctl[P1_SWEPT] = 0;
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
// behind the scalars: one (count, largest key) pair per workgroup of k_log_ingest, folded by k_log_prep
constexpr int LG_PARTS_AT = 8, LG_MAX_PARTS = 4096;

__global__ void __launch_bounds__(256) k_log_ingest(const Geom g, const State st, const Batch b) {
  if (st.status[ST_ABORT] != 0) return;
  // grid-stride, counts and the largest key accumulated per thread, then per workgroup; every workgroup leaves its pair
  // behind the scalars and k_log_prep folds them.  (Round 1: an atomic pair per wavefront -- 2^15 same-address atomics per
  // 2^20 records, 375 us; rounds 1 - 5: a pair per workgroup, still 2 x 2048 atomics on one cache line, ~35 of the kernel's
  // 52 us -- the effect that cost k_dp_keys 43 us, profiles/r06_depgraph_dev.md.)
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
    st.log_scalars[LG_PARTS_AT + 2 * blockIdx.x] = c;
    st.log_scalars[LG_PARTS_AT + 2 * blockIdx.x + 1] = t;
  }
}

// scan range = [executedWatermark, min(S, largestKey + 1)): the slot after largestKey is absent by
// definition, so LG_FIRST_MISSING starts at the end of the range and only ever decreases
// nparts > 0: behind k_log_ingest -- its workgroups' (count, largest key) pairs are folded into numChosen / largestKey first
__global__ void __launch_bounds__(256) k_log_prep(const Geom g, const State st, int nparts) {
  __shared__ int s_cnt[256], s_top[256];
  if (st.status[ST_ABORT] != 0) return;
  int c = 0, t = -1;
  for (int j = threadIdx.x; j < nparts; j += 256) {
    c += st.log_scalars[LG_PARTS_AT + 2 * j];
    const int o = st.log_scalars[LG_PARTS_AT + 2 * j + 1];
    t = o > t ? o : t;
  }
  s_cnt[threadIdx.x] = c, s_top[threadIdx.x] = t;
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (nparts > 0) {
    for (int j = 1; j < 256; ++j) {
      c += s_cnt[j];
      t = s_top[j] > t ? s_top[j] : t;
    }
    st.log_scalars[LG_NUM_CHOSEN] += c;
    if (t > st.log_scalars[LG_LARGEST]) st.log_scalars[LG_LARGEST] = t;
  }
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
