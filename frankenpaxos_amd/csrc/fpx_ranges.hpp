// fpx_ranges.hpp -- K4: Mencius noop ranges as batched gfx950 kernels.
//
//   mencius.Acceptor.handlePhase2aNoopRange      mencius/Acceptor.scala:237-291
//   mencius.ProxyLeader.handlePhase2aNoopRange   mencius/ProxyLeader.scala:255-303
//   mencius.ProxyLeader.handlePhase2bNoopRange   mencius/ProxyLeader.scala:355-411
//
// A Mencius leader that has nothing to propose skips its slots with ONE message per lagging stretch of the
// log: Phase2aNoopRange(slotStart, slotEnd, round) stands for Noop in every slot of [slotStart, slotEnd) that
// its leader group owns (slotStart + k * numLeaderGroups).  With 256 leader groups (BASELINE.json configs[4])
// every tick carries hundreds of such ranges, so they are processed n at a time:
//
//   k_ranges_validate   argument ranges; one round per leader group within the launch (the run contract of the
//                       per-acceptor `round` scalar, as for Phase2a batches)
//   k_ranges_open       the proxy leader's states map for ranges: an open-addressing hash table keyed by
//                       (slotStart, slotEnd) + round, claimed with one 64-bit CAS; a key seen twice in one launch
//                       is the reference's "already received this Phase2aNoopRange: ignoring" for every copy but
//                       the one with the lowest index
//   k_ranges_resolve    who opened what (is_new), the per-slot shadow of a length-1 range (its key collides with
//                       the single-slot tally's, mencius/ProxyLeader.scala:86-90)
//   k_ranges_acceptors  one thread per (range, acceptor group, acceptor): Nack, or round := round and a vote
//   k_ranges_fill       (round, Noop) into every owned slot of the range for the voting acceptors
//   k_ranges_tally      Phase2bNoopRange votes: a quorum f+1 from EVERY acceptor group -> ChosenNoopRange, Done
//   k_ranges_rehash     garbage collection (fpx_proxy_forget): live entries outside the forgotten window move to
//                       the other table buffer; nothing is ever deleted in place, so probing needs no tombstones
//
// HBM-write-bound where it moves data at all (8 B per voted cell); the rest is a few bytes per range.
#pragma once

#include "fpx_kernels.hpp"

namespace fpx {

constexpr uint32_t RT_PENDING = 1, RT_DONE = 2;

struct RangeTable {
  uint64_t* key;    // [cap][2]  k0 = (start + 1) << 32 | end (0 = empty); k1 = stamp << 32 | round << 2 | state
  uint64_t* bits;   // [cap][A * 4]  Phase2bNoopRange votes per acceptor group (bit = acceptor index)
  int32_t* owner;   // [cap]  lowest index of the message that inserted the entry in launch `stamp`
  int32_t* count;   // [1]    entries in use
  int32_t cap;      // power of two
};

struct RangeBatch {
  int32_t n;
  const int32_t* start;
  const int32_t* end;
  const int32_t* round;
  const uint64_t* target;  // [n][A][4] or null
  uint64_t* vote_bits;     // [n][A][4]
  uint64_t* nack_bits;     // [n][A][4]
  int32_t* nack_round;     // [n]
  int32_t* entry;          // [n] scratch: table entry of the range, -1 = ignored / invalid
  uint8_t* is_new;         // [n]
  uint8_t* chosen;         // [n]
  const uint64_t* votes_in;  // tally input [n][A][4] (the unfused Phase2bNoopRange entry point)
  uint32_t run_id;
  int32_t fused;           // acceptors / fill / tally act only on ranges this launch opened
  int32_t quorum;          // f + 1
};

__device__ __forceinline__ uint64_t range_k0(int start, int end) {
  return ((uint64_t)(uint32_t)(start + 1) << 32) | (uint64_t)(uint32_t)end;
}
__device__ __forceinline__ uint32_t range_hash(uint64_t k0) { return (uint32_t)(mix64(k0) >> 20); }

__global__ void __launch_bounds__(256) k_ranges_validate(const Geom g, const State st, const RangeBatch b, int check_round) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.n) return;
  const int s = b.start[i], e = b.end[i], r = b.round[i];
  if (s < 0 || e < s || e > g.S || r < 0 || r > MAX_ROUND) {
    report_abort(st, 1 /*FPX_EINVAL*/, i, s, r);
    return;
  }
  if (check_round) {  // one round per leader group within the launch
    int* rr = &st.run_round[(s % g.num_leader_groups) * g.num_groups];
    int cur = *reinterpret_cast<volatile int*>(rr);  // ordinary load first (see k_validate)
    if (cur == -1) {
      cur = atomicCAS(rr, -1, r);
      if (cur == -1) cur = r;
    }
    if (cur != r) report_abort(st, 6 /*FPX_EORDER*/, i, s, r);
  }
}

// One step of a Mencius proxy leader = the commands of the leader groups that have some + the noop ranges of those that skip
// (fpx_mencius_band_fused_dev).  The two halves touch disjoint rows, tallies and acceptor scalars iff no leader group
// (slot % numLeaderGroups, mencius/ProxyLeader.scala:231-234) has both a command and a range in the step: then they may
// run side by side.  The caller says so; unless the context is FPX_F_TRUSTED these two kernels hold it to its word before
// anything is applied (FPX_EORDER, nothing applied).
__global__ void __launch_bounds__(256) k_band_mark(const Geom g, int n, const int32_t* start, int32_t* lg_mark) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = start[i];
  if (s >= 0) lg_mark[s % g.num_leader_groups] = 1;
}
__global__ void __launch_bounds__(256) k_band_check(const Geom g, const State st, int n, const int32_t* slot, const int32_t* round,
                                                    const int32_t* lg_mark, int index_base) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = slot[i];
  // (the reported index counts from the caller's first message, as k_validate's does: ADVICE r05)
  if (s >= 0 && lg_mark[s % g.num_leader_groups] != 0) report_abort(st, 6 /*FPX_EORDER*/, i + index_base, s, round ? round[i] : 0);
}

// mencius/ProxyLeader.scala:255-303.  lookup = 1: find only (the Phase2bNoopRange entry point).
// Returns the table entry (-1 unknown / refused, -2 swallowed).  *inserted: this call created the entry; *shared: the
// entry was created by ANOTHER message of this launch (the same key twice in one batch).  count_hint: a value of
// rt.count read at launch start, or -1 -- with n messages in the launch the table cannot reach its limit while
// count_hint + n stays below it, and the live read of the counter (a dependent round trip to L2) is skipped.
__device__ __forceinline__ int ranges_open_core(const Geom& g, const State& st, const RangeTable& rt, const RangeBatch& b, int lookup, int i,
                                                int s, int e, int rnd, int count_hint, bool* inserted, bool* shared) {
  *inserted = false, *shared = false;
  const uint32_t want = (uint32_t)rnd + 1u;
  if (e == s + 1) {
    // the key (slot, slot + 1, round) is also the key of the single-slot tally: if that one exists -- Pending or
    // Done -- the range message is swallowed (:259-266 on open, :370-385 on Phase2bNoopRange)
    const uint32_t* kr = st.pl_key + (size_t)phys_slot(g, s) * g.wp;
    for (int w = 0; w < g.ways; ++w)
      if ((kr[w] & KEY_ROUND_MASK) == want && !(kr[w] & KEY_RANGE)) {
        return -2;  // swallowed
      }
  }
  const uint64_t k0 = range_k0(s, e);
  const uint32_t mask = (uint32_t)rt.cap - 1u;
  uint32_t p = range_hash(k0) & mask;
  for (int probes = 0; probes < rt.cap; ++probes, p = (p + 1) & mask) {
    uint64_t cur = __hip_atomic_load(&rt.key[(size_t)p * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == 0) {
      if (lookup) return -1;  // unknown key
      if ((count_hint < 0 || count_hint + b.n >= rt.cap / 2) &&
          __hip_atomic_load(rt.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= rt.cap / 2) {
        report(st, 5 /*FPX_ECAPACITY*/, i, s, rnd);
        return -1;
      }
      const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&rt.key[(size_t)p * 2]), 0ull, (unsigned long long)k0);
      if (prev == 0) {  // mine: :295-301 PendingPhase2aNoopRange(phase2a, no votes) -- bits and owner were initialised
                        // when the table buffer was (entries are never reused in place)
        rt.key[(size_t)p * 2 + 1] = ((uint64_t)b.run_id << 32) | ((uint64_t)(uint32_t)rnd << 2) | RT_PENDING;
        atomicAdd(rt.count, 1);
        atomicMin(&rt.owner[p], i);
        *inserted = true;
        return (int)p;
      }
      cur = prev;  // somebody else claimed the slot between the load and the CAS: look at what is there now
    }
    if (cur == k0) {
      const uint64_t k1 = __hip_atomic_load(&rt.key[(size_t)p * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // k1 == 0: being inserted by another message of THIS launch -- same (start, end), hence (run contract: one
      // round per leader group) the same round: the same key.  Its stamp says the same once it is written.
      const bool this_launch = k1 == 0 || (uint32_t)(k1 >> 32) == b.run_id;
      if (this_launch || (uint32_t)((k1 >> 2) & 0x3fffffffu) == (uint32_t)rnd) {
        if (this_launch && !lookup) atomicMin(&rt.owner[p], i);
        *shared = this_launch;
        return (int)p;
      }
      // the same range in another round: a different key, keep probing
    }
  }
  if (!lookup) report(st, 5, i, s, rnd);
  return -1;
}

__device__ __forceinline__ void ranges_open_one(const Geom& g, const State& st, const RangeTable& rt, const RangeBatch& b, int lookup, int i) {
  bool inserted, shared;
  b.entry[i] = ranges_open_core(g, st, rt, b, lookup, i, b.start[i], b.end[i], b.round[i], -1, &inserted, &shared);
}

__global__ void __launch_bounds__(256) k_ranges_open(const Geom g, const State st, const RangeTable rt, const RangeBatch b, int lookup) {
  if (st.status[ST_ABORT] != 0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < b.n) ranges_open_one(g, st, rt, b, lookup, i);
}

// after k_ranges_open: is_new[i] <=> this launch inserted the entry and i is its lowest index; the owner of a
// new length-1 range also claims the per-slot shadow way
__device__ __forceinline__ void ranges_resolve_one(const Geom& g, const State& st, const RangeTable& rt, const RangeBatch& b, int i) {
  const int e = b.entry[i];
  bool fresh = false;
  if (e >= 0) {
    const uint64_t k1 = rt.key[(size_t)e * 2 + 1];
    fresh = (uint32_t)(k1 >> 32) == b.run_id && rt.owner[e] == i;
  }
  if (fresh && b.end[i] == b.start[i] + 1) {
    uint32_t* kr = st.pl_key + (size_t)phys_slot(g, b.start[i]) * g.wp;
    int way = -1;
    for (int w = g.ways - 1; w >= 0; --w)
      if (kr[w] == 0) way = w;
    if (way < 0) report(st, 5, i, b.start[i], b.round[i]);
    else kr[way] = ((uint32_t)b.round[i] + 1u) | KEY_RANGE;
  }
  if (b.is_new) b.is_new[i] = fresh ? 1 : 0;
  if (!fresh && b.fused) b.entry[i] = e >= 0 ? -3 - e : e;  // not mine to drive: acceptors / fill / tally skip it
}

__global__ void __launch_bounds__(256) k_ranges_resolve(const Geom g, const State st, const RangeTable rt, const RangeBatch b) {
  if (st.status[ST_ABORT] != 0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < b.n) ranges_resolve_one(g, st, rt, b, i);
}

// mencius/Acceptor.scala:237-260, 279-290: one thread per (range, acceptor group, acceptor)
__device__ __forceinline__ void ranges_acceptor_one(const Geom& g, const State& st, const RangeBatch& b, long long idx) {
  const int A = g.num_groups, L = g.num_leader_groups;
  const long long per = (long long)A * g.R;
  const int i = (int)(idx / per), rem = (int)(idx % per);
  if (b.fused && b.entry[i] < 0) return;
  const int ag = rem / g.R, r = rem % g.R, bit = g.base + r;
  const size_t row = ((size_t)i * A + ag) * 4;
  if (b.target && !((b.target[row + (bit >> 6)] >> (bit & 63)) & 1ull)) return;
  const int start = b.start[i], end = b.end[i], round = b.round[i];
  const int lg = start % L;  // slotSystem.leader(slotStartInclusive)
  const size_t acc = (size_t)(lg * A + ag) * g.R + r;
  const int pr = st.promised[acc];
  if (round < pr) {  // :245-256 Nack(round = my round)
    if (b.nack_bits) atomicOr((unsigned long long*)&b.nack_bits[row + (bit >> 6)], 1ull << (bit & 63));
    if (b.nack_round) atomicMax(&b.nack_round[i], pr);
    return;
  }
  if (pr != round) st.promised[acc] = round;  // :260 (every message of this leader group carries this round)
  atomicOr((unsigned long long*)&b.vote_bits[row + (bit >> 6)], 1ull << (bit & 63));
  // the largest slot of the range owned by my acceptor group: maxVotedSlot, a multipaxos scalar (Acceptor.scala:104) the
  // library keeps in every mode for one readback format -- mencius/Acceptor.scala has none and no message carries it
  const int rows = (end - start + L - 1) / L;
  for (int j = rows - 1; j >= 0 && j >= rows - A; --j) {
    const int s = start + j * L;
    if ((s / L) % A == ag) {
      if (s > st.max_voted[acc]) atomicMax(&st.max_voted[acc], s);
      break;
    }
  }
}

__global__ void __launch_bounds__(256) k_ranges_acceptors(const Geom g, const State st, const RangeBatch b) {
  if (st.status[ST_ABORT] != 0) return;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (long long)g.num_groups * g.R * b.n) ranges_acceptor_one(g, st, b, idx);
}

// mencius/Acceptor.scala:262-277: blockIdx.y strides over the ranges, x over the cells of one range
__global__ void __launch_bounds__(256) k_ranges_fill(const Geom g, const State st, const RangeBatch b) {
  if (st.status[ST_ABORT] != 0) return;
  const int A = g.num_groups, L = g.num_leader_groups;
  for (int i = blockIdx.y; i < b.n; i += gridDim.y) {
    if (b.fused && b.entry[i] < 0) continue;
    const int start = b.start[i], end = b.end[i], round = b.round[i];
    const long long rows = ((long long)end - start + L - 1) / L;
    const long long total = rows * g.R;
    const uint64_t* votes = b.vote_bits + (size_t)i * A * 4;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long long)gridDim.x * blockDim.x) {
      const int s = start + (int)(c / g.R) * L;
      const int r = (int)(c % g.R);
      const int ag = (s / L) % A, bit = g.base + r;
      if ((votes[(size_t)ag * 4 + (bit >> 6)] >> (bit & 63)) & 1ull) {
        const size_t cell = (size_t)phys_slot(g, s) * g.VS + r;
        st.vote_round[cell] = round;  // :271-276 State(voteRound = round, voteValue = Noop)
        st.vote_value[cell] = -1;
        if (st.row_voted[phys_slot(g, s)] == 0) st.row_voted[phys_slot(g, s)] = 1;
      }
    }
  }
}

// The same, walking the LOG instead of the ranges.  One range writes a 16-byte row (R = 3) every L x 16 B = 4 KB: a
// different DRAM page per store, 3 four-byte stores per row (k_ranges_fill above: 0.8 TB/s of written bytes on
// BASELINE.json configs[4]).  Here a workgroup takes a span of RF_JB consecutive rows of L slots, marks in LDS which
// range (if any) covers each slot, and sweeps the span in memory order: the rows of neighbouring leader groups that
// skip are neighbours in memory, every row of fully voting acceptors leaves as one aligned 16-byte store per array.
// Measured on configs[4] (every other leader group skips: a 16-byte row every 32 bytes): 67 us, the same as the
// range-major kernel -- and the same again with the two vote arrays interleaved into one 32-byte sector per slot
// (FPX_INTERLEAVE=1), which halves the bytes written (138 -> 71 MB).  So neither DRAM page locality nor bytes bound
// it; what is left is one request per 16 useful bytes (profiles/r03_cfg5.md).
// Two ranges of the launch that cover the same slot (a leader group that sends overlapping ranges in one tick) send
// the span through a per-slot loop over all ranges instead.  Chosen by the host for launches of at most RF_MAXN ranges.
constexpr int RF_JB = 8, RF_MAXN = 1024, RF_MAXL = 2048;
__global__ void __launch_bounds__(256) k_ranges_fill_rows(const Geom g, const State st, const RangeBatch b) {
  extern __shared__ uint32_t rf_own[];  // [RF_JB][L] index + 1 of the range that covers the slot, 0 = none
  __shared__ int span_lo, span_hi, overlap;
  if (st.status[ST_ABORT] != 0) return;
  const int A = g.num_groups, L = g.num_leader_groups, Q = g.RS >> 2;
  if (threadIdx.x == 0) span_lo = 0x7fffffff, span_hi = -1;
  __syncthreads();
  {
    int lo = 0x7fffffff, hi = -1;
    for (int i = threadIdx.x; i < b.n; i += 256) {
      if (b.fused && b.entry[i] < 0) continue;
      const int s0 = b.start[i], e0 = b.end[i];
      if (e0 <= s0) continue;
      lo = min(lo, s0 / L), hi = max(hi, (e0 - 1 - s0 % L) / L);  // rows of its first and last slot
    }
    if (hi >= 0) atomicMin(&span_lo, lo), atomicMax(&span_hi, hi);
  }
  __syncthreads();
  const int r_lo = span_lo, r_hi = span_hi;
  if (r_hi < r_lo) return;
  const int nspans = (r_hi - r_lo) / RF_JB + 1;
  for (int q = blockIdx.x; q < nspans; q += gridDim.x) {
    const int j0 = r_lo + q * RF_JB;
    for (int t = threadIdx.x; t < RF_JB * L; t += 256) rf_own[t] = 0;
    if (threadIdx.x == 0) overlap = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < b.n; i += 256) {
      if (b.fused && b.entry[i] < 0) continue;
      const int s0 = b.start[i], e0 = b.end[i];
      if (e0 <= s0) continue;
      const int lg = s0 % L, ja = max(s0 / L, j0), jb = min((e0 - 1 - lg) / L, j0 + RF_JB - 1);
      for (int j = ja; j <= jb; ++j) {
        const uint32_t old = atomicCAS(&rf_own[(j - j0) * L + lg], 0u, (uint32_t)i + 1u);
        if (old != 0 && old != (uint32_t)i + 1u) overlap = 1;
      }
    }
    __syncthreads();
    const bool slow = overlap != 0;
    // one row of L x Q quads at a time (no division by run-time values per cell when Q == 1)
    for (int t = threadIdx.x; t < RF_JB * L * Q; t += 256) {
      const int jrel = t / (L * Q), u = t - jrel * (L * Q);
      const int lg = Q == 1 ? u : u / Q, quad = Q == 1 ? 0 : u - lg * Q;
      const uint32_t o = rf_own[jrel * L + lg];
      if (o == 0) continue;
      const int row = j0 + jrel, s = row * L + lg, ag = row % A, r0 = quad * 4;
      unsigned voted = 0, valid = 0;
      int round = 0;
      if (!slow) {
        const int i = (int)o - 1;
        const uint64_t* votes = b.vote_bits + ((size_t)i * A + ag) * 4;
        round = b.round[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int bit = g.base + r0 + c;
          if (r0 + c < g.R) valid |= 1u << c, voted |= (unsigned)((votes[bit >> 6] >> (bit & 63)) & 1ull) << c;
        }
      } else {  // every range that covers the slot (one round per leader group and launch: the votes are a union)
        for (int i = 0; i < b.n; ++i) {
          if (b.fused && b.entry[i] < 0) continue;
          const int s0 = b.start[i], e0 = b.end[i];
          if (s < s0 || s >= e0 || s0 % L != lg) continue;
          const uint64_t* votes = b.vote_bits + ((size_t)i * A + ag) * 4;
          round = b.round[i];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int bit = g.base + r0 + c;
            if (r0 + c < g.R) valid |= 1u << c, voted |= (unsigned)((votes[bit >> 6] >> (bit & 63)) & 1ull) << c;
          }
        }
      }
      if (voted == 0) continue;
      const size_t cell = (size_t)phys_slot(g, s) * g.VS + r0;
      if (voted == valid) {  // :271-276 State(voteRound = round, voteValue = Noop); padding cells stay -1
        int4 vr = make_int4(round, round, round, round);
        if (!(valid & 2u)) vr.y = -1;
        if (!(valid & 4u)) vr.z = -1;
        if (!(valid & 8u)) vr.w = -1;
        *reinterpret_cast<int4*>(st.vote_round + cell) = vr;
        *reinterpret_cast<int4*>(st.vote_value + cell) = make_int4(-1, -1, -1, -1);
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if ((voted >> c) & 1u) st.vote_round[cell + c] = round, st.vote_value[cell + c] = -1;
      }
      st.row_voted[phys_slot(g, s)] = 1;
    }
    __syncthreads();
  }
}

// The same with leader-group-major rows (Geom::lg_rows): the slots of a range are rows ja..jb of ONE leader group and
// neighbours in memory, so the range is two runs of 16-byte stores (voteRound = round, voteValue = Noop) and a run of
// row_voted bytes -- every 128-byte line leaves the wavefront whole.  blockIdx.y strides over the ranges, x over the
// range's rows.  A range that shares rows with another range of its leader group in the same launch, acceptor groups
// that differ in who voted (A > 1), or interleaved vote arrays go row by row through the union of the covering ranges.
// (the body by the workgroup's row `by` of `gy`: k_ranges_fill_lg is its own grid; k_ranges_fill_lg_fin gives further
// rows of the grid to the vote kernel's k_finalize)
__device__ __forceinline__ void ranges_fill_lg_body(const Geom& g, const State& st, const RangeBatch& b, int by, int gy) {
  __shared__ int overlap;
  if (st.status[ST_ABORT] != 0) return;
  const int A = g.num_groups, L = g.num_leader_groups, Q = g.RS >> 2;
  const int tid = threadIdx.x;
  for (int i = by; i < b.n; i += gy) {
    if (b.fused && b.entry[i] < 0) continue;
    const int s0 = b.start[i], e0 = b.end[i];
    if (e0 <= s0) continue;
    const int lg = s0 % L, ja = s0 / L, jb = (e0 - 1 - lg) / L;
    __syncthreads();
    if (tid == 0) overlap = 0;
    __syncthreads();
    for (int k = tid; k < b.n; k += 256) {
      if (k == i || (b.fused && b.entry[k] < 0)) continue;
      const int s1 = b.start[k], e1 = b.end[k];
      if (e1 <= s1 || s1 % L != lg) continue;
      if (s1 / L <= jb && (e1 - 1 - lg) / L >= ja) overlap = 1;
    }
    __syncthreads();
    const size_t p0 = (size_t)lg * g.lg_rows;
    const long long nthreads = (long long)gridDim.x * 256, me = (long long)blockIdx.x * 256 + tid;
    if (overlap == 0 && A == 1 && g.VS == g.RS) {
      const int round = b.round[i];
      const uint64_t* votes = b.vote_bits + (size_t)i * 4;
      uint64_t vb[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) vb[w] = votes[w];
      // the int4 units of rows ja..jb: unit e is quad e % Q of row e / Q
      const long long e_lo = (long long)(p0 + ja) * Q, e_hi = (long long)(p0 + jb + 1) * Q;
      bool any = false;
      for (int q4 = 0; q4 < Q; ++q4) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int bit = g.base + q4 * 4 + c;
          if (q4 * 4 + c < g.R) any |= (vb[bit >> 6] >> (bit & 63)) & 1ull;
        }
      }
      if (!any) continue;
      for (long long e = e_lo + me; e < e_hi; e += nthreads) {
        const int quad = Q == 1 ? 0 : (int)(e % Q), r0 = quad * 4;
        unsigned voted = 0, valid = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int bit = g.base + r0 + c;
          if (r0 + c < g.R) valid |= 1u << c, voted |= (unsigned)((vb[bit >> 6] >> (bit & 63)) & 1ull) << c;
        }
        if (voted == 0) continue;
        const size_t cell = (size_t)e * 4;
        if (voted == valid) {  // :271-276 State(voteRound = round, voteValue = Noop); padding cells stay -1
          int4 vr = make_int4(round, round, round, round);
          if (!(valid & 2u)) vr.y = -1;
          if (!(valid & 4u)) vr.z = -1;
          if (!(valid & 8u)) vr.w = -1;
          *reinterpret_cast<int4*>(st.vote_round + cell) = vr;
          *reinterpret_cast<int4*>(st.vote_value + cell) = make_int4(-1, -1, -1, -1);
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if ((voted >> c) & 1u) st.vote_round[cell + c] = round, st.vote_value[cell + c] = -1;
        }
      }
      // row_voted[pa..pb] = 1, four rows to a store where the range covers the aligned four
      const long long pa = (long long)(p0 + ja), pb = (long long)(p0 + jb);
      for (long long q4 = (pa >> 2) + me; q4 <= (pb >> 2); q4 += nthreads) {
        const long long r4 = q4 << 2;
        if (r4 >= pa && r4 + 3 <= pb) {
          *reinterpret_cast<uint32_t*>(st.row_voted + r4) = 0x01010101u;
        } else {
          for (int c = 0; c < 4; ++c)
            if (r4 + c >= pa && r4 + c <= pb) st.row_voted[r4 + c] = 1;
        }
      }
      continue;
    }
    // row by row: every range of the launch that covers the slot (one round per leader group and launch)
    for (long long t = me; t < (long long)(jb - ja + 1) * Q; t += nthreads) {
      const int row = ja + (int)(t / Q), quad = (int)(t % Q), r0 = quad * 4, s = row * L + lg, ag = row % A;
      unsigned voted = 0;
      int round = 0;
      for (int k = 0; k < b.n; ++k) {
        if (k != i && overlap == 0) continue;
        if (b.fused && b.entry[k] < 0) continue;
        const int s1 = b.start[k], e1 = b.end[k];
        if (s < s1 || s >= e1 || s1 % L != lg) continue;
        const uint64_t* votes = b.vote_bits + ((size_t)k * A + ag) * 4;
        round = b.round[k];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int bit = g.base + r0 + c;
          if (r0 + c < g.R) voted |= (unsigned)((votes[bit >> 6] >> (bit & 63)) & 1ull) << c;
        }
      }
      if (voted == 0) continue;
      const size_t ps = p0 + row, cell = ps * g.VS + r0;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if ((voted >> c) & 1u) st.vote_round[cell + c] = round, st.vote_value[cell + c] = -1;
      st.row_voted[ps] = 1;
    }
  }
}

__global__ void __launch_bounds__(256) k_ranges_fill_lg(const Geom g, const State st, const RangeBatch b) {
  ranges_fill_lg_body(g, st, b, (int)blockIdx.y, (int)gridDim.y);
}
// A Mencius band whose halves are independent (fpx_mencius_band_fused_dev): the first fin_rows rows of the grid are the
// vote kernel's k_finalize (fgx x slices workgroups, laid row by row over the grid's width), the other gy rows the fill --
// what the commands raised in promised / max_voted is of no concern to the ranges of OTHER leader groups, so the two
// need no order, and one launch (and its gap) less is on the step's critical path
__global__ void __launch_bounds__(256) k_ranges_fill_lg_fin(const Geom g, const State st, const RangeBatch b, int gy, int fin_rows, int par,
                                                            int grid, uint32_t seq, int fgx, int slices) {
  if ((int)blockIdx.y >= fin_rows) {
    ranges_fill_lg_body(g, st, b, (int)blockIdx.y - fin_rows, gy);
    return;
  }
  const int f = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
  if (f < fgx * slices) finalize_body(g, st, par, grid, seq, f % fgx, f / fgx, slices);
}

// mencius/ProxyLeader.scala:355-411: one thread per range
__device__ __forceinline__ void ranges_tally_one(const Geom& g, const State& st, const RangeTable& rt, const RangeBatch& b, int i) {
  uint8_t ch = 0;
  const int e = b.entry[i];
  const int A = g.num_groups;
  const uint64_t* in = (b.votes_in ? b.votes_in : b.vote_bits) + (size_t)i * A * 4;
  if (e == -1 && !b.fused) {
    // never opened: fatal (:361-368) -- unless the message carries no vote at all ("no message")
    bool any = false;
    for (int w = 0; w < A * 4; ++w) any = any || in[w] != 0;
    if (any) report(st, 2 /*FPX_EFATAL_UNKNOWN_SLOTROUND*/, i, b.start[i], b.round[i]);
  } else if (e >= 0) {
    uint64_t* k1p = &rt.key[(size_t)e * 2 + 1];
    if ((uint32_t)(*k1p & 3u) == RT_PENDING) {  // Done: ignored (:370-376)
      uint64_t* bits = rt.bits + (size_t)e * A * 4;
      bool all = true;
      for (int ag = 0; ag < A; ++ag) {
        int c = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const uint64_t x = bits[ag * 4 + w] | (in[ag * 4 + w] & g.member[w]);  // :389-390
          bits[ag * 4 + w] = x;
          c += __popcll(x);
        }
        all = all && c >= b.quorum;  // :391
      }
      if (all) {
        *k1p = (*k1p & ~3ull) | RT_DONE;  // :410 ; ChosenNoopRange(start, end) :395-407
        ch = 1;
      }
    }
  }
  if (b.chosen) b.chosen[i] = ch;
}

__global__ void __launch_bounds__(256) k_ranges_tally(const Geom g, const State st, const RangeTable rt, const RangeBatch b) {
  if (st.status[ST_ABORT] != 0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < b.n) ranges_tally_one(g, st, rt, b, i);
}

// A fused launch of a few ranges (what a tick of a Mencius deployment carries: one per lagging leader group) is a
// chain of dependent steps over a few hundred items -- as five launches, ~5 us each plus the gaps between them, several
// times what the steps compute.  ONE workgroup walks the chain instead, and everything the steps say to each other
// stays in LDS: the ranges' fields, their table entries, the vote and Nack bitmaps.  Global memory sees what must
// persist -- the table insert (one load + one CAS per range), promised / maxVotedSlot, the tally's votes -- and the
// outputs, written once at the end; k_ranges_fill* follows as its own launch and reads entry[] and vote_bits from
// there.  What the round-3 chain paid for and this one does not: a device-scope fence + L1 invalidate after every
// barrier (the steps talked through global memory), the resolve step's reads of k1 / owner (an insert is the owner's
// unless the same key came twice in the launch, which the open step notices and flags in LDS), the tally's reads of
// an entry it has just created (no votes, Pending: known), and the live read of the table's fill counter.
// 16 -> 6 us for the 256 ranges of BASELINE.json configs[4] (profiles/r04_cfg5.md).
constexpr int RANGES_CHAIN_MAX = 2048, RANGES_CHAIN_LDS_WORDS = 36000;  // 144 KB of the CU's 160
__host__ __device__ inline long long ranges_chain_words(int n, int A) { return 5ll * n + 16ll * n * A; }
// (NT threads: k_ranges_chain is one workgroup of 1024; the first workgroup of k_phase2_band walks the chain with 256)
template <int NT>
__device__ __forceinline__ void ranges_chain_body(const Geom& g, const State& st, const RangeTable& rt, const RangeBatch& b, uint32_t* ch_lds) {
  __shared__ int shared_keys;
  if (st.status[ST_ABORT] != 0) return;
  const int tid = threadIdx.x, n = b.n, A = g.num_groups, R = g.R, L = g.num_leader_groups;
  int32_t* c_start = reinterpret_cast<int32_t*>(ch_lds);
  int32_t* c_end = c_start + n;
  int32_t* c_round = c_end + n;
  int32_t* c_entry = c_round + n;
  int32_t* c_nr = c_entry + n;
  uint32_t* c_votes = reinterpret_cast<uint32_t*>(c_nr + n);  // [n][A][8]
  uint32_t* c_nacks = c_votes + (size_t)n * A * 8;
  const int count0 = __hip_atomic_load(rt.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int i = tid; i < n; i += NT) c_start[i] = b.start[i], c_end[i] = b.end[i], c_round[i] = b.round[i], c_nr[i] = -1;
  for (int t = tid; t < n * A * 16; t += NT) c_votes[t] = 0;
  if (tid == 0) shared_keys = 0;
  __syncthreads();
  // the acceptor this thread plays in the vote step: its round is requested now, the open step runs while it travels
  // (two ranges of one leader group carry the same round -- the run contract -- so whichever of them writes promised
  // first, a value read before that write leads to the same vote)
  const int per = A * R, cells = n * per;
  int pr0 = 0;
  if (tid < cells) pr0 = st.promised[(size_t)((c_start[tid / per] % L) * A + (tid % per) / R) * R + (tid % per) % R];
  // open (mencius/ProxyLeader.scala:255-303)
  uint32_t mine = 0;  // bit k: the range of my k-th pass was inserted by this launch (RANGES_CHAIN_MAX / NT <= 8 passes)
  for (int i = tid, k = 0; i < n; i += NT, ++k) {
    bool inserted, shared;
    c_entry[i] = ranges_open_core(g, st, rt, b, 0, i, c_start[i], c_end[i], c_round[i], count0, &inserted, &shared);
    mine |= inserted ? 1u << k : 0u;
    if (shared) shared_keys = 1;
  }
  __syncthreads();
  // resolve: is_new <=> this launch created the entry and i is the lowest index that carries its key
  const bool twice = shared_keys != 0;
  for (int i = tid, k = 0; i < n; i += NT, ++k) {
    const int e = c_entry[i];
    bool fresh = (mine >> k) & 1u;
    if (twice && e >= 0) {
      const uint64_t k1 = __hip_atomic_load(&rt.key[(size_t)e * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      fresh = (uint32_t)(k1 >> 32) == b.run_id && __hip_atomic_load(&rt.owner[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == i;
    }
    if (fresh && c_end[i] == c_start[i] + 1) {  // the owner of a new length-1 range also claims the per-slot shadow way
      uint32_t* kr = st.pl_key + (size_t)phys_slot(g, c_start[i]) * g.wp;
      int way = -1;
      for (int w = g.ways - 1; w >= 0; --w)
        if (kr[w] == 0) way = w;
      if (way < 0) report(st, 5, i, c_start[i], c_round[i]);
      else kr[way] = ((uint32_t)c_round[i] + 1u) | KEY_RANGE;
    }
    if (b.is_new) b.is_new[i] = fresh ? 1 : 0;
    const int out = fresh ? e : (e >= 0 ? -3 - e : e);  // not mine to drive: acceptors / fill / tally skip it
    c_entry[i] = out, b.entry[i] = out;
  }
  __syncthreads();
  // acceptors (mencius/Acceptor.scala:237-260, 279-290): one thread per (range, acceptor group, acceptor)
  for (int idx = tid; idx < cells; idx += NT) {
    const int i = idx / per, rem = idx % per;
    if (c_entry[i] < 0) continue;
    const int ag = rem / R, r = rem % R, bit = g.base + r;
    const size_t row = ((size_t)i * A + ag) * 4;
    if (b.target && !((b.target[row + (bit >> 6)] >> (bit & 63)) & 1ull)) continue;
    const int start = c_start[i], end = c_end[i], round = c_round[i];
    const size_t acc = (size_t)((start % L) * A + ag) * R + r;
    const int pr = idx == tid ? pr0 : st.promised[acc];
    const int word = (int)row * 2 + (bit >> 5);
    if (round < pr) {  // :245-256 Nack(round = my round)
      atomicOr(&c_nacks[word], 1u << (bit & 31));
      atomicMax(&c_nr[i], pr);
      continue;
    }
    if (pr != round) st.promised[acc] = round;  // :260
    atomicOr(&c_votes[word], 1u << (bit & 31));
    const int rows = (end - start + L - 1) / L;  // maxVotedSlot: see ranges_acceptor_one
    for (int j = rows - 1; j >= 0 && j >= rows - A; --j) {
      const int s = start + j * L;
      if ((s / L) % A == ag) {
        atomicMax(&st.max_voted[acc], s);
        break;
      }
    }
  }
  __syncthreads();
  // tally (mencius/ProxyLeader.scala:355-411) of the entries this launch created: no earlier votes, Pending
  for (int i = tid; i < n; i += NT) {
    const int e = c_entry[i];
    uint8_t ch = 0;
    if (e >= 0) {
      uint64_t* bits = rt.bits + (size_t)e * A * 4;
      bool all = true;
      for (int ag = 0; ag < A; ++ag) {
        int c = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const uint32_t* v = &c_votes[((size_t)i * A + ag) * 8 + w * 2];
          const uint64_t x = (((uint64_t)v[1] << 32) | v[0]) & g.member[w];  // :389-390
          bits[ag * 4 + w] = x;
          c += __popcll(x);
        }
        all = all && c >= b.quorum;  // :391
      }
      if (all) {  // :410 ; ChosenNoopRange(start, end) :395-407
        rt.key[(size_t)e * 2 + 1] = ((uint64_t)b.run_id << 32) | ((uint64_t)(uint32_t)c_round[i] << 2) | RT_DONE;
        ch = 1;
      }
    }
    if (b.chosen) b.chosen[i] = ch;
    if (b.nack_round) b.nack_round[i] = c_nr[i];
  }
  uint32_t* vo = reinterpret_cast<uint32_t*>(b.vote_bits);
  uint32_t* no = reinterpret_cast<uint32_t*>(b.nack_bits);
  for (int t = tid; t < n * A * 8; t += NT) {
    vo[t] = c_votes[t];
    if (no) no[t] = c_nacks[t];
  }
}
__global__ void __launch_bounds__(1024) k_ranges_chain(const Geom g, const State st, const RangeTable rt, const RangeBatch b) {
  extern __shared__ uint32_t ch_lds[];
  ranges_chain_body<1024>(g, st, rt, b, ch_lds);
}

// k_phase2 whose FIRST workgroup walks the range chain of the same Mencius band (fpx_mencius_band_fused_dev, independent
// halves): the chain -- one workgroup, a string of dependent steps -- hides under the vote kernel instead of standing in
// front of the fill.  (The first, not the last: a grid that fills the chip's LDS would start its last workgroup when the
// first voters are done.)
#define FPX_P2_NAME k_phase2_band
#define FPX_P2_EXTRA_PARAMS , const RangeTable rt, const RangeBatch rb
#define FPX_P2_NBLK (gridDim.x - 1)
#define FPX_P2_BID (blockIdx.x - 1)
#define FPX_P2_PROLOGUE                                                        \
  if (blockIdx.x == 0) {                                                       \
    ranges_chain_body<256>(g, st, rt, rb, reinterpret_cast<uint32_t*>(smem)); \
    return;                                                                    \
  }
#include "fpx_phase2_body.inc"
#undef FPX_P2_NAME
#undef FPX_P2_EXTRA_PARAMS
#undef FPX_P2_NBLK
#undef FPX_P2_BID
#undef FPX_P2_PROLOGUE

// fpx_proxy_forget: live entries that do not lie inside [first, first + count) move to the other buffer
__global__ void __launch_bounds__(256) k_ranges_rehash(const RangeTable from, const RangeTable to, int words, int first, int count) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= from.cap) return;
  const uint64_t k0 = from.key[(size_t)p * 2];
  if (k0 == 0) return;
  const int start = (int)(uint32_t)(k0 >> 32) - 1, end = (int)(uint32_t)k0;
  if (start >= first && (long long)end <= (long long)first + count) return;  // forgotten
  const uint32_t mask = (uint32_t)to.cap - 1u;
  uint32_t q = range_hash(k0) & mask;
  for (int probes = 0; probes < to.cap; ++probes, q = (q + 1) & mask) {
    if (atomicCAS(reinterpret_cast<unsigned long long*>(&to.key[(size_t)q * 2]), 0ull, (unsigned long long)k0) == 0) {
      to.key[(size_t)q * 2 + 1] = from.key[(size_t)p * 2 + 1];
      to.owner[q] = from.owner[p];
      for (int w = 0; w < words; ++w) to.bits[(size_t)q * words + w] = from.bits[(size_t)p * words + w];
      atomicAdd(to.count, 1);
      return;
    }
  }
}

// readback of one range tally (parity): out[0] = state (0 unknown, 1 Pending, 2 Done), then A * 4 words of votes
__global__ void k_ranges_read(const Geom g, const RangeTable rt, int start, int end, int round, uint64_t* out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const int words = g.num_groups * 4;
  out[0] = 0;
  for (int w = 0; w < words; ++w) out[1 + w] = 0;
  const uint64_t k0 = range_k0(start, end);
  const uint32_t mask = (uint32_t)rt.cap - 1u;
  uint32_t p = range_hash(k0) & mask;
  for (int probes = 0; probes < rt.cap; ++probes, p = (p + 1) & mask) {
    const uint64_t cur = rt.key[(size_t)p * 2];
    if (cur == 0) return;
    const uint64_t k1 = rt.key[(size_t)p * 2 + 1];
    if (cur == k0 && (uint32_t)((k1 >> 2) & 0x3fffffffu) == (uint32_t)round) {
      out[0] = k1 & 3u;
      if ((k1 & 3u) == RT_PENDING)
        for (int w = 0; w < words; ++w) out[1 + w] = rt.bits[(size_t)p * words + w];
      return;
    }
  }
}

// digest of the range tallies (fpx_state_digest out[7]): order-independent sum over the live entries
__global__ void __launch_bounds__(256) k_digest_ranges(const RangeTable rt, int words, uint64_t* out) {
  uint64_t acc = 0;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < rt.cap; p += gridDim.x * blockDim.x) {
    const uint64_t k0 = rt.key[(size_t)p * 2];
    if (k0 == 0) continue;
    const uint64_t k1 = rt.key[(size_t)p * 2 + 1];
    const uint64_t done = (k1 & 3u) == RT_DONE ? 1ull : 0ull;
    // (start, end, round, Done?) then the votes of a Pending entry
    uint64_t t = mix64((k0 - (1ull << 32)) * 0x9E3779B97F4A7C15ull + (((k1 >> 2) & 0x3fffffffull) << 1) + done);
    if (!done)
      for (int w = 0; w < words; ++w) t = mix64(t ^ rt.bits[(size_t)p * words + w]);
    acc += t;
  }
  block_add_u64(acc, out);
}

}  // namespace fpx
