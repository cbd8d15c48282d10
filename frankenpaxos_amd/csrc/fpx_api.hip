// fpx_api.hip -- the C ABI of include/fpx.h on top of the gfx950 kernels in fpx_kernels.hpp.
//
// There is NO CPU path in this file: every entry point that computes anything launches a HIP kernel
// and fails with FPX_ENODEVICE / FPX_EHIP when no device is usable.
#include "../../include/fpx.h"

#include <dlfcn.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <set>
#include <vector>

#include "fpx_kernels.hpp"
#include "fpx_ranges.hpp"
#include "fpx_wire_dev.hpp"
#include "../../include/fpx_wire.h"

using namespace fpx;

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace

// RCCL is bound at run time (dlopen), never at link time: a process that already carries an RCCL (PyTorch
// bundles its own librccl.so) must keep using THAT copy -- two RCCLs in one address space do not share their
// communicator state -- and a single-GPU caller needs no RCCL at all.  The handful of types below mirror
// rccl.h (NCCL 2.x ABI: ncclUniqueId is 128 opaque bytes, ncclSum = 0, ncclUint64 = 5, ncclSuccess = 0).
struct RcclUniqueId { char internal[128]; };
typedef void* RcclComm;
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(RcclUniqueId*) = nullptr;
  int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
  int (*CommDestroy)(RcclComm) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;  // optional: several collectives as one launch (fpx_comm_allgather_chosen_dev)
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
static_assert(sizeof(RcclUniqueId) == FPX_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
enum { RCCL_SUM = 0, RCCL_MAX = 2, RCCL_UINT8 = 1, RCCL_INT32 = 2, RCCL_UINT64 = 5 };  // ncclRedOp_t / ncclDataType_t values

// One call in flight: staging buffers for its inputs, two events, a page-locked copy of the status words taken right after
// its fused step.
constexpr int HOST_DEPTH = 3;
struct HostSlot {
  DevBuf in[4];               // the call's inputs staged in HBM (validation and the vote kernel read them there)
  DevBuf sink[3];             // where the records go that the caller did not ask for (a null output array)
  hipEvent_t up = nullptr;    // the inputs are up (the copy engine's stream)
  hipEvent_t done = nullptr;  // validation + fused step + status snapshot are done: the records are in the caller's arrays
  int32_t* status = nullptr;  // page-locked, 8 words: the device's status right after the call (k_status_snap)
  int32_t* status_dev = nullptr;
  bool busy = false;          // submitted, not waited for yet
  int n = 0;
};

struct fpx_ctx {
  fpx_config cfg;
  Geom g;
  State st;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;
  int lanes_per_slot = 64;  // G
  bool vec = false;
  int num_cus = 256;
  int max_grid = 2048;
  uint32_t run_id = 0;
  int64_t bytes = 0;
  int last_hip = 0;
  int32_t err_index = -1, err_slot = -1, err_round = -1;
  // host-API staging (device)
  DevBuf d_slot, d_round, d_value, d_target, d_bits_a, d_bits_b, d_i32_a, d_i32_b, d_i32_c, d_u8, d_scratch;
  // host-side run splitting
  std::vector<uint32_t> hstamp;
  uint32_t hrun = 0;
  std::vector<int32_t> hround;
  // kernel timing (fpx_profile_*)
  void* slab = nullptr;  // vote_round | vote_value | ballot
  // chunk placement (place_chunks): the slab is a reserved address range backed by 1 GiB physical allocations
  std::vector<hipMemGenericAllocationHandle_t> vmm_chunks;  // the chunks mapped into the slab, in address order
  size_t vmm_reserved = 0;                                  // bytes of the reservation at `slab` (0: the slab is one hipMalloc)
  int placement_probes = 0, placement_unprobed = 0;          // the search itself: probes run, decisions taken unprobed (budget spent)
  float placement_ms = 0.f;                                 // wall clock of the search
  float placement[5] = {0, 0, 0, 0, 0};                     // mode (0 one allocation, 1 chunks), windows, min / median / max probe ms
  uint32_t phase2_launches = 0;
  int lz_min_from = 0x7fffffff;     // the smallest watermark of any Phase1a since the lazy records were last cleared: a Phase1a whose
                                    // watermark is not above it cannot meet an older record that starts below its own (k_p1a_fast)
  // the fold (k_finalize's work) of the last K3 launch, when it has not been launched: it rides in the next eligible vote
  // kernel (k_phase2_fin) or is launched by whatever touches the context next (DeviceGuard -> flush_pending_fin)
  FinJob pending_fin = {};       // nblk != 0: pending
  FinJob carry_fin = {};         // what the launch being enqueued may take into its grid (consumed by launch_phase2_3)
  State launch_st = {};          // the State of the launch being enqueued: st with ITS half of part / part_stamp
  long long fins_carried = 0;    // diagnostic
  uint32_t launch_seq = 0;  // stamps the partial-maxima rows of a K1 / K3 launch (never 0 in a row that counts)
  bool batch_increasing = false, batch_one_round = false;  // check_inputs' findings about the current host batch
  bool force_validate = false;  // host-pointer K3's optimistic whole-batch run is validated even under FPX_F_TRUSTED
  bool host_validated = false;  // set while a host entry point drives runs it cut itself (split_runs)
  bool profiling = false;
  std::vector<hipEvent_t> ev;  // start/stop pairs
  size_t ev_used = 0;
  // the events of the K1 / K3 launch being enqueued: they ride on the kernel's own dispatch packet (hipExtLaunchKernelGGL),
  // so a timed launch puts no marker packets on the stream (two hipEventRecords per launch cost 5 - 15 us of a step)
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  // host-pointer K3 on big batches: upload / K3 / download of consecutive pieces overlap on three streams
  hipStream_t up_stream = nullptr, down_stream = nullptr;
  HostSlot* hslots = nullptr;  // calls in flight on page-locked arrays (host_submit / host_wait)
  int hnext = 0;
  std::vector<hipEvent_t> pipe_ev;  // [2 * pieces]: uploaded, computed
  int32_t index_base = 0;           // message index of the piece being launched (error reports are batch-relative)
  bool lazy_active = false;  // PER_SLOT: lazy Phase1a promises may be outstanding (k_phase2 runs its lazy-aware form)
  bool packed_pass = false;  // the launch being enqueued is the packed walk over runs of acceptors (k_phase2 MODE 3)
  DevBuf d_run_done;         // one byte per chunk: taken by the packed walk
  // K4: the proxy leader's noop-range tallies (two buffers: fpx_proxy_forget rehashes into the other one)
  RangeTable rt[2];
  int rt_cur = 0;
  DevBuf d_rng;  // staging of range batches
  // fpx_mencius_band_fused_dev: the ranges' half of an independent step runs here, between a fork and a join event
  hipStream_t band_stream = nullptr;
  int64_t band_merged_steps = 0;
  hipEvent_t band_fork = nullptr, band_join = nullptr;
  DevBuf d_band;  // [num_leader_groups] marks: the leader groups with a range in the step being checked
  // multi-GPU (fpx_comm_*): one communicator per context, rank = this context's GPU
  RcclComm comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  int last_rccl = 0;
  DevBuf d_part, d_mine;       // partial vote bitmaps of my acceptors (all slots) / full bitmaps of my slots
  std::vector<hipEvent_t> cev;  // start/stop pairs around the collective
  size_t cev_used = 0;
};

namespace {

#define HIPCHK(ctx, expr)                       \
  do {                                          \
    hipError_t _e = (expr);                     \
    if (_e != hipSuccess) {                     \
      if (ctx) (ctx)->last_hip = (int)_e;       \
      return _e == hipErrorOutOfMemory ? FPX_ENOMEM : FPX_EHIP; \
    }                                           \
  } while (0)

// Every entry point runs with the context's device current and restores the caller's on return: allocations
// (staging buffers, events) and launches otherwise land on whatever device the calling thread last selected --
// two contexts on two GPUs in one process (or a torch.cuda.set_device elsewhere) would fault.
void flush_pending_fin(fpx_ctx* ctx);
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int device) { enter(device); }
  // Every entry point that takes the context starts here; all but fpx_phase2_fused_dev (which enters by device number
  // and deals with a pending fold itself) first launch the fold of the last K3 launch if it is still pending -- so
  // nothing that reads or moves the acceptors' scalars ever sees them short of a launch (see k_phase2_fin).
  explicit DeviceGuard(const fpx_ctx* ctx) {
    if (!ctx) return;
    enter(ctx->cfg.device);
    if (ctx->pending_fin.nblk) flush_pending_fin(const_cast<fpx_ctx*>(ctx));
  }
  void enter(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) switched = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched && prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

int check_config(const fpx_config* c) {
  if (!c) return FPX_EINVAL;
  if (c->num_slots < 1) return FPX_EINVAL;
  if (c->num_replicas < 1 || c->num_replicas > FPX_MAX_REPLICAS) return FPX_EINVAL;
  if (c->num_groups < 1 || c->num_leader_groups < 1) return FPX_EINVAL;
  if (c->num_leaders < 1) return FPX_EINVAL;
  if (c->tally_ways < 1 || c->tally_ways > 8) return FPX_EINVAL;
  if (c->ballot_mode != FPX_BALLOT_ACCEPTOR && c->ballot_mode != FPX_BALLOT_PER_SLOT) return FPX_EINVAL;
  const int total = c->replicas_total ? c->replicas_total : c->num_replicas;
  if (total < 1 || total > FPX_MAX_REPLICAS) return FPX_EINVAL;
  if (c->replica_base < 0 || (c->replica_base & 3) || c->replica_base + c->num_replicas > total) return FPX_EINVAL;
  // the per-block maxima tables live in LDS
  if ((int64_t)c->num_groups * c->num_leader_groups * c->num_replicas > 8192) return FPX_EINVAL;
  switch (c->quorum_kind) {
    case FPX_Q_THRESHOLD:
      if (c->f < 0 || c->f + 1 > total) return FPX_EINVAL;
      break;
    case FPX_Q_SIMPLE_MAJORITY:
    case FPX_Q_UNANIMOUS:
      break;
    case FPX_Q_GRID:
      if (c->grid_rows < 1 || c->grid_cols < 1 || c->grid_rows * c->grid_cols != total) return FPX_EINVAL;
      break;
    default:
      return FPX_EINVAL;
  }
  return FPX_OK;
}

// the row of slot s in the cell arrays and tally tables (device: phys_slot)
static inline size_t host_phys_slot(const Geom& g, int s) {
  return g.lg_rows ? (size_t)(s % g.num_leader_groups) * g.lg_rows + (size_t)(s / g.num_leader_groups) : (size_t)s;
}

void make_geom(const fpx_config& c, Geom* g) {
  memset(g, 0, sizeof(*g));
  g->S = c.num_slots;
  g->R = c.num_replicas;
  g->RS = (c.num_replicas + 3) & ~3;
  // small groups (R <= 4): the vote round and vote value rows of a slot share one 32-byte sector
  { const char* il = getenv("FPX_INTERLEAVE"); g->VS = (g->RS == 4 && il && atoi(il) != 0) ? 8 : g->RS; }
  g->num_groups = c.num_groups;
  g->num_leader_groups = c.num_leader_groups;
  g->ngroups = c.num_groups * c.num_leader_groups;
  // Mencius: rows leader-group-major when the window is a whole number of rounds over the leader groups
  g->lg_rows = (c.num_leader_groups > 1 && c.num_slots % c.num_leader_groups == 0 && c.num_replicas <= 32 &&
                !(c.flags & FPX_F_SLOT_MAJOR_ROWS) && !getenv("FPX_SLOT_MAJOR"))
                   ? c.num_slots / c.num_leader_groups : 0;
  // slot / L and row / A by multiplication (fpx_fastdiv.hpp)
  fast_div_magic(c.num_leader_groups, &g->l_magic, &g->l_shift);
  fast_div_magic(c.num_groups, &g->a_magic, &g->a_shift);
  g->qkind = c.quorum_kind;
  g->total = c.replicas_total ? c.replicas_total : c.num_replicas;
  g->base = c.replica_base;
  switch (c.quorum_kind) {
    case FPX_Q_THRESHOLD: g->qsize = c.f + 1; break;                 // ProxyLeader.scala:238
    case FPX_Q_SIMPLE_MAJORITY: g->qsize = g->total / 2 + 1; break;  // SimpleMajority.scala:30
    case FPX_Q_UNANIMOUS: g->qsize = g->total; break;                // UnanimousWrites.scala:50
    default: g->qsize = 0;
  }
  g->grid_rows = c.grid_rows;
  g->grid_cols = c.grid_cols;
  g->per_slot = c.ballot_mode == FPX_BALLOT_PER_SLOT;
  g->ways = c.tally_ways;
  g->wp = c.tally_ways <= 4 ? 4 : 8;
  for (int w = 0; w < 4; ++w) {
    const int lo = w * 64;
    g->member[w] = g->total >= lo + 64 ? ~0ull : (g->total <= lo ? 0ull : ((1ull << (g->total - lo)) - 1ull));
  }
}

int grow(fpx_ctx* ctx, DevBuf* b, size_t bytes) {
  if (bytes <= b->cap) return FPX_OK;
  if (b->p) HIPCHK(ctx, hipFree(b->p));
  b->p = nullptr;
  b->cap = 0;
  size_t cap = std::max<size_t>(bytes, 4096);
  HIPCHK(ctx, hipMalloc(&b->p, cap));
  b->cap = cap;
  return FPX_OK;
}

template <typename T>
int dalloc(fpx_ctx* ctx, T** p, size_t count) {
  HIPCHK(ctx, hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
  ctx->bytes += (int64_t)(count * sizeof(T));
  return FPX_OK;
}

size_t lds_bytes(const fpx_ctx* ctx, bool fused, bool targets) {
  const size_t tab = (((size_t)ctx->g.ngroups * ctx->g.R * 8) + 16 + 15) & ~(size_t)15;  // tables + 4 workgroup words
  // (targets: per wavefront the chunk's 64 masks + 16 words for the packed walk's Nack path)
  return tab + 4 * (fused ? sizeof(WaveOut<true>) : sizeof(WaveOut<false>)) + (targets ? 4 * (256 + 16) * sizeof(uint64_t) : 0);
}

// messages per wavefront at G = 64: FPX_CHUNK for batches that fill the chip anyway, fewer (down to 4) for small
// ones -- a wave walks its chunk one row at a time, so a 20 k-message epoch in 32-message chunks is 670 waves
// walking 32 dependent steps each on a machine with room for 6000
int chunk_for(const fpx_ctx* ctx, int n) {
  if (ctx->lanes_per_slot != 64) return 64;
  // (FPX_MIN_CHUNK: the sweep of profiles/r06_raw/adversarial_chunk_sweep.txt -- 8 / 4 / 2 / 1 messages per wavefront at ~21 500
  // messages per launch: 6.12 / 6.50 / 6.55 / 6.53e8 proposals/s; 4 stays)
  static const int min_chunk = [] { const char* e = getenv("FPX_MIN_CHUNK"); return e ? std::max(1, atoi(e)) : 4; }();
  const long long want_waves = (long long)ctx->num_cus * 16;
  int ch = FPX_CHUNK;
  while (ch > min_chunk && (long long)(n + ch - 1) / ch < want_waves) ch >>= 1;
  return ch;
}

int grid_for(const fpx_ctx* ctx, int n) {
  const int per_block = 4 * chunk_for(ctx, n);
  const int need = (n + per_block - 1) / per_block;
  return std::max(1, std::min(need, ctx->max_grid));
}

// dynamic LDS above the 64 KiB default needs an opt-in per kernel (gfx950 has 160 KiB per CU): the maxima
// tables of a context with thousands of acceptors (num_groups * num_leader_groups * R up to 8192) reach 83 KiB
template <typename K>
void allow_lds(K kernel, size_t lds) {
  if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

// dynamic LDS of one vote-kernel launch; the offsets of its optional parts go into the batch
size_t phase2_lds(fpx_ctx* ctx, Batch& b, bool fused, bool targets, bool acceptor_rounds) {
  size_t lds = lds_bytes(ctx, fused, targets);
  // leader-group-major rows: 3 KiB of LDS per wavefront for the column quads of a slot-ordered batch (k_phase2), when
  // every array they move as 16 bytes is aligned for it
  auto al = [](const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
  if (ctx->g.lg_rows && al(b.slot, 16) && al(b.round, 16) && al(b.value, 16) && al(b.chosen, 4) && al(b.chosen_round, 16) &&
      al(b.chosen_value, 16) && al(b.nack_round, 16) && !getenv("FPX_NO_QUADS")) {
    lds = (lds + 15) & ~(size_t)15;
    b.sc_lds = (int32_t)lds;
    lds += 4 * 3 * 4 * 64 * sizeof(int32_t);
  }
  if (acceptor_rounds && ctx->g.ngroups != 1) {  // the acceptors' rounds of every group, staged per workgroup
    lds = (lds + 15) & ~(size_t)15;
    b.th_lds = (int32_t)lds;
    lds += (size_t)ctx->g.ngroups * ctx->g.R * sizeof(int32_t);
  }
  return lds;
}

template <int G, int MODE, int PS>
void launch_phase2_3(fpx_ctx* ctx, const Batch& b0, bool fused, int grid) {
  Batch b = b0;
  const size_t lds = phase2_lds(ctx, b, fused, MODE != 0, PS == 0);
  if (fused) allow_lds(k_phase2<G, MODE, PS, true>, lds);
  else allow_lds(k_phase2<G, MODE, PS, false>, lds);
  if constexpr ((G == 64 || G == 1) && (MODE == 0 || MODE == 2) && PS != 0) {
    // the fold of the launch before rides in this one (k_phase2_fin: the shapes of the steady streams -- the headline,
    // configs 2 and 3, the adversarial stream; other shapes leave it to the caller, who launches it by itself)
    if (fused && ctx->carry_fin.nblk) {
      allow_lds(k_phase2_fin<G, MODE, PS, true>, lds);
      hipExtLaunchKernelGGL((k_phase2_fin<G, MODE, PS, true>), dim3(grid + ctx->carry_fin.nblk), dim3(256), (uint32_t)lds, ctx->stream,
                            ctx->ev_start, ctx->ev_stop, 0, ctx->g, ctx->launch_st, b, ctx->carry_fin);
      ctx->carry_fin.nblk = 0;
      ++ctx->fins_carried;
      return;
    }
  }
  if (fused)
    hipExtLaunchKernelGGL((k_phase2<G, MODE, PS, true>), dim3(grid), dim3(256), (uint32_t)lds, ctx->stream, ctx->ev_start,
                          ctx->ev_stop, 0, ctx->g, ctx->launch_st, b);
  else
    hipExtLaunchKernelGGL((k_phase2<G, MODE, PS, false>), dim3(grid), dim3(256), (uint32_t)lds, ctx->stream, ctx->ev_start,
                          ctx->ev_stop, 0, ctx->g, ctx->launch_st, b);
}

template <int G, int MODE>
void launch_phase2_2b(fpx_ctx* ctx, const Batch& b, bool fused, int grid) {
  // ballot model: 0 = per-acceptor scalar, 1 = per cell, 2 = per cell with lazy Phase1a promises outstanding
  if (!ctx->g.per_slot) launch_phase2_3<G, MODE, 0>(ctx, b, fused, grid);
  else if (!ctx->lazy_active) launch_phase2_3<G, MODE, 1>(ctx, b, fused, grid);
  else launch_phase2_3<G, MODE, 2>(ctx, b, fused, grid);
}

template <int G>
void launch_phase2_2(fpx_ctx* ctx, const Batch& b, bool fused, int grid) {
  // rows are padded to a multiple of 4 cells (Geom::RS), so int4 accesses serve every R.
  // mode 0: dense delivery (no target masks); 1: target masks; 2: target masks + FPX_F_SCATTERED_TARGETS
  const int mode = !b.target ? 0 : ((ctx->cfg.flags & FPX_F_SCATTERED_TARGETS) ? 2 : 1);
  if constexpr (G == 64) {
    if (ctx->packed_pass) {  // the packed walk (enqueue_phase2 runs it ahead of the row-at-a-time walk)
      if (!ctx->g.per_slot) launch_phase2_3<64, 3, 0>(ctx, b, fused, grid);
      else launch_phase2_3<64, 3, 1>(ctx, b, fused, grid);
      return;
    }
  }
  if (mode == 0) launch_phase2_2b<G, 0>(ctx, b, fused, grid);
  else if (mode == 1) launch_phase2_2b<G, 1>(ctx, b, fused, grid);
  else launch_phase2_2b<G, 2>(ctx, b, fused, grid);
}

void launch_phase2(fpx_ctx* ctx, const Batch& b, bool fused, int grid) {
  switch (ctx->lanes_per_slot) {
    case 1: launch_phase2_2<1>(ctx, b, fused, grid); break;
    case 2: launch_phase2_2<2>(ctx, b, fused, grid); break;
    case 4: launch_phase2_2<4>(ctx, b, fused, grid); break;
    case 8: launch_phase2_2<8>(ctx, b, fused, grid); break;
    case 16: launch_phase2_2<16>(ctx, b, fused, grid); break;
    case 32: launch_phase2_2<32>(ctx, b, fused, grid); break;
    default: launch_phase2_2<64>(ctx, b, fused, grid); break;
  }
}

// the vote kernel of a Mencius band with the band's range chain as its first workgroup (enqueue_band_merged)
template <int G>
void launch_band_g(fpx_ctx* ctx, const Batch& b, size_t lds, int grid, const RangeTable& rt, const RangeBatch& rb) {
  allow_lds(k_phase2_band<G, 0, 0, true>, lds);
  hipExtLaunchKernelGGL((k_phase2_band<G, 0, 0, true>), dim3(grid + 1), dim3(256), (uint32_t)lds, ctx->stream, ctx->ev_start,
                        ctx->ev_stop, 0, ctx->g, ctx->launch_st, b, rt, rb);
}
void launch_band(fpx_ctx* ctx, const Batch& b, size_t lds, int grid, const RangeTable& rt, const RangeBatch& rb) {
  switch (ctx->lanes_per_slot) {
    case 1: launch_band_g<1>(ctx, b, lds, grid, rt, rb); break;
    case 2: launch_band_g<2>(ctx, b, lds, grid, rt, rb); break;
    case 4: launch_band_g<4>(ctx, b, lds, grid, rt, rb); break;
    default: launch_band_g<8>(ctx, b, lds, grid, rt, rb); break;
  }
}

// fill n int32 words on the stream with a kernel (see k_fill32)
void fill32(fpx_ctx* ctx, void* p, int32_t v, size_t n) {
  if (n == 0) return;
  const int grid = (int)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, (size_t)ctx->num_cus * 8));
  hipLaunchKernelGGL(k_fill32, dim3(grid), dim3(256), 0, ctx->stream, (int32_t*)p, v, n);
}

int launch_check(fpx_ctx* ctx) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ctx->last_hip = (int)e;
    return FPX_EHIP;
  }
  return FPX_OK;
}

// While a host entry point drives its own device runs the run contract holds by construction
// (split_runs), so the device-side validation pass is skipped.
struct HostRun {
  fpx_ctx* ctx;
  explicit HostRun(fpx_ctx* c) : ctx(c) { ctx->host_validated = true; }
  ~HostRun() { ctx->host_validated = false; }
};

// validation pass of one device run (skipped with FPX_F_TRUSTED)
int enqueue_validate(fpx_ctx* ctx, Batch& b, bool check_round) {
  b.run_id = ++ctx->run_id;
  if (ctx->run_id == 0xFFFFFFFFu) {  // stamp space exhausted: start over
    HIPCHK(ctx, hipMemsetAsync(ctx->st.stamp, 0, sizeof(uint32_t) * (size_t)ctx->g.S, ctx->stream));
    ctx->run_id = 0;
    b.run_id = ++ctx->run_id;
  }
  if (((ctx->cfg.flags & FPX_F_TRUSTED) && !ctx->force_validate) || ctx->host_validated) return FPX_OK;
  b.check_round = check_round && !ctx->g.per_slot;
  if (b.check_round) fill32(ctx, ctx->st.run_round, -1, (size_t)ctx->g.ngroups);
  hipLaunchKernelGGL(k_validate, dim3((b.n + 255) / 256), dim3(256), 0, ctx->stream, ctx->g, ctx->st, b);
  return launch_check(ctx);
}

// The launch counter of a K1 / K3 launch decides which third of part_all (Batch::parity) and which half of part /
// part_stamp (the State handed to the launch and to its fold) it uses: see finalize_body / k_phase2_fin.
void begin_phase2_launch(fpx_ctx* ctx, Batch& b) {
  const uint32_t li = ctx->phase2_launches++;
  b.parity = (int32_t)(li % 3u);
  const size_t half = (size_t)(li & 1u), ntab = (size_t)ctx->g.ngroups * ctx->g.R;
  ctx->launch_st = ctx->st;
  ctx->launch_st.part = ctx->st.part + half * (size_t)ctx->max_grid * 2 * ntab;
  ctx->launch_st.part_stamp = ctx->st.part_stamp + half * (size_t)ctx->max_grid;
}

FinJob fin_job_of(const fpx_ctx* ctx, const Batch& b, int grid) {
  FinJob fj;
  const int ntab = ctx->g.ngroups * ctx->g.R;
  fj.fgx = (ntab + 63) / 64;
  fj.slices = std::max(FINALIZE_SLICES, std::min(256, grid / 32));  // ~8 partial rows per wavefront
  fj.nblk = fj.fgx * fj.slices;
  fj.par = (int)b.parity, fj.grid = grid, fj.seq = b.launch_seq;
  fj.part = ctx->launch_st.part, fj.part_stamp = ctx->launch_st.part_stamp;
  return fj;
}

void launch_fin_alone(fpx_ctx* ctx, const FinJob& fj) {
  State fs = ctx->st;
  fs.part = fj.part, fs.part_stamp = fj.part_stamp;
  hipLaunchKernelGGL(k_finalize, dim3(fj.fgx, fj.slices), dim3(256), 0, ctx->stream, ctx->g, fs, fj.par, fj.grid, fj.seq);
}

void flush_pending_fin(fpx_ctx* ctx) {
  if (!ctx->pending_fin.nblk) return;
  launch_fin_alone(ctx, ctx->pending_fin);
  ctx->pending_fin.nblk = 0;
  (void)launch_check(ctx);
}

// K1 / K3 on one device run
int enqueue_phase2(fpx_ctx* ctx, Batch& b, bool fused) {
  if (b.n == 0) return FPX_OK;
  b.index_base = ctx->index_base;
  int rc = enqueue_validate(ctx, b, true);
  if (rc) return rc;
  int grid = grid_for(ctx, b.n);
  b.chunk = chunk_for(ctx, b.n);
  b.index_base = ctx->index_base;
  // at most two chunks per wavefront of one workgroup: that workgroup finalises itself (k_phase2, `solo`); the parity
  // of the partial-maxima buffers is not consumed
  b.solo = ((b.n + b.chunk - 1) / b.chunk <= 8 && !getenv("FPX_NO_SOLO")) ? 1 : 0;
  if (b.solo) grid = 1;
  // Target masks on 256-cell rows of one group: thrifty delivery to RUNS of neighbouring acceptors (what GpuProxyLeader
  // sends by default; any f + 1 will do: ProxyLeader.scala:190-191) takes two rows per wavefront step.  Two launches: the
  // packed walk takes the chunks all of whose messages go to such runs of rows nobody voted in yet and marks them, the
  // row-at-a-time walk behind it takes the rest (nothing is decided on the host, nothing is walked twice).
  const Geom& g = ctx->g;
  const bool two = b.target && ctx->lanes_per_slot == 64 && g.RS == 256 && g.ngroups == 1 && g.base == 0 && g.total == g.R &&
                   g.qkind != 2 && !(g.per_slot && ctx->lazy_active) && !(ctx->cfg.flags & FPX_F_SCATTERED_TARGETS) &&
                   !getenv("FPX_NO_PACKED_RUNS");  // (FPX_F_SCATTERED_TARGETS: the caller says its targets are no runs)
  if (two) {
    rc = grow(ctx, &ctx->d_run_done, (size_t)(b.n + b.chunk - 1) / b.chunk + 64);
    if (rc) return rc;
    b.run_done = (uint8_t*)ctx->d_run_done.p;
  }
  const bool prof = ctx->profiling && ctx->ev_used + 2 <= ctx->ev.size();
  for (int pass = two ? 0 : 1; pass < 2; ++pass) {
    // start of the first vote kernel to end of the last (two with the packed walk ahead)
    ctx->ev_start = prof && pass == (two ? 0 : 1) ? ctx->ev[ctx->ev_used] : nullptr;
    ctx->ev_stop = prof && pass == 1 ? ctx->ev[ctx->ev_used + 1] : nullptr;
    if (b.solo) {
      // (the one workgroup of a solo launch raises the acceptors' scalars with plain loads and stores: nothing may fold
      // beside it -- a pending fold goes first, by itself)
      flush_pending_fin(ctx);
      b.parity = 0, ctx->launch_st = ctx->st;
    } else {
      begin_phase2_launch(ctx, b);
    }
    if (++ctx->launch_seq == 0) ctx->launch_seq = 1;
    b.launch_seq = ctx->launch_seq;
    ctx->packed_pass = pass == 0;
    // the fold of the launch before, if it is still pending: offered to this launch (launch_phase2_3 takes it into its
    // grid when the kernel has that form), else launched by itself behind it -- before the launch after this one either way
    ctx->carry_fin = ctx->pending_fin;
    ctx->pending_fin.nblk = 0;
    launch_phase2(ctx, b, fused, grid);
    ctx->packed_pass = false;
    ctx->ev_start = ctx->ev_stop = nullptr;
    if (ctx->carry_fin.nblk) {
      launch_fin_alone(ctx, ctx->carry_fin);
      ctx->carry_fin.nblk = 0;
    }
    rc = launch_check(ctx);
    if (rc) return rc;
    if (pass == 1 && prof) ctx->ev_used += 2;
    if (b.solo) continue;
    // this launch's own fold: with a ballot per cell no vote kernel reads what it writes -- it waits for the next launch
    // (or for whatever touches the context next); with a round per acceptor the next vote kernel starts from it: at once
    const FinJob fj = fin_job_of(ctx, b, grid);
    const bool no_defer = getenv("FPX_NO_DEFER_FINALIZE") != nullptr;  // (read per launch: a test switches it between contexts)
    if (fused && ctx->g.per_slot && !no_defer) {
      ctx->pending_fin = fj;
    } else {
      launch_fin_alone(ctx, fj);
      rc = launch_check(ctx);
      if (rc) return rc;
    }
  }
  return FPX_OK;
}

int enqueue_open(fpx_ctx* ctx, Batch& b) {
  if (b.n == 0) return FPX_OK;
  int rc = enqueue_validate(ctx, b, false);
  if (rc) return rc;
  hipLaunchKernelGGL(k_open, dim3((b.n + 255) / 256), dim3(256), 0, ctx->stream, ctx->g, ctx->st, b);
  return launch_check(ctx);
}

int enqueue_tally(fpx_ctx* ctx, Batch& b) {
  if (b.n == 0) return FPX_OK;
  int rc = enqueue_validate(ctx, b, false);
  if (rc) return rc;
  hipLaunchKernelGGL(k_tally, dim3((b.n + 255) / 256), dim3(256), 0, ctx->stream, ctx->g, ctx->st, b);
  return launch_check(ctx);
}

// fetch + clear the sticky device status (synchronises the stream)
int fetch_status(fpx_ctx* ctx) {
  int32_t h[4] = {0, 0, 0, 0};
  HIPCHK(ctx, hipMemcpyAsync(h, ctx->st.status, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (h[0] != 0) {
    ctx->err_index = h[ST_INDEX];
    ctx->err_slot = h[ST_SLOT];
    ctx->err_round = h[ST_ROUND];
    HIPCHK(ctx, hipMemsetAsync(ctx->st.status, 0, sizeof(int32_t) * 8, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return h[0];
}

int clear_range_table(fpx_ctx* ctx, const RangeTable& t) {
  if (!t.key) return FPX_OK;
  const size_t words = (size_t)ctx->g.num_groups * 4;
  HIPCHK(ctx, hipMemsetAsync(t.key, 0, (size_t)t.cap * 16, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(t.bits, 0, (size_t)t.cap * words * 8, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(t.owner, 0x7F, (size_t)t.cap * 4, ctx->stream));  // 0x7f7f7f7f: above any message index
  HIPCHK(ctx, hipMemsetAsync(t.count, 0, 4, ctx->stream));
  return FPX_OK;
}

int init_state(fpx_ctx* ctx) {
  const Geom& g = ctx->g;
  State& st = ctx->st;
  const size_t ncell = (size_t)g.S * g.RS, nsc = (size_t)g.ngroups * g.R;
  HIPCHK(ctx, hipMemsetAsync(st.promised, 0xFF, nsc * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(st.max_voted, 0xFF, nsc * 4, ctx->stream));
  if (g.VS != g.RS) {
    HIPCHK(ctx, hipMemsetAsync(st.vote_round, 0xFF, ncell * 8, ctx->stream));  // both rows of every slot, interleaved
  } else {
    HIPCHK(ctx, hipMemsetAsync(st.vote_round, 0xFF, ncell * 4, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(st.vote_value, 0xFF, ncell * 4, ctx->stream));
  }
  if (st.ballot) HIPCHK(ctx, hipMemsetAsync(st.ballot, 0xFF, ncell * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(st.pl_key, 0, (size_t)g.S * g.wp * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(st.pl_value, 0xFF, (size_t)g.S * g.wp * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(st.pl_bits, 0, (size_t)g.S * g.wp * 32, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(st.stamp, 0, (size_t)g.S * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(st.row_voted, 0, (size_t)g.S, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(st.lz_round, 0xFF, (nsc + 4) * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(st.lz_from, 0, (nsc + 4) * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(st.max_ballot, 0xFF, (nsc + 4) * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(st.p1, 0, ((size_t)4 * g.R + 64) * 4, ctx->stream));
  ctx->lazy_active = false, ctx->lz_min_from = 0x7fffffff;
  HIPCHK(ctx, hipMemsetAsync(st.run_round, 0xFF, (size_t)g.ngroups * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(st.status, 0, 8 * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(st.part_stamp, 0, (size_t)2 * ctx->max_grid * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(st.part_all, 0xFF, (size_t)3 * 64 * PART_ALL_STRIDE * 4, ctx->stream));
  ctx->pending_fin.nblk = 0, ctx->carry_fin.nblk = 0;
  HIPCHK(ctx, hipMemsetAsync(st.log_value, 0xFF, (size_t)g.S * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(st.log_present, 0, (size_t)g.S, ctx->stream));
  for (int k = 0; k < 2; ++k) {
    int rc2 = clear_range_table(ctx, ctx->rt[k]);
    if (rc2) return rc2;
  }
  ctx->rt_cur = 0;
  {
    // executedWatermark = 0, numChosen = 0, largestKey = -1, scan result = 0
    static const int32_t init[8] = {0, 0, -1, 0, 0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(st.log_scalars, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
  }
  ctx->run_id = 0;
  ctx->hrun = 0;
  std::fill(ctx->hstamp.begin(), ctx->hstamp.end(), 0u);
  return FPX_OK;
}


// ---- chunk placement of the cell slab (round 5; profiles/r05_placement.md) --------------------------------------------
// The hot kernel streams row s of two or three arrays in lockstep: it reads the ballot row and WRITES the two vote rows.
// How fast that goes depends on which physical memory the rows written together sit in: two 1 GiB regions are either
// compatible (written side by side at 6.7 TB/s) or not (5.6 TB/s, the rate of one write stream alone) -- a property of the
// pair's physical addresses, about half of all pairs each way, identical for every repetition; the read stream's region
// matters a few percent more.  One hipMalloc of the whole slab pairs window w of vote_round with window w of vote_value at
// whatever distance the allocation's shape dictates: stretches of windows at the slow rate, stretches at the fast one (the
// "placement lottery" of rounds 1 - 4: 0.535 - 0.607 ms for the same launch).  So the slab is a reserved address range
// backed by 1 GiB physical allocations (hipMemCreate), and WHICH allocation backs which gigabyte of which array is chosen
// by measurement: for every gigabyte of vote_round a partner for vote_value that probes fast (k_probe: the hot access
// pattern on half of the chunk), then the best of a few candidates for the ballots.  The kernels see one contiguous slab.
constexpr size_t PLACE_CHUNK = (size_t)1 << 30;

struct Placer {
  fpx_ctx* ctx;
  char* base = nullptr;
  size_t reserved = 0;
  std::vector<hipMemGenericAllocationHandle_t> h;  // pool; chunk i is mapped at base + i * PLACE_CHUNK while probing
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int q4 = 64, rows = 1 << 19;
  int probes = 0;
  char* at(int i) const { return base + (size_t)i * PLACE_CHUNK; }
  // the hot access pattern on `rows` rows: write chunks b and c (c < 0: one stream), read chunk a (a < 0: none); min of 2
  float probe(int b, int c, int a) {
    float best = 1e30f;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0, ctx->stream);
      hipLaunchKernelGGL(k_probe, dim3((rows + 127) / 128), dim3(256), 0, ctx->stream, a < 0 ? nullptr : (int32_t*)at(a), (int32_t*)at(b),
                         c < 0 ? nullptr : (int32_t*)at(c), rows, q4, 1, (long long)rows);
      (void)hipEventRecord(e1, ctx->stream);
      if (hipEventSynchronize(e1) != hipSuccess) return -1.f;
      float m = 0;
      if (hipEventElapsedTime(&m, e0, e1) != hipSuccess) return -1.f;
      best = std::min(best, m);
    }
    ++probes;
    return best;
  }
  void release_all() {
    if (base) {
      for (size_t i = 0; i < h.size(); ++i) (void)hipMemUnmap((hipDeviceptr_t)at((int)i), PLACE_CHUNK);
    }
    for (auto x : h) (void)hipMemRelease(x);
    h.clear();
    if (base) (void)hipMemAddressFree((hipDeviceptr_t)base, reserved);
    base = nullptr;
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    e0 = e1 = nullptr;
    (void)hipGetLastError();
  }
};

// Builds the slab of `narr` arrays of `array_bytes` each out of chosen chunks.  On success: *slab_out / *stride_out (the
// arrays' distance, a multiple of the chunk), the context owns the reservation and the chunks.  Any failure leaves nothing
// behind and the caller falls back to one hipMalloc.
bool place_chunks(fpx_ctx* ctx, int narr, size_t array_bytes, int q4, char** slab_out, size_t* stride_out) {
  const int per = (int)((array_bytes + PLACE_CHUNK - 1) / PLACE_CHUNK), need = per * narr;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = ctx->cfg.device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0 || PLACE_CHUNK % gran) {
    (void)hipGetLastError();
    return false;
  }
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
  // spare chunks give the search somewhere to go when the pool's mix is lopsided; they are released afterwards
  int spare = std::min(need / 2 + 4, 24);
  if (const char* e = getenv("FPX_PLACEMENT_SPARE")) spare = std::max(0, std::min(64, atoi(e)));
  while (spare > 0 && free_b < (size_t)(need + spare) * PLACE_CHUNK + ((size_t)8 << 30)) --spare;
  if (free_b < (size_t)need * PLACE_CHUNK + ((size_t)4 << 30)) return false;
  int pool = need + spare;
  Placer P;
  P.ctx = ctx, P.q4 = q4;
  P.rows = (int)std::min<size_t>(PLACE_CHUNK / 2 / ((size_t)q4 * 16), (size_t)1 << 19);
  P.reserved = (size_t)pool * PLACE_CHUNK;
  hipDeviceptr_t va = nullptr;
  if (hipMemAddressReserve(&va, P.reserved, PLACE_CHUNK, nullptr, 0) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  P.base = (char*)va;
  if (hipEventCreate(&P.e0) != hipSuccess || hipEventCreate(&P.e1) != hipSuccess) {
    P.release_all();
    return false;
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  // The pool grows chunk by chunk.  Other contexts may be created on this device at the same moment (the world-N tests:
  // N ranks probing at once), so the one snapshot above is not trusted for the spares: free memory is asked again every
  // 8 chunks, and the first hipMemCreate that fails -- or free memory under the 4 GiB of headroom -- means "stop growing":
  // with the `need` chunks in hand the search goes on without spares, short of them the caller takes one hipMalloc
  // (the spares must not starve another process's allocation, nor cost the placement itself).
  int got = 0;
  for (int i = 0; i < pool; ++i) {
    if (i >= need && (i & 7) == 0) {
      size_t fb2 = 0, tb2 = 0;
      if (hipMemGetInfo(&fb2, &tb2) != hipSuccess || fb2 < PLACE_CHUNK + ((size_t)8 << 30)) break;
    }
    hipMemGenericAllocationHandle_t hd;
    if (hipMemCreate(&hd, PLACE_CHUNK, &prop, 0) != hipSuccess) {
      (void)hipGetLastError();
      break;
    }
    P.h.push_back(hd);
    if (hipMemMap((hipDeviceptr_t)P.at(i), PLACE_CHUNK, 0, hd, 0) != hipSuccess) {
      P.h.pop_back();
      (void)hipMemRelease(hd);
      (void)hipGetLastError();
      break;
    }
    ++got;
  }
  if (got < need) {
    P.release_all();
    return false;
  }
  pool = got;
  if (hipMemSetAccess((hipDeviceptr_t)P.base, (size_t)pool * PLACE_CHUNK, &acc, 1) != hipSuccess) {
    P.release_all();
    return false;
  }
  const bool dbg = getenv("FPX_DEBUG") != nullptr;
  // The search has a budget in wall-clock milliseconds (FPX_PLACEMENT_BUDGET_MS, default 300; VERDICT r05 weak #8 / next #7:
  // eight ranks probing one node at once must not stretch fpx_create): once it is spent every remaining decision takes its
  // first candidate unprobed -- the slab is then placed as allocated from that window on, which is what one hipMalloc does.
  double budget_ms = 300.0;
  if (const char* e = getenv("FPX_PLACEMENT_BUDGET_MS")) budget_ms = atof(e);
  const auto t_start = std::chrono::steady_clock::now();
  auto over_budget = [&]() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count() > budget_ms;
  };
  int unprobed = 0;
  (void)P.probe(0, narr > 1 ? 1 : -1, -1);  // clocks up, code object loaded
  std::vector<int> freec;                   // chunks not assigned yet, in allocation order
  for (int i = 0; i < pool; ++i) freec.push_back(i);
  std::vector<int> pb(per), pc(per, -1), pa(per, -1);
  float pair_min = 1e30f, tri_min = 1e30f;
  bool bad = false;
  // vote_round's chunk of window w is the next free one; vote_value's the first candidate that probes within 6 % of the
  // fastest pair seen (a slow pair is 18 - 20 % off), else the best of up to 16.  Compatible regions come in stretches of
  // tens of gigabytes, so the candidates are taken SPREAD over the free list (its positions in bit-reversed order: the
  // middle, the quarters, the eighths ..), not from its front -- eight neighbours were eight chunks of one kind and left
  // 6 of 25 windows slow (profiles/r05_placement.md)
  auto spread = [](int k, int nfree) {  // the k-th candidate position among nfree
    int bits = 0;
    while ((1 << bits) < nfree) ++bits;
    for (int i = 0, seen = 0; i < (1 << bits); ++i) {
      int r = 0;
      for (int b = 0; b < bits; ++b) r |= ((i >> b) & 1) << (bits - 1 - b);
      if (r < nfree && seen++ == k) return r;
    }
    return k % std::max(1, nfree);
  };
  std::vector<int> aside;  // chunks that found no partner as vote_round's: back into the pool for the ballots / as spares
  for (int w = 0; w < per && !bad; ++w) {
    // up to 4 different chunks for vote_round: late in the search the pool may have run out of partners for the kind of
    // chunk at its front (both other arrays draw from it); such a chunk is set aside and the next one tried
    int fb = -1, fcid = -1;  // the pair so far (chunk ids)
    float ft = 1e30f;
    for (int attempt = 0; attempt < 4 && !bad; ++attempt) {
      // (the windows after this one need two chunks each, and the ballots one per window: ADVICE r05)
      if ((int)freec.size() + (int)aside.size() < 2 + (per - w - 1) * 2 + (narr >= 3 ? per : 0) || freec.size() < 2) break;
      const int b = freec.front();
      freec.erase(freec.begin());
      int best_c = -1;
      float best_t = 1e30f;
      const int nfree = (int)freec.size(), tries = std::min(nfree, 16);
      bool fast = false;
      for (int q = 0; q < tries; ++q) {
        const int c = freec[spread(q, nfree)];
        if (over_budget()) {  // no more probes: the first candidate
          if (best_c < 0) best_c = c, best_t = pair_min < 1e29f ? pair_min : 0.f, ++unprobed;
          fast = true;
          break;
        }
        const float t = P.probe(b, c, -1);
        if (t < 0) { bad = true; break; }
        if (t < best_t) best_t = t, best_c = c;
        pair_min = std::min(pair_min, t);
        if ((w > 0 || q >= 11) && t <= pair_min * 1.06f) { fast = true; break; }  // (window 0 looks at 12 at least: it calibrates pair_min)
      }
      if (bad) break;
      if (w == 0 && best_t <= pair_min * 1.06f) fast = true;
      if (fb < 0 || best_t < ft) {
        if (fb >= 0) aside.push_back(fb);
        fb = b, fcid = best_c, ft = best_t;
      } else {
        aside.push_back(b);
      }
      if (fast) break;
    }
    if (bad || fb < 0 || fcid < 0) { bad = true; break; }
    // the partner is still free: in the pool, or set aside after a turn as vote_round's candidate
    auto take = [&](std::vector<int>& from) {
      auto it = std::find(from.begin(), from.end(), fcid);
      if (it == from.end()) return false;
      from.erase(it);
      return true;
    };
    if (!take(freec) && !take(aside)) { bad = true; break; }
    pb[w] = fb, pc[w] = fcid;
    if (dbg) fprintf(stderr, "libfpx: placement window %d: vote_round chunk %d + vote_value chunk %d: %.4f ms (fastest pair so far %.4f)\n", w, pb[w], pc[w], ft, pair_min);
  }
  freec.insert(freec.end(), aside.begin(), aside.end());
  // the ballots' chunk: the first candidate (spread over what is left, as above) within 1 % of the fastest triple stream
  // seen, else the best of 12 (FPX_PLACEMENT_A_TRIES).  The read stream's region matters as much as the pair's: with three
  // candidates 3 of 25 windows ended 10 % slow, with eight none (profiles/r05_placement.md)
  int a_tries = 12;
  if (const char* e = getenv("FPX_PLACEMENT_A_TRIES")) a_tries = std::max(1, atoi(e));
  std::vector<float> win_ms(per, 0.f);
  for (int w = 0; w < per && !bad; ++w) {
    if (narr < 3) {
      win_ms[w] = over_budget() ? (pair_min < 1e29f ? pair_min : 0.f) : P.probe(pb[w], pc[w], -1);
      continue;
    }
    int best_k = -1;
    float best_t = 1e30f;
    const int left = per - w;  // windows that still need a chunk: never look at more candidates than can be spared
    const int na = std::max(1, std::min({(int)freec.size() - (left - 1), a_tries, 16}));
    for (int q = 0; q < na; ++q) {
      const int k = spread(q, (int)freec.size());
      if (over_budget()) {
        if (best_k < 0) best_k = k, best_t = tri_min < 1e29f ? tri_min : 0.f, ++unprobed;
        break;
      }
      const float t = P.probe(pb[w], pc[w], freec[k]);
      if (t < 0) { bad = true; break; }
      if (t < best_t) best_t = t, best_k = k;
      tri_min = std::min(tri_min, t);
      if (w > 0 && t <= tri_min * 1.01f) break;
    }
    if (bad || best_k < 0) { bad = true; break; }
    pa[w] = freec[best_k];
    freec.erase(freec.begin() + best_k);
    win_ms[w] = best_t;
    if (dbg) fprintf(stderr, "libfpx: placement window %d: ballot chunk %d: %.4f ms (fastest triple so far %.4f)\n", w, pa[w], best_t, tri_min);
  }
  if (bad) {
    P.release_all();
    return false;
  }
  // the final order: array 0 = vote_round, 1 = vote_value, 2 = ballot, `per` chunks each; what is left goes back
  std::vector<hipMemGenericAllocationHandle_t> order;
  for (int w = 0; w < per; ++w) order.push_back(P.h[pb[w]]);
  for (int w = 0; w < per; ++w) order.push_back(P.h[pc[w]]);
  if (narr == 3)
    for (int w = 0; w < per; ++w) order.push_back(P.h[pa[w]]);
  for (int i = 0; i < pool; ++i) (void)hipMemUnmap((hipDeviceptr_t)P.at(i), PLACE_CHUNK);
  for (int i : freec) (void)hipMemRelease(P.h[i]);
  P.h = order;
  bool ok = true;
  for (int i = 0; i < need && ok; ++i) ok = hipMemMap((hipDeviceptr_t)P.at(i), PLACE_CHUNK, 0, P.h[i], 0) == hipSuccess;
  ok = ok && hipMemSetAccess((hipDeviceptr_t)P.base, (size_t)need * PLACE_CHUNK, &acc, 1) == hipSuccess;
  if (!ok) {
    P.release_all();
    return false;
  }
  std::vector<float> sorted(win_ms);
  std::sort(sorted.begin(), sorted.end());
  ctx->placement[0] = 1.f, ctx->placement[1] = (float)per, ctx->placement[2] = sorted.front(), ctx->placement[3] = sorted[per / 2],
  ctx->placement[4] = sorted.back();
  if (dbg)
    fprintf(stderr, "libfpx: slab of %d x %d chunks of 1 GiB placed with %d probes in %.0f ms (pool %d, budget %.0f ms, %d decisions unprobed): per-window probe min %.4f median %.4f max %.4f ms\n",
            narr, per, P.probes, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(), pool, budget_ms,
            unprobed, sorted.front(), sorted[per / 2], sorted.back());
  ctx->placement_probes = P.probes, ctx->placement_unprobed = unprobed;
  ctx->placement_ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
  (void)hipEventDestroy(P.e0);
  (void)hipEventDestroy(P.e1);
  ctx->vmm_chunks = P.h;
  ctx->vmm_reserved = P.reserved;
  ctx->slab = P.base;
  *slab_out = P.base;
  *stride_out = (size_t)per * PLACE_CHUNK;
  return true;
}

void free_state(fpx_ctx* ctx) {
  State& st = ctx->st;
  if (ctx->vmm_reserved) {  // the slab is a reservation backed by chunks
    for (size_t i = 0; i < ctx->vmm_chunks.size(); ++i) (void)hipMemUnmap((hipDeviceptr_t)((char*)ctx->slab + i * PLACE_CHUNK), PLACE_CHUNK);
    for (auto h : ctx->vmm_chunks) (void)hipMemRelease(h);
    ctx->vmm_chunks.clear();
    (void)hipMemAddressFree((hipDeviceptr_t)ctx->slab, ctx->vmm_reserved);
    ctx->slab = nullptr, ctx->vmm_reserved = 0;
  }
  void* ps[] = {st.promised, st.max_voted, ctx->slab, st.pl_key, st.pl_value,
                st.pl_bits,  st.stamp,     st.run_round,  st.status,     st.part,
                st.log_value, st.log_present, st.log_scalars, st.part_stamp, st.part_all,
                st.row_voted, st.lz_round, st.lz_from, st.max_ballot, st.p1,
                ctx->rt[0].key, ctx->rt[0].bits, ctx->rt[0].owner, ctx->rt[0].count,
                ctx->rt[1].key, ctx->rt[1].bits, ctx->rt[1].owner, ctx->rt[1].count, ctx->d_rng.p, ctx->d_run_done.p};
  for (void* p : ps)
    if (p) (void)hipFree(p);
  DevBuf* bs[] = {&ctx->d_slot,   &ctx->d_round, &ctx->d_value, &ctx->d_target, &ctx->d_bits_a, &ctx->d_bits_b,
                  &ctx->d_i32_a,  &ctx->d_i32_b, &ctx->d_i32_c, &ctx->d_u8,     &ctx->d_scratch};
  for (DevBuf* b : bs)
    if (b->p) (void)hipFree(b->p);
  for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
  ctx->ev.clear();
  for (hipEvent_t e : ctx->cev) (void)hipEventDestroy(e);
  ctx->cev.clear();
  for (hipEvent_t e : ctx->pipe_ev) (void)hipEventDestroy(e);
  ctx->pipe_ev.clear();
  if (ctx->hslots) {
    for (int k = 0; k < HOST_DEPTH; ++k) {
      HostSlot& h = ctx->hslots[k];
      for (DevBuf& b : h.in)
        if (b.p) (void)hipFree(b.p);
      for (DevBuf& b : h.sink)
        if (b.p) (void)hipFree(b.p);
      if (h.done) (void)hipEventDestroy(h.done);
      if (h.up) (void)hipEventDestroy(h.up);
      if (h.status) (void)hipHostFree(h.status);
    }
    delete[] ctx->hslots;
    ctx->hslots = nullptr;
  }
  if (ctx->band_fork) (void)hipEventDestroy(ctx->band_fork);
  if (ctx->band_join) (void)hipEventDestroy(ctx->band_join);
  if (ctx->band_stream) (void)hipStreamDestroy(ctx->band_stream);
  if (ctx->d_band.p) (void)hipFree(ctx->d_band.p);
  if (ctx->up_stream) (void)hipStreamDestroy(ctx->up_stream);
  if (ctx->down_stream) (void)hipStreamDestroy(ctx->down_stream);
  if (ctx->d_part.p) (void)hipFree(ctx->d_part.p);
  if (ctx->d_mine.p) (void)hipFree(ctx->d_mine.p);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
}

// ---- host-side run splitting ---------------------------------------------------------------------
// Cuts [0, n) into maximal runs that satisfy the run contract, so that running them back to back
// equals message-at-a-time delivery in array order.
int host_group(const fpx_config& c, int slot) {
  const int lg = slot % c.num_leader_groups;
  const int ag = (slot / c.num_leader_groups) % c.num_groups;
  return lg * c.num_groups + ag;
}

void split_runs(fpx_ctx* ctx, int n, const int32_t* slot, const int32_t* round, bool check_round,
                std::vector<int>* cuts) {
  cuts->clear();
  cuts->push_back(0);
  // check_inputs saw strictly increasing slots (hence distinct) in one round: one run
  const bool one_run = ctx->batch_increasing && (ctx->batch_one_round || !check_round || ctx->g.per_slot);
  ctx->batch_increasing = ctx->batch_one_round = false;  // findings are about one batch only
  if (one_run) {
    cuts->push_back(n);
    return;
  }
  if (ctx->hstamp.empty()) ctx->hstamp.assign((size_t)ctx->g.S, 0u);
  if (ctx->hround.empty()) ctx->hround.assign((size_t)ctx->g.ngroups, -1);
  check_round = check_round && !ctx->g.per_slot;
  std::vector<int> touched;
  auto new_run = [&]() {
    if (++ctx->hrun == 0) {
      std::fill(ctx->hstamp.begin(), ctx->hstamp.end(), 0u);
      ctx->hrun = 1;
    }
    for (int gidx : touched) ctx->hround[gidx] = -1;
    touched.clear();
  };
  new_run();
  for (int i = 0; i < n; ++i) {
    const int s = slot[i];
    bool cut = ctx->hstamp[s] == ctx->hrun;
    int gidx = 0;
    if (check_round) {
      gidx = ctx->g.ngroups == 1 ? 0 : host_group(ctx->cfg, s);
      if (ctx->hround[gidx] != -1 && ctx->hround[gidx] != round[i]) cut = true;
    }
    if (cut) {
      cuts->push_back(i);
      new_run();
    }
    ctx->hstamp[s] = ctx->hrun;
    if (check_round) {
      if (ctx->hround[gidx] == -1) touched.push_back(gidx);
      ctx->hround[gidx] = round[i];
    }
  }
  cuts->push_back(n);
  for (int gidx : touched) ctx->hround[gidx] = -1;
}

int check_args(fpx_ctx* ctx, int n, const int32_t* slot, const int32_t* round) {
  return (!ctx || n < 0 || (n > 0 && (!slot || !round))) ? FPX_EINVAL : FPX_OK;
}

// range check of a host batch; the host entry points run it (and split_runs) AFTER enqueueing the uploads,
// so with page-locked buffers (fpx_host_alloc) the CPU passes overlap the DMA.  Nothing is launched on
// a bad batch: the staging copies alone change no protocol state.
int check_inputs(fpx_ctx* ctx, int n, const int32_t* slot, const int32_t* round) {
  // one branch-free pass (vectorises): range of slots and rounds, and whether the batch is the common
  // shape -- strictly increasing slots in one round -- that split_runs can accept as a single run
  int32_t smin = INT32_MAX, smax = INT32_MIN, rmin = INT32_MAX, rmax = INT32_MIN;
  int32_t unordered = 0;
  for (int i = 0; i < n; ++i) {
    smin = std::min(smin, slot[i]), smax = std::max(smax, slot[i]);
    rmin = std::min(rmin, round[i]), rmax = std::max(rmax, round[i]);
  }
  for (int i = 1; i < n; ++i) unordered |= (int32_t)(slot[i] <= slot[i - 1]);
  ctx->batch_increasing = unordered == 0;
  ctx->batch_one_round = rmin == rmax;
  if (smin >= 0 && smax < ctx->g.S && rmin >= 0 && rmax <= MAX_ROUND) return FPX_OK;
  for (int i = 0; i < n; ++i) {
    if (slot[i] < 0 || slot[i] >= ctx->g.S || round[i] < 0 || round[i] > MAX_ROUND) {
      ctx->err_index = i;
      ctx->err_slot = slot[i];
      ctx->err_round = round[i];
      break;
    }
  }
  ctx->batch_increasing = ctx->batch_one_round = false;
  (void)hipStreamSynchronize(ctx->stream);  // the caller's buffers are free again when we return
  return FPX_EINVAL;
}

template <typename T>
int h2d(fpx_ctx* ctx, DevBuf* b, const T* src, size_t count) {
  int rc = grow(ctx, b, count * sizeof(T));
  if (rc) return rc;
  if (src && count) HIPCHK(ctx, hipMemcpyAsync(b->p, src, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return FPX_OK;
}

template <int G>
void launch_p1b(fpx_ctx* ctx, const uint64_t* d_q, int wm, int count, int32_t* d_sr, int32_t* d_sv) {
  const int per_block = 4 * (64 / G);
  const int grid = std::max(1, std::min((count + per_block - 1) / per_block, ctx->num_cus * 16));
  hipLaunchKernelGGL((k_phase1b_scan<G>), dim3(grid), dim3(256), 0, ctx->stream, ctx->g, ctx->st, d_q, wm, count,
                     ctx->vec ? 1 : 0, d_sr, d_sv);
}

template <typename T>
int d2h(fpx_ctx* ctx, T* dst, const DevBuf& b, size_t count) {
  if (dst && count) HIPCHK(ctx, hipMemcpyAsync(dst, b.p, count * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
  return FPX_OK;
}


// ---- host-pointer K3 helpers --------------------------------------------------------------------------------
// the host-split replay of [from, n) of a staged batch (the slices are still in the staging buffers): cut into
// runs on the host, launch them back to back, download that part of the outputs
int host_fused_replay(fpx_ctx* ctx, int n, const int32_t* slot, const int32_t* round, bool has_target, int from,
                      uint8_t* chosen, int32_t* chosen_round, int32_t* chosen_value, int32_t* nack_round) {
  int32_t *d_slot = (int32_t*)ctx->d_slot.p, *d_round = (int32_t*)ctx->d_round.p, *d_value = (int32_t*)ctx->d_value.p;
  uint64_t* d_target = has_target ? (uint64_t*)ctx->d_target.p : nullptr;
  uint8_t* d_ch = (uint8_t*)ctx->d_u8.p;
  int32_t *d_cr = (int32_t*)ctx->d_i32_a.p, *d_cv = (int32_t*)ctx->d_i32_b.p, *d_nr = (int32_t*)ctx->d_i32_c.p;
  const int len_all = n - from;
  int rc;
  if ((rc = check_inputs(ctx, len_all, slot + from, round + from))) {
    ctx->err_index += from;
    return rc;
  }
  HostRun host_run(ctx);
  std::vector<int> cuts;
  split_runs(ctx, len_all, slot + from, round + from, true, &cuts);
  for (size_t k = 0; k + 1 < cuts.size(); ++k) {
    const int lo = from + cuts[k], len = cuts[k + 1] - cuts[k];
    ctx->index_base = lo;
    rc = fpx_phase2_fused_dev(ctx, len, d_slot + lo, d_round + lo, d_value + lo,
                              d_target ? d_target + (size_t)lo * 4 : nullptr, d_ch + lo, d_cr + lo, d_cv + lo, d_nr + lo);
    ctx->index_base = 0;
    if (rc) {
      (void)hipStreamSynchronize(ctx->stream);
      return rc;
    }
  }
  const size_t cnt = (size_t)len_all;
  if (chosen) HIPCHK(ctx, hipMemcpyAsync(chosen + from, d_ch + from, cnt, hipMemcpyDeviceToHost, ctx->stream));
  if (chosen_round) HIPCHK(ctx, hipMemcpyAsync(chosen_round + from, d_cr + from, cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (chosen_value) HIPCHK(ctx, hipMemcpyAsync(chosen_value + from, d_cv + from, cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (nack_round) HIPCHK(ctx, hipMemcpyAsync(nack_round + from, d_nr + from, cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
  return fetch_status(ctx);
}

// piece size of the pipelined host path.  OFF unless FPX_HOST_PIECE is set: measured on two boxes
// (profiles/r02_host_path.txt) the three-stream pipeline buys nothing over upload -> K3 -> download in sequence --
// 1.90-2.07 ms against 1.96 ms per 2^20 messages on the slow-PCIe box -- as in round 1: the host copies do not
// overlap a kernel that saturates HBM on this stack.  At most 64 pieces.
int host_piece(const fpx_ctx* ctx, int n) {
  (void)ctx;
  const char* e = getenv("FPX_HOST_PIECE");
  if (!e || !*e) return n > 0 ? n : 1;
  int piece = std::max(1024, atoi(e));
  while ((long long)piece * 64 < n) piece <<= 1;
  return piece;
}

// Big host batches: the three stages of the host-pointer K3 -- upload (12 B per message), the fused step, download
// (13 B per message) -- run as a pipeline over pieces on three streams (opt-in, see host_piece).  Pieces are
// separate device runs, launched in order on the context's stream: sequential semantics across pieces for free.
// Range errors are found on the host BEFORE anything is launched (FPX_EINVAL => nothing applied, as ever); a run
// -contract violation inside a piece aborts that piece and everything after it on the device, and the host-split
// replay takes over from that piece's first message.
int host_fused_pipelined(fpx_ctx* ctx, int n, const int32_t* slot, const int32_t* round, const int32_t* value_id,
                         const uint64_t* target_mask, uint8_t* chosen, int32_t* chosen_round, int32_t* chosen_value,
                         int32_t* nack_round) {
  int rc;
  if ((rc = check_inputs(ctx, n, slot, round))) return rc;
  ctx->batch_increasing = ctx->batch_one_round = false;
  if ((rc = grow(ctx, &ctx->d_slot, (size_t)n * 4))) return rc;
  if ((rc = grow(ctx, &ctx->d_round, (size_t)n * 4))) return rc;
  if ((rc = grow(ctx, &ctx->d_value, (size_t)n * 4))) return rc;
  if (target_mask && (rc = grow(ctx, &ctx->d_target, (size_t)n * 32))) return rc;
  const int piece = host_piece(ctx, n), pieces = (n + piece - 1) / piece;
  if (!ctx->up_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking));
  if (!ctx->down_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking));
  while (ctx->pipe_ev.size() < (size_t)2 * pieces) {
    hipEvent_t e;
    HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->pipe_ev.push_back(e);
  }
  int32_t *d_slot = (int32_t*)ctx->d_slot.p, *d_round = (int32_t*)ctx->d_round.p, *d_value = (int32_t*)ctx->d_value.p;
  uint64_t* d_target = target_mask ? (uint64_t*)ctx->d_target.p : nullptr;
  uint8_t* d_ch = (uint8_t*)ctx->d_u8.p;
  int32_t *d_cr = (int32_t*)ctx->d_i32_a.p, *d_cv = (int32_t*)ctx->d_i32_b.p, *d_nr = (int32_t*)ctx->d_i32_c.p;
  // every early return below (HIPCHK) must leave the context as it found it: a FPX_F_TRUSTED context that kept
  // force_validate / index_base would silently validate and report shifted indices ever after
  struct PieceGuard {
    fpx_ctx* c;
    ~PieceGuard() { c->force_validate = false, c->index_base = 0; }
  } _pg{ctx};
  ctx->force_validate = true;
  for (int k = 0; k < pieces && rc == FPX_OK; ++k) {
    const int lo = k * piece, len = std::min(piece, n - lo);
    const size_t c = (size_t)len;
    hipEvent_t up = ctx->pipe_ev[2 * k], done = ctx->pipe_ev[2 * k + 1];
    HIPCHK(ctx, hipMemcpyAsync(d_slot + lo, slot + lo, c * 4, hipMemcpyHostToDevice, ctx->up_stream));
    HIPCHK(ctx, hipMemcpyAsync(d_round + lo, round + lo, c * 4, hipMemcpyHostToDevice, ctx->up_stream));
    HIPCHK(ctx, hipMemcpyAsync(d_value + lo, value_id + lo, c * 4, hipMemcpyHostToDevice, ctx->up_stream));
    if (target_mask)
      HIPCHK(ctx, hipMemcpyAsync(d_target + (size_t)lo * 4, target_mask + (size_t)lo * 4, c * 32, hipMemcpyHostToDevice, ctx->up_stream));
    HIPCHK(ctx, hipEventRecord(up, ctx->up_stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, up, 0));
    ctx->index_base = lo;
    rc = fpx_phase2_fused_dev(ctx, len, d_slot + lo, d_round + lo, d_value + lo, d_target ? d_target + (size_t)lo * 4 : nullptr,
                              d_ch + lo, d_cr + lo, d_cv + lo, d_nr + lo);
    ctx->index_base = 0;
    if (rc) break;
    HIPCHK(ctx, hipEventRecord(done, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->down_stream, done, 0));
    if (chosen) HIPCHK(ctx, hipMemcpyAsync(chosen + lo, d_ch + lo, c, hipMemcpyDeviceToHost, ctx->down_stream));
    if (chosen_round) HIPCHK(ctx, hipMemcpyAsync(chosen_round + lo, d_cr + lo, c * 4, hipMemcpyDeviceToHost, ctx->down_stream));
    if (chosen_value) HIPCHK(ctx, hipMemcpyAsync(chosen_value + lo, d_cv + lo, c * 4, hipMemcpyDeviceToHost, ctx->down_stream));
    if (nack_round) HIPCHK(ctx, hipMemcpyAsync(nack_round + lo, d_nr + lo, c * 4, hipMemcpyDeviceToHost, ctx->down_stream));
  }
  ctx->force_validate = false;
  (void)hipStreamSynchronize(ctx->up_stream);
  const hipError_t dsync = hipStreamSynchronize(ctx->down_stream);
  if (rc) {
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
  }
  rc = fetch_status(ctx);
  if (dsync != hipSuccess) {
    ctx->last_hip = (int)dsync;
    return FPX_EHIP;
  }
  if (rc != FPX_EORDER) return rc;
  // the piece that holds the first offender and everything after it applied nothing: replay from its first message
  const int from = (ctx->err_index / piece) * piece;
  return host_fused_replay(ctx, n, slot, round, target_mask != nullptr, from, chosen, chosen_round, chosen_value, nack_round);
}

// ---- page-locked host batches (fpx_phase2_fused_submit / _wait) ------------------------------------------------------
// A call moves 12 B per slot up (slot, round, value) and 9 - 13 B per slot down (Chosen records, Nack rounds).  How, was
// decided by measuring what each way of crossing PCIe costs the vote kernel it runs beside (profiles/r06_host_path.md,
// microbench/r06_pcie_beside_vote.py; 2^20 x 256 step, vote kernel alone 0.546 ms):
//   inputs by the copy engine on a stream of its own (hipMemcpyAsync, 51 GB/s, 3 x 82 us)             + 3 %
//   inputs read by the vote kernel straight from the mapped arrays                                    + 12 %
//   inputs copied by WORKGROUPS (a staging kernel beside the vote kernel: rounds 2 - 5; or workgroups of the vote
//     kernel's own grid)                                                       + 14 % ... + 50 % with the requests in flight
//   records written by the vote kernel straight into the mapped arrays (posted writes)               + 3 %
//   records copied by workgroups                                                                      + 30 %
// (a shader's reads of host memory hold entries of the L2's fabric queues for microseconds each and the vote kernel's HBM
// requests queue behind them; the copy engine does not go through the L2.)  So: the inputs go up by the copy engine on
// `up_stream` into staging buffers in HBM, validation and the fused step run on the context's stream behind ONE event,
// and the vote kernel's output arrays ARE the caller's page-locked arrays -- no copy down, no third stream, no staging
// kernel.  The status words of the call are snapshotted into page-locked words by a one-wavefront kernel behind the step.
// Every array must be mapped host memory (FPX_EINVAL otherwise).

// the device address of mapped (page-locked) host memory, nullptr for anything else
void* mapped_host(const void* p) {
  if (!p) return nullptr;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();  // pageable memory: "invalid value", not an error of ours
    return nullptr;
  }
  return at.type == hipMemoryTypeHost ? at.devicePointer : nullptr;
}

int host_submit(fpx_ctx* ctx, int n, const int32_t* slot, const int32_t* round, const int32_t* value_id,
                const uint64_t* target_mask, uint8_t* chosen, int32_t* chosen_round, int32_t* chosen_value,
                int32_t* nack_round, int* ticket) {
  const void* in[4] = {slot, round, value_id, target_mask};
  void* out[4] = {chosen, chosen_round, chosen_value, nack_round};
  void* dout[4];
  for (int a = 0; a < 4; ++a) {
    dout[a] = mapped_host(out[a]);
    if ((in[a] && !mapped_host(in[a])) || (out[a] && !dout[a])) return FPX_EINVAL;
  }
  if (!ctx->hslots) {
    ctx->hslots = new (std::nothrow) HostSlot[HOST_DEPTH];
    if (!ctx->hslots) return FPX_ENOMEM;
  }
  int t = -1;
  for (int k = 0; k < HOST_DEPTH && t < 0; ++k)
    if (!ctx->hslots[(ctx->hnext + k) % HOST_DEPTH].busy) t = (ctx->hnext + k) % HOST_DEPTH;
  if (t < 0) return FPX_ECAPACITY;  // HOST_DEPTH calls in flight: wait for the oldest first
  HostSlot& h = ctx->hslots[t];
  if (!h.up) {
    HIPCHK(ctx, hipEventCreateWithFlags(&h.up, hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&h.done, hipEventDisableTiming));
    HIPCHK(ctx, hipHostMalloc((void**)&h.status, 64, hipHostMallocDefault));
    HIPCHK(ctx, hipHostGetDevicePointer((void**)&h.status_dev, h.status, 0));
  }
  if (!ctx->up_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking));
  const size_t in_elem[4] = {4, 4, 4, 32};
  int rc;
  for (int a = 0; a < 4; ++a)
    if (in[a] && (rc = grow(ctx, &h.in[a], (size_t)n * in_elem[a]))) return rc;
  const size_t out_elem[3] = {1, 4, 4};
  for (int a = 0; a < 3; ++a)
    if (!dout[a]) {  // (nack_round alone may be null for the vote kernel)
      if ((rc = grow(ctx, &h.sink[a], (size_t)n * out_elem[a]))) return rc;
      dout[a] = h.sink[a].p;
    }
  for (int a = 0; a < 4; ++a)
    if (in[a]) HIPCHK(ctx, hipMemcpyAsync(h.in[a].p, in[a], (size_t)n * in_elem[a], hipMemcpyHostToDevice, ctx->up_stream));
  HIPCHK(ctx, hipEventRecord(h.up, ctx->up_stream));
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, h.up, 0));
  {
    struct Guard {
      fpx_ctx* c;
      ~Guard() { c->force_validate = false; }
    } _g{ctx};
    ctx->force_validate = true;  // also under FPX_F_TRUSTED: that flag is a promise about _dev batches only
    rc = fpx_phase2_fused_dev(ctx, n, (int32_t*)h.in[0].p, (int32_t*)h.in[1].p, (int32_t*)h.in[2].p,
                              target_mask ? (uint64_t*)h.in[3].p : nullptr, (uint8_t*)dout[0], (int32_t*)dout[1],
                              (int32_t*)dout[2], (int32_t*)dout[3]);
  }
  if (rc) {
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
  }
  hipLaunchKernelGGL(k_status_snap, dim3(1), dim3(64), 0, ctx->stream, ctx->st.status, h.status_dev);
  HIPCHK(ctx, hipEventRecord(h.done, ctx->stream));
  if ((rc = launch_check(ctx))) return rc;
  h.busy = true, h.n = n;
  ctx->hnext = (t + 1) % HOST_DEPTH;
  *ticket = t;
  return FPX_OK;
}

// the status of call `ticket` as the device left it right after that call's fused step (a run-contract violation or a
// range error aborts the call -- and the calls queued behind it -- before anything is applied)
int host_wait(fpx_ctx* ctx, int ticket) {
  if (!ctx->hslots || ticket < 0 || ticket >= HOST_DEPTH || !ctx->hslots[ticket].busy) return FPX_EINVAL;
  HostSlot& h = ctx->hslots[ticket];
  const hipError_t e = hipEventSynchronize(h.done);
  h.busy = false;
  if (e != hipSuccess) {
    ctx->last_hip = (int)e;
    return FPX_EHIP;
  }
  const int st = h.status[0];
  if (st != 0) {
    // drain and clear the sticky status (the calls behind this one have aborted too and report it from their own copy)
    const int now = fetch_status(ctx);
    if (now == 0) ctx->err_index = h.status[ST_INDEX], ctx->err_slot = h.status[ST_SLOT], ctx->err_round = h.status[ST_ROUND];
  }
  return st;
}

// the synchronous call on page-locked arrays = submit + wait; *used = false: not every array is mapped host memory (or
// the knob is off, or calls are in flight) -- the caller takes the copy-engine path
int host_fused_staged(fpx_ctx* ctx, int n, const int32_t* slot, const int32_t* round, const int32_t* value_id,
                      const uint64_t* target_mask, uint8_t* chosen, int32_t* chosen_round, int32_t* chosen_value,
                      int32_t* nack_round, bool* used) {
  *used = false;
  if (getenv("FPX_HOST_NO_STAGE")) return FPX_OK;
  int ticket = -1;
  int rc = host_submit(ctx, n, slot, round, value_id, target_mask, chosen, chosen_round, chosen_value, nack_round, &ticket);
  if (rc == FPX_EINVAL || rc == FPX_ECAPACITY) return FPX_OK;  // not page-locked / ring busy: the other path
  *used = true;
  if (rc) return rc;
  rc = host_wait(ctx, ticket);
  if (rc == FPX_EINVAL) return check_inputs(ctx, n, slot, round);  // the FIRST offender, for fpx_error_detail
  if (rc != FPX_EORDER) return rc;
  // not one device run: cut into runs on the host and replayed (from the copy-engine path's staging buffers)
  *used = false;
  return FPX_OK;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
// ---- the wire adapter on the device (include/fpx_wire.h) ---------------------------------------------------
template <int WHICH>
static int32_t wire_decode_dev(fpx_ctx* ctx, const uint8_t* d_buf, int64_t buf_len, const int64_t* d_offsets, int32_t n,
                               const WireOut& o) {
  DeviceGuard _dg(ctx);
  if (!ctx || n < 0 || n >= (1 << 30) || buf_len < 0) return FPX_EINVAL;
  if (n == 0) return FPX_OK;
  if (!d_buf || !d_offsets || !o.kind || !o.slot || !o.round) return FPX_EINVAL;
  hipLaunchKernelGGL(k_wire_decode<WHICH>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->st, d_buf, buf_len,
                     d_offsets, n, o);
  hipLaunchKernelGGL(k_wire_tail, dim3(1), dim3(1), 0, ctx->stream, ctx->st);
  HIPCHK(ctx, hipGetLastError());
  return FPX_OK;
}

extern "C" {

int32_t fpx_version(void) { return FPX_VERSION; }

const char* fpx_strerror(int32_t s) {
  switch (s) {
    case FPX_OK: return "ok";
    case FPX_EINVAL: return "invalid argument (require failed)";
    case FPX_EFATAL_UNKNOWN_SLOTROUND: return "Phase2b for a (slot, round) that was never opened (logger.fatal)";
    case FPX_EHIP: return "HIP runtime error";
    case FPX_ENODEVICE: return "no usable gfx950 device (libfpx has no CPU fallback)";
    case FPX_ECAPACITY: return "more live (slot, round) tallies for one slot than tally_ways";
    case FPX_EORDER: return "device batch violates the run contract; nothing was applied";
    case FPX_ENOMEM: return "out of device memory";
    case FPX_ERCCL: return "RCCL error (or RCCL could not be loaded)";
    case FPX_EFATAL_PROTOCOL: return "a logger.fatal / logger.check of the reference would have fired; the message was skipped";
    default: return "unknown status";
  }
}

int32_t fpx_config_check(const fpx_config* cfg) { return check_config(cfg); }

int32_t fpx_round_leader(int32_t n, int32_t round) { return round % n; }

int32_t fpx_next_classic_round(int32_t n, int32_t leader_index, int32_t round) {
  if (round < 0) return leader_index;
  const int32_t smallest_multiple = n * (round / n);
  const int32_t offset = leader_index % n;
  return smallest_multiple + offset > round ? smallest_multiple + offset : smallest_multiple + n + offset;
}

int32_t fpx_create(const fpx_config* cfg, fpx_ctx** out) {
  if (!out) return FPX_EINVAL;
  *out = nullptr;
  int rc = check_config(cfg);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev)
    return FPX_ENODEVICE;
  fpx_ctx* ctx = new (std::nothrow) fpx_ctx();
  if (!ctx) return FPX_ENOMEM;
  ctx->cfg = *cfg;
  if (!ctx->cfg.replicas_total) ctx->cfg.replicas_total = cfg->num_replicas;
  make_geom(ctx->cfg, &ctx->g);
  memset(&ctx->st, 0, sizeof(ctx->st));
  memset(ctx->rt, 0, sizeof(ctx->rt));
  DeviceGuard _dg(cfg->device);  // the caller's current device is restored on every return path
  if (hipSetDevice(cfg->device) != hipSuccess) {
    delete ctx;
    return FPX_ENODEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
  // 32 workgroups per CU: at 2^20 messages every wavefront gets exactly one FPX_CHUNK-message chunk and
  // the hardware dispatcher load-balances them (measured r01: +10 % over a 2048-workgroup persistent
  // grid, a further +8 % going from 64- to 32-message chunks)
  int G = 1;
  while (G * 4 < ctx->g.R) G <<= 1;
  ctx->lanes_per_slot = G;
  // small groups (R <= 16, the reference's everyday f = 1..3): a 64-slot chunk is three dependent round
  // trips and a workgroup's fixed costs (LDS tables, the final reduction) dominate -- 4 workgroups per CU,
  // each wave walking many chunks, measured 2x faster at R = 3 (profiles/r01_small_r.txt).  Six per CU is what the kernel's
  // LDS (23.5 KB with the column-quad staging) lets be resident at once: 3 - 4 % faster than four on BASELINE.json configs[4],
  // 5 - 15 % on configs[2]; seven and more (a second round of workgroups) are 8 % slower (profiles/r05_cfg5.md)
  ctx->max_grid = ctx->num_cus * (G <= 4 ? 6 : 32);
  if (const char* e = getenv("FPX_MAX_GRID")) ctx->max_grid = std::max(1, atoi(e));  // tuning aid
  ctx->vec = true;  // rows are padded to a multiple of 4 cells (Geom::RS)
  // big tables: fewer resident blocks so that the partial table stays small
  const int ntab = ctx->g.ngroups * ctx->g.R;
  if (ntab > 1024) ctx->max_grid = std::max(ctx->num_cus, ctx->max_grid * 1024 / ntab);

  auto fail = [&](int code) {
    free_state(ctx);
    delete ctx;
    return code;
  };
  if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) return fail(FPX_EHIP);
  ctx->stream = ctx->own_stream;
  const Geom& g = ctx->g;
  State& st = ctx->st;
  const size_t ncell = (size_t)g.S * g.RS, nsc = (size_t)g.ngroups * g.R;
  if ((rc = dalloc(ctx, &st.promised, nsc))) return fail(rc);
  if ((rc = dalloc(ctx, &st.max_voted, nsc))) return fail(rc);
  {
    // the 2-3 cell arrays are streamed in lockstep (row s of each at the same time): one slab, the
    // arrays staggered by FPX_STAGGER bytes so that their rows do not all start in the same DRAM
    // channel / bank (tuning aid; measured in profiles/r01_stagger.txt)
    size_t stagger = 0;
    if (const char* e = getenv("FPX_STAGGER")) stagger = (size_t)atoll(e) & ~(size_t)15;
    const int narr = g.per_slot ? 3 : 2;
    size_t stride = ncell * 4 + stagger;
    // Placement: the very same kernel streams a slab at one of two speeds ~7 % apart depending on where the
    // allocation landed (deterministic per allocation, profiles/r02_placement.txt).  A big context therefore
    // allocates up to FPX_PLACEMENT_TRIES slabs (default 4, while free HBM allows), times the hot access pattern
    // on each (k_probe, 3 x 0.5 ms) and keeps the fastest; the others are freed before anything else is allocated.
    int tries = stride * narr >= ((size_t)2 << 30) && g.RS >= 64 ? 4 : 1;
    if (const char* e = getenv("FPX_PLACEMENT_TRIES")) tries = std::max(1, std::min(8, atoi(e)));
    char* slab = nullptr;
    // round 5: chunk placement first (place_chunks above) -- a big slab of separate arrays is built from 1 GiB physical
    // chunks paired by measurement; one hipMalloc with candidate probing (rounds 2 - 4) stays the fallback
    bool chunked = false;
    if (tries > 1 && g.VS == g.RS && stagger == 0 && !(getenv("FPX_PLACEMENT_CHUNKS") && atoi(getenv("FPX_PLACEMENT_CHUNKS")) == 0)) {
      size_t cstride = 0;
      if (place_chunks(ctx, narr, ncell * 4, g.RS / 4, &slab, &cstride)) chunked = true, stride = cstride, tries = 0;
    }
    float best_ms = 0;
    std::vector<char*> losers;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (tries > 1 && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) tries = 1;
    for (int t = 0; t < tries; ++t) {
      if (t > 0) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < stride * narr + ((size_t)8 << 30)) break;
      }
      char* cand = nullptr;
      if (hipMalloc((void**)&cand, stride * narr) != hipSuccess) {
        (void)hipGetLastError();
        break;
      }
      float ms = 0;
      if (tries > 1) {
        // 2^20 rows in 16 runs spread evenly over the whole slab: placement quality varies WITHIN an allocation too
        const int pieces = g.S >= (1 << 22) ? 16 : 1;
        const int rows = (int)std::min<int64_t>(g.S, 1 << 20) / pieces * pieces, q4 = g.RS / 4;
        const long long piece_stride = g.S / pieces;
        const int grid = (rows + 127) / 128;
        int32_t* a = g.per_slot ? (int32_t*)(cand + 2 * stride) : nullptr;
        float tmin = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
          (void)hipEventRecord(e0, ctx->stream);
          hipLaunchKernelGGL(k_probe, dim3(grid), dim3(256), 0, ctx->stream, a, (int32_t*)cand, (int32_t*)(cand + stride), rows, q4, pieces, piece_stride);
          (void)hipEventRecord(e1, ctx->stream);
          (void)hipEventSynchronize(e1);
          float m = 0;
          if (hipEventElapsedTime(&m, e0, e1) == hipSuccess && rep > 0) tmin = std::min(tmin, m);
        }
        ms = tmin;
        if (getenv("FPX_DEBUG")) fprintf(stderr, "libfpx: slab placement %d at %p: probe %.4f ms\n", t, (void*)cand, ms);
      }
      if (!slab || ms < best_ms) {
        if (slab) losers.push_back(slab);
        slab = cand, best_ms = ms;
      } else {
        losers.push_back(cand);
      }
    }
    for (char* l : losers) (void)hipFree(l);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (!slab) return fail(FPX_ENOMEM);
    if (!chunked) ctx->placement[0] = 0.f, ctx->placement[1] = 1.f, ctx->placement[2] = ctx->placement[3] = ctx->placement[4] = best_ms;
    ctx->bytes += (int64_t)(stride * narr);
    ctx->slab = slab;
    st.vote_round = (int32_t*)slab;
    st.vote_value = g.VS != g.RS ? st.vote_round + g.RS : (int32_t*)(slab + stride);  // interleaved: the slab's first two
                                                                                      // arrays are one [S][2][RS] array
    st.ballot = g.per_slot ? (int32_t*)(slab + 2 * stride) : nullptr;
  }
  if ((rc = dalloc(ctx, &st.pl_key, (size_t)g.S * g.wp))) return fail(rc);
  if ((rc = dalloc(ctx, &st.pl_value, (size_t)g.S * g.wp))) return fail(rc);
  if ((rc = dalloc(ctx, &st.pl_bits, (size_t)g.S * g.wp * 4))) return fail(rc);
  if ((rc = dalloc(ctx, &st.stamp, (size_t)g.S))) return fail(rc);
  if ((rc = dalloc(ctx, &st.row_voted, (size_t)g.S))) return fail(rc);
  if ((rc = dalloc(ctx, &st.lz_round, nsc + 4))) return fail(rc);
  if ((rc = dalloc(ctx, &st.lz_from, nsc + 4))) return fail(rc);
  if ((rc = dalloc(ctx, &st.max_ballot, nsc + 4))) return fail(rc);
  if ((rc = dalloc(ctx, &st.p1, (size_t)4 * g.R + 64))) return fail(rc);
  if ((rc = dalloc(ctx, &st.run_round, (size_t)g.ngroups))) return fail(rc);
  if ((rc = dalloc(ctx, &st.status, (size_t)8))) return fail(rc);
  ctx->g.part_rows = ctx->max_grid;
  // (two halves of the partial rows and their stamps, three thirds of the whole-group shards: finalize_body / k_phase2_fin)
  if ((rc = dalloc(ctx, &st.part, (size_t)2 * ctx->max_grid * 2 * ntab))) return fail(rc);
  if ((rc = dalloc(ctx, &st.part_stamp, (size_t)2 * ctx->max_grid))) return fail(rc);
  if ((rc = dalloc(ctx, &st.part_all, (size_t)3 * 64 * PART_ALL_STRIDE))) return fail(rc);
  if ((rc = dalloc(ctx, &st.log_value, (size_t)g.S))) return fail(rc);
  if ((rc = dalloc(ctx, &st.log_present, (size_t)g.S))) return fail(rc);
  if ((rc = dalloc(ctx, &st.log_scalars, (size_t)LG_PARTS_AT + 2 * LG_MAX_PARTS))) return fail(rc);
  if (!g.per_slot) {
    // noop-range tallies: at most cap / 2 live entries; sized for a few hundred ranges in flight per leader group,
    // bounded to 64 MiB of vote bitmaps per buffer (A x 32 B per entry)
    int64_t want = std::max<int64_t>(4096, (int64_t)g.num_leader_groups * 64);
    const int64_t by_mem = std::max<int64_t>(1024, ((int64_t)64 << 20) / ((int64_t)g.num_groups * 32));
    want = std::min<int64_t>(std::min<int64_t>(want, by_mem), 1 << 18);
    int cap = 1024;
    while (cap < want) cap <<= 1;
    for (int k = 0; k < 2; ++k) {
      RangeTable& t = ctx->rt[k];
      t.cap = cap;
      if ((rc = dalloc(ctx, &t.key, (size_t)cap * 2))) return fail(rc);
      if ((rc = dalloc(ctx, &t.bits, (size_t)cap * g.num_groups * 4))) return fail(rc);
      if ((rc = dalloc(ctx, &t.owner, (size_t)cap))) return fail(rc);
      if ((rc = dalloc(ctx, &t.count, (size_t)1))) return fail(rc);
    }
  }
  if ((rc = init_state(ctx))) return fail(rc);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(FPX_EHIP);
  *out = ctx;
  return FPX_OK;
}

int32_t fpx_destroy(fpx_ctx* ctx) {
  DeviceGuard _dg(ctx);
  if (!ctx) return FPX_EINVAL;
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->comm) (void)fpx_comm_destroy(ctx);
  free_state(ctx);
  delete ctx;
  return FPX_OK;
}

int32_t fpx_reset(fpx_ctx* ctx) {
  DeviceGuard _dg(ctx);
  if (!ctx) return FPX_EINVAL;
  return init_state(ctx);
}

int32_t fpx_set_stream(fpx_ctx* ctx, void* hip_stream) {
  DeviceGuard _dg(ctx);
  if (!ctx) return FPX_EINVAL;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream == FPX_STREAM_OWN ? ctx->own_stream : (hipStream_t)hip_stream;
  return FPX_OK;
}

int32_t fpx_sync(fpx_ctx* ctx) {
  DeviceGuard _dg(ctx);
  if (!ctx) return FPX_EINVAL;
  return fetch_status(ctx);
}

int32_t fpx_error_detail(fpx_ctx* ctx, int32_t* index, int32_t* slot, int32_t* round) {
  DeviceGuard _dg(ctx);
  if (!ctx) return FPX_EINVAL;
  if (index) *index = ctx->err_index;
  if (slot) *slot = ctx->err_slot;
  if (round) *round = ctx->err_round;
  return FPX_OK;
}

int32_t fpx_profile_enable(fpx_ctx* ctx, int32_t on) {
  DeviceGuard _dg(ctx);
  if (!ctx) return FPX_EINVAL;
  if (on && ctx->ev.empty()) {
    ctx->ev.resize(2 * 1024);
    for (auto& e : ctx->ev) HIPCHK(ctx, hipEventCreate(&e));
    ctx->cev.resize(2 * 1024);
    for (auto& e : ctx->cev) HIPCHK(ctx, hipEventCreate(&e));
  }
  ctx->profiling = on != 0;
  ctx->ev_used = 0;
  ctx->cev_used = 0;
  return FPX_OK;
}

int32_t fpx_profile_read(fpx_ctx* ctx, int32_t* launches, double* total_ms) {
  DeviceGuard _dg(ctx);
  if (!ctx) return FPX_EINVAL;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  double sum = 0;
  for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]));
    sum += ms;
  }
  if (launches) *launches = (int32_t)(ctx->ev_used / 2);
  if (total_ms) *total_ms = sum;
  ctx->ev_used = 0;
  return FPX_OK;
}

int32_t fpx_profile_read_launches(fpx_ctx* ctx, int32_t cap, float* ms_out, int32_t* launches) {
  DeviceGuard _dg(ctx);
  if (!ctx || cap < 0 || (cap > 0 && !ms_out)) return FPX_EINVAL;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  int32_t k = 0;
  for (size_t i = 0; i + 1 < ctx->ev_used; i += 2, ++k) {
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]));
    if (k < cap) ms_out[k] = ms;
  }
  if (launches) *launches = k;
  ctx->ev_used = 0;
  return FPX_OK;
}

int32_t fpx_last_hip_error(fpx_ctx* ctx) {
  DeviceGuard _dg(ctx); return ctx ? ctx->last_hip : 0; }
int32_t fpx_get_config(fpx_ctx* ctx, fpx_config* out) {
  if (!ctx || !out) return FPX_EINVAL;
  *out = ctx->cfg;
  return FPX_OK;
}

int64_t fpx_device_bytes(fpx_ctx* ctx) {
  DeviceGuard _dg(ctx); return ctx ? ctx->bytes : 0; }

int32_t fpx_placement_stats(fpx_ctx* ctx, float out[5]) {
  if (!ctx || !out) return FPX_EINVAL;
  for (int i = 0; i < 5; ++i) out[i] = ctx->placement[i];
  return FPX_OK;
}

int64_t fpx_deferred_folds(fpx_ctx* ctx) { return ctx ? ctx->fins_carried : 0; }

int32_t fpx_placement_search(fpx_ctx* ctx, int32_t* probes, int32_t* unprobed, float* ms) {
  if (!ctx) return FPX_EINVAL;
  if (probes) *probes = ctx->placement_probes;
  if (unprobed) *unprobed = ctx->placement_unprobed;
  if (ms) *ms = ctx->placement_ms;
  return FPX_OK;
}

int32_t fpx_host_alloc(int64_t bytes, void** out) {
  if (!out || bytes <= 0) return FPX_EINVAL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FPX_ENODEVICE;
  const hipError_t e = hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault);
  if (e != hipSuccess) {
    *out = nullptr;
    return e == hipErrorOutOfMemory ? FPX_ENOMEM : FPX_EHIP;
  }
  return FPX_OK;
}

int32_t fpx_host_free(void* p) {
  if (!p) return FPX_EINVAL;
  return hipHostFree(p) == hipSuccess ? FPX_OK : FPX_EHIP;
}

// ---- a5 --------------------------------------------------------------------------------------------
static int32_t quorum_eval_impl(const fpx_config* cfg, int32_t n, const uint64_t* nodes, int32_t strict, int read,
                                uint8_t* out) {
  if (!cfg || n < 0 || (n > 0 && (!nodes || !out))) return FPX_EINVAL;
  fpx_config c = *cfg;
  // only the quorum fields matter here; make the rest valid
  if (c.num_slots < 1) c.num_slots = 1;
  if (c.num_groups < 1) c.num_groups = 1;
  if (c.num_leader_groups < 1) c.num_leader_groups = 1;
  if (c.num_leaders < 1) c.num_leaders = 1;
  if (c.tally_ways < 1) c.tally_ways = 1;
  int rc = check_config(&c);
  if (rc) return rc;
  if (n == 0) return FPX_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FPX_ENODEVICE;
  DeviceGuard _dg(c.device >= 0 && c.device < ndev ? c.device : 0);
  Geom g;
  make_geom(c, &g);
  uint64_t* d_nodes = nullptr;
  uint8_t* d_out = nullptr;
  int32_t* d_status = nullptr;
  fpx_ctx* none = nullptr;
  HIPCHK(none, hipMalloc((void**)&d_nodes, (size_t)n * 32));
  HIPCHK(none, hipMalloc((void**)&d_out, (size_t)n));
  HIPCHK(none, hipMalloc((void**)&d_status, 32));
  HIPCHK(none, hipMemset(d_status, 0, 32));
  HIPCHK(none, hipMemcpy(d_nodes, nodes, (size_t)n * 32, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_quorum_eval, dim3((n + 255) / 256), dim3(256), 0, 0, g, n, d_nodes, strict, read, d_out, d_status);
  HIPCHK(none, hipGetLastError());
  int32_t hs[2] = {0, 0};
  HIPCHK(none, hipMemcpy(out, d_out, (size_t)n, hipMemcpyDeviceToHost));
  HIPCHK(none, hipMemcpy(hs, d_status, sizeof(hs), hipMemcpyDeviceToHost));
  (void)hipFree(d_nodes);
  (void)hipFree(d_out);
  (void)hipFree(d_status);
  return hs[0];
}

int32_t fpx_quorum_eval(const fpx_config* cfg, int32_t n, const uint64_t* nodes, int32_t strict, uint8_t* out) {
  return quorum_eval_impl(cfg, n, nodes, strict, 0, out);
}
int32_t fpx_read_quorum_eval(const fpx_config* cfg, int32_t n, const uint64_t* nodes, int32_t strict, uint8_t* out) {
  return quorum_eval_impl(cfg, n, nodes, strict, 1, out);
}
int32_t fpx_is_write_quorum(const fpx_config* cfg, const uint64_t nodes[4], int32_t strict, uint8_t* out) {
  return quorum_eval_impl(cfg, 1, nodes, strict, 0, out);
}

// ---- device-pointer entry points -----------------------------------------------------------------
int32_t fpx_acceptor_phase2a_dev(fpx_ctx* ctx, int32_t n, const int32_t* d_slot, const int32_t* d_round,
                                 const int32_t* d_value_id, const uint64_t* d_target_mask, uint64_t* d_vote_bits,
                                 uint64_t* d_nack_bits, int32_t* d_nack_round) {
  DeviceGuard _dg(ctx);
  if (!ctx || n < 0) return FPX_EINVAL;
  Batch b;
  memset(&b, 0, sizeof(b));
  b.n = n, b.slot = d_slot, b.round = d_round, b.value = d_value_id, b.target = d_target_mask;
  b.vote_bits = d_vote_bits, b.nack_bits = d_nack_bits, b.nack_round = d_nack_round;
  return enqueue_phase2(ctx, b, false);
}

int32_t fpx_wire_decode_proxy_leader_inbound_dev(fpx_ctx* ctx, const uint8_t* d_buf, int64_t buf_len,
                                                 const int64_t* d_offsets, int32_t n, int32_t* d_kind, int32_t* d_slot,
                                                 int32_t* d_round, int32_t* d_is_noop, int64_t* d_value_off,
                                                 int32_t* d_value_len, int32_t* d_group_index, int32_t* d_acceptor_index,
                                                 int32_t value_id_base, int32_t* d_value_id) {
  WireOut o{d_kind, d_slot, d_round, d_is_noop, d_value_len, d_group_index, d_acceptor_index, d_value_id, d_value_off,
            value_id_base};
  return wire_decode_dev<0>(ctx, d_buf, buf_len, d_offsets, n, o);
}

int32_t fpx_wire_decode_acceptor_inbound_dev(fpx_ctx* ctx, const uint8_t* d_buf, int64_t buf_len,
                                             const int64_t* d_offsets, int32_t n, int32_t* d_kind, int32_t* d_slot,
                                             int32_t* d_round, int32_t* d_is_noop, int64_t* d_value_off,
                                             int32_t* d_value_len, int32_t* d_chosen_watermark, int32_t value_id_base,
                                             int32_t* d_value_id) {
  WireOut o{d_kind, d_slot, d_round, d_is_noop, d_value_len, d_chosen_watermark, nullptr, d_value_id, d_value_off,
            value_id_base};
  return wire_decode_dev<1>(ctx, d_buf, buf_len, d_offsets, n, o);
}

int32_t fpx_phase2_fused_dev(fpx_ctx* ctx, int32_t n, const int32_t* d_slot, const int32_t* d_round,
                             const int32_t* d_value_id, const uint64_t* d_target_mask, uint8_t* d_chosen,
                             int32_t* d_chosen_round, int32_t* d_chosen_value, int32_t* d_nack_round) {
  if (!ctx || n < 0) return FPX_EINVAL;
  DeviceGuard _dg(ctx->cfg.device);  // (by device number: a pending fold of the launch before is this call's to carry)
  Batch b;
  memset(&b, 0, sizeof(b));
  b.n = n, b.slot = d_slot, b.round = d_round, b.value = d_value_id, b.target = d_target_mask;
  b.chosen = d_chosen, b.chosen_round = d_chosen_round, b.chosen_value = d_chosen_value, b.nack_round = d_nack_round;
  return enqueue_phase2(ctx, b, true);
}

int32_t fpx_proxy_open_dev(fpx_ctx* ctx, int32_t n, const int32_t* d_slot, const int32_t* d_round,
                           const int32_t* d_value_id, uint8_t* d_is_new) {
  DeviceGuard _dg(ctx);
  if (!ctx || n < 0) return FPX_EINVAL;
  Batch b;
  memset(&b, 0, sizeof(b));
  b.n = n, b.slot = d_slot, b.round = d_round, b.value = d_value_id, b.is_new = d_is_new;
  return enqueue_open(ctx, b);
}

int32_t fpx_proxy_phase2b_dev(fpx_ctx* ctx, int32_t n, const int32_t* d_slot, const int32_t* d_round,
                              const uint64_t* d_vote_bits, uint8_t* d_newly_chosen, int32_t* d_chosen_round,
                              int32_t* d_chosen_value) {
  DeviceGuard _dg(ctx);
  if (!ctx || n < 0) return FPX_EINVAL;
  Batch b;
  memset(&b, 0, sizeof(b));
  b.n = n, b.slot = d_slot, b.round = d_round, b.vote_bits = const_cast<uint64_t*>(d_vote_bits);
  b.chosen = d_newly_chosen, b.chosen_round = d_chosen_round, b.chosen_value = d_chosen_value;
  return enqueue_tally(ctx, b);
}

// ---- host-pointer entry points: stage, split into runs, run, copy back ----------------------------
int32_t fpx_acceptor_phase2a(fpx_ctx* ctx, int32_t n, const int32_t* slot, const int32_t* round,
                             const int32_t* value_id, const uint64_t* target_mask, uint64_t* vote_bits,
                             uint64_t* nack_bits, int32_t* nack_round) {
  DeviceGuard _dg(ctx);
  int rc = check_args(ctx, n, slot, round);
  if (rc) return rc;
  if (n == 0) return FPX_OK;
  if (!value_id) return FPX_EINVAL;
  if ((rc = h2d(ctx, &ctx->d_slot, slot, n))) return rc;
  if ((rc = h2d(ctx, &ctx->d_round, round, n))) return rc;
  if ((rc = h2d(ctx, &ctx->d_value, value_id, n))) return rc;
  if (target_mask && (rc = h2d(ctx, &ctx->d_target, target_mask, (size_t)n * 4))) return rc;
  if ((rc = grow(ctx, &ctx->d_bits_a, (size_t)n * 32))) return rc;
  if ((rc = grow(ctx, &ctx->d_bits_b, (size_t)n * 32))) return rc;
  if ((rc = grow(ctx, &ctx->d_i32_a, (size_t)n * 4))) return rc;
  if ((rc = check_inputs(ctx, n, slot, round))) return rc;
  HostRun host_run(ctx);
  std::vector<int> cuts;
  split_runs(ctx, n, slot, round, true, &cuts);
  for (size_t k = 0; k + 1 < cuts.size(); ++k) {
    const int lo = cuts[k], len = cuts[k + 1] - cuts[k];
    rc = fpx_acceptor_phase2a_dev(ctx, len, (int32_t*)ctx->d_slot.p + lo, (int32_t*)ctx->d_round.p + lo,
                                  (int32_t*)ctx->d_value.p + lo,
                                  target_mask ? (uint64_t*)ctx->d_target.p + (size_t)lo * 4 : nullptr,
                                  (uint64_t*)ctx->d_bits_a.p + (size_t)lo * 4,
                                  nack_bits ? (uint64_t*)ctx->d_bits_b.p + (size_t)lo * 4 : nullptr,
                                  (int32_t*)ctx->d_i32_a.p + lo);
    if (rc) return rc;
  }
  if ((rc = d2h(ctx, vote_bits, ctx->d_bits_a, (size_t)n * 4))) return rc;
  if ((rc = d2h(ctx, nack_bits, ctx->d_bits_b, (size_t)n * 4))) return rc;
  if ((rc = d2h(ctx, nack_round, ctx->d_i32_a, (size_t)n))) return rc;
  return fetch_status(ctx);
}

// Host-pointer K3.  Optimistic: the uploads are followed at once by the whole batch as ONE validated
// device run (k_validate checks ranges and the run contract on the GPU), so a well-formed batch -- the
// common case -- costs no CPU pass over the messages.  Only when the device reports a contract violation
// (nothing was applied) does the host cut the batch into runs itself and replay them.
int32_t fpx_phase2_fused(fpx_ctx* ctx, int32_t n, const int32_t* slot, const int32_t* round, const int32_t* value_id,
                         const uint64_t* target_mask, uint8_t* chosen, int32_t* chosen_round, int32_t* chosen_value,
                         int32_t* nack_round) {
  DeviceGuard _dg(ctx);
  int rc = check_args(ctx, n, slot, round);
  if (rc) return rc;
  if (n == 0) return FPX_OK;
  if (!value_id) return FPX_EINVAL;
  if ((rc = grow(ctx, &ctx->d_u8, (size_t)n))) return rc;
  if ((rc = grow(ctx, &ctx->d_i32_a, (size_t)n * 4))) return rc;
  if ((rc = grow(ctx, &ctx->d_i32_b, (size_t)n * 4))) return rc;
  if ((rc = grow(ctx, &ctx->d_i32_c, (size_t)n * 4))) return rc;
  if (n >= 4096) {  // page-locked arrays: staged by kernels, pipelined with the fused step
    bool used = false;
    rc = host_fused_staged(ctx, n, slot, round, value_id, target_mask, chosen, chosen_round, chosen_value, nack_round, &used);
    if (used) return rc;
  }
  if (n >= 2 * host_piece(ctx, n))
    return host_fused_pipelined(ctx, n, slot, round, value_id, target_mask, chosen, chosen_round, chosen_value, nack_round);
  if ((rc = h2d(ctx, &ctx->d_slot, slot, n))) return rc;
  if ((rc = h2d(ctx, &ctx->d_round, round, n))) return rc;
  if ((rc = h2d(ctx, &ctx->d_value, value_id, n))) return rc;
  if (target_mask && (rc = h2d(ctx, &ctx->d_target, target_mask, (size_t)n * 4))) return rc;
  int32_t *d_slot = (int32_t*)ctx->d_slot.p, *d_round = (int32_t*)ctx->d_round.p, *d_value = (int32_t*)ctx->d_value.p;
  uint64_t* d_target = target_mask ? (uint64_t*)ctx->d_target.p : nullptr;
  uint8_t* d_ch = (uint8_t*)ctx->d_u8.p;
  int32_t *d_cr = (int32_t*)ctx->d_i32_a.p, *d_cv = (int32_t*)ctx->d_i32_b.p, *d_nr = (int32_t*)ctx->d_i32_c.p;
  auto download = [&]() -> int {
    int r2;
    if ((r2 = d2h(ctx, chosen, ctx->d_u8, (size_t)n))) return r2;
    if ((r2 = d2h(ctx, chosen_round, ctx->d_i32_a, (size_t)n))) return r2;
    if ((r2 = d2h(ctx, chosen_value, ctx->d_i32_b, (size_t)n))) return r2;
    return d2h(ctx, nack_round, ctx->d_i32_c, (size_t)n);
  };
  int replay_from = 0;  // the host-split replay below covers [replay_from, n)
  {
    ctx->force_validate = true;  // also under FPX_F_TRUSTED: that flag is a promise about _dev batches only
    rc = fpx_phase2_fused_dev(ctx, n, d_slot, d_round, d_value, d_target, d_ch, d_cr, d_cv, d_nr);
    ctx->force_validate = false;
    if (rc == FPX_OK) rc = download();
    if (rc) {
      (void)hipStreamSynchronize(ctx->stream);
      return rc;
    }
    rc = fetch_status(ctx);
    if (rc == FPX_EINVAL) return check_inputs(ctx, n, slot, round);  // the FIRST offender, for fpx_error_detail
    if (rc != FPX_EORDER) return rc;
  }
  return host_fused_replay(ctx, n, slot, round, target_mask != nullptr, replay_from, chosen, chosen_round, chosen_value,
                           nack_round);
}

int32_t fpx_phase2_fused_submit(fpx_ctx* ctx, int32_t n, const int32_t* slot, const int32_t* round, const int32_t* value_id,
                                const uint64_t* target_mask, uint8_t* chosen, int32_t* chosen_round, int32_t* chosen_value,
                                int32_t* nack_round, int32_t* ticket) {
  if (!ctx) return FPX_EINVAL;
  DeviceGuard _dg(ctx->cfg.device);  // (by device number, like fpx_phase2_fused_dev: the fold of the call before rides in this one's)
  int rc = check_args(ctx, n, slot, round);
  if (rc) return rc;
  if (n == 0 || !value_id || !ticket) return FPX_EINVAL;
  int t = -1;
  rc = host_submit(ctx, n, slot, round, value_id, target_mask, chosen, chosen_round, chosen_value, nack_round, &t);
  *ticket = t;
  return rc;
}

int32_t fpx_phase2_fused_wait(fpx_ctx* ctx, int32_t ticket) {
  if (!ctx) return FPX_EINVAL;
  DeviceGuard _dg(ctx->cfg.device);  // (waits for an event and reads the call's status words: no state is touched)
  return host_wait(ctx, ticket);
}

int32_t fpx_proxy_open(fpx_ctx* ctx, int32_t n, const int32_t* slot, const int32_t* round, const int32_t* value_id,
                       uint8_t* is_new) {
  DeviceGuard _dg(ctx);
  int rc = check_args(ctx, n, slot, round);
  if (rc) return rc;
  if (n == 0) return FPX_OK;
  if (!value_id) return FPX_EINVAL;
  if ((rc = h2d(ctx, &ctx->d_slot, slot, n))) return rc;
  if ((rc = h2d(ctx, &ctx->d_round, round, n))) return rc;
  if ((rc = h2d(ctx, &ctx->d_value, value_id, n))) return rc;
  if ((rc = grow(ctx, &ctx->d_u8, (size_t)n))) return rc;
  if ((rc = check_inputs(ctx, n, slot, round))) return rc;
  HostRun host_run(ctx);
  std::vector<int> cuts;
  split_runs(ctx, n, slot, round, false, &cuts);
  for (size_t k = 0; k + 1 < cuts.size(); ++k) {
    const int lo = cuts[k], len = cuts[k + 1] - cuts[k];
    rc = fpx_proxy_open_dev(ctx, len, (int32_t*)ctx->d_slot.p + lo, (int32_t*)ctx->d_round.p + lo,
                            (int32_t*)ctx->d_value.p + lo, (uint8_t*)ctx->d_u8.p + lo);
    if (rc) return rc;
  }
  if ((rc = d2h(ctx, is_new, ctx->d_u8, (size_t)n))) return rc;
  return fetch_status(ctx);
}

int32_t fpx_proxy_phase2b(fpx_ctx* ctx, int32_t n, const int32_t* slot, const int32_t* round, const uint64_t* vote_bits,
                          uint8_t* newly_chosen, int32_t* chosen_round, int32_t* chosen_value) {
  DeviceGuard _dg(ctx);
  int rc = check_args(ctx, n, slot, round);
  if (rc) return rc;
  if (n == 0) return FPX_OK;
  if (!vote_bits) return FPX_EINVAL;
  if ((rc = h2d(ctx, &ctx->d_slot, slot, n))) return rc;
  if ((rc = h2d(ctx, &ctx->d_round, round, n))) return rc;
  if ((rc = h2d(ctx, &ctx->d_bits_a, vote_bits, (size_t)n * 4))) return rc;
  if ((rc = grow(ctx, &ctx->d_u8, (size_t)n))) return rc;
  if ((rc = grow(ctx, &ctx->d_i32_a, (size_t)n * 4))) return rc;
  if ((rc = grow(ctx, &ctx->d_i32_b, (size_t)n * 4))) return rc;
  if ((rc = check_inputs(ctx, n, slot, round))) return rc;
  HostRun host_run(ctx);
  std::vector<int> cuts;
  split_runs(ctx, n, slot, round, false, &cuts);
  for (size_t k = 0; k + 1 < cuts.size(); ++k) {
    const int lo = cuts[k], len = cuts[k + 1] - cuts[k];
    rc = fpx_proxy_phase2b_dev(ctx, len, (int32_t*)ctx->d_slot.p + lo, (int32_t*)ctx->d_round.p + lo,
                               (uint64_t*)ctx->d_bits_a.p + (size_t)lo * 4, (uint8_t*)ctx->d_u8.p + lo,
                               (int32_t*)ctx->d_i32_a.p + lo, (int32_t*)ctx->d_i32_b.p + lo);
    if (rc) return rc;
  }
  if ((rc = d2h(ctx, newly_chosen, ctx->d_u8, (size_t)n))) return rc;
  if ((rc = d2h(ctx, chosen_round, ctx->d_i32_a, (size_t)n))) return rc;
  if ((rc = d2h(ctx, chosen_value, ctx->d_i32_b, (size_t)n))) return rc;
  return fetch_status(ctx);
}

// Phase1a on device-resident arguments, asynchronous: d_target_mask 4 words or NULL, d_outp / d_outn 4 words each = promised
// bits, nack bits (both written in full).  With a ballot per cell: ONE launch (k_p1a_fast) that leaves a pending fold pending
// -- the callers enter by device number, like fpx_phase2_fused_dev; FPX_P1A_SPLIT=1 is the order of rounds 2 - 5 (fold,
// k_p1a_decide, k_p1a_sweep as launches of their own).
static int enqueue_phase1a(fpx_ctx* ctx, int group, int round, int watermark, const uint64_t* d_tgt, uint64_t* d_outp,
                           uint64_t* d_outn) {
  int rc;
  const int wm = watermark < 0 ? 0 : watermark;
  const dim3 gr(1), blk(256);  // R <= 256: one block, which also initialises the reply bits
  if (!ctx->g.per_slot) {
    flush_pending_fin(ctx);    // (nothing is ever pending with a round per acceptor)
    hipLaunchKernelGGL(k_phase1a_scalar, gr, blk, 0, ctx->stream, ctx->g, ctx->st, group, round, d_tgt, d_outp, d_outn);
  } else if (wm > ctx->lz_min_from || getenv("FPX_P1A_SPLIT")) {
    // the watermark passes the start of an older lazy promise, maybe (the host knows every watermark it handed in): the
    // cells in between keep the older promise, which k_p1a_sweep first writes into them
    flush_pending_fin(ctx);
    // O(R) unless the Phase1a is stale for some acceptor or an older lazy promise has to be made explicit below
    // the new watermark: k_p1a_sweep returns at once in the common case; when it does sweep, its last block writes
    // the swept acceptors' promises (those none of whose cells was ahead)
    hipLaunchKernelGGL(k_p1a_decide, gr, blk, 0, ctx->stream, ctx->g, ctx->st, group, round, watermark, d_tgt, d_outp, d_outn);
    hipLaunchKernelGGL(k_p1a_sweep, dim3(ctx->num_cus * 2), dim3(256), 0, ctx->stream, ctx->g, ctx->st, group, round,
                       watermark, d_outp, d_outn);
    ctx->lz_min_from = std::min(ctx->lz_min_from, wm);
    ctx->lazy_active = true;
  } else {
    // (the pending fold stays pending: the decision takes the launch's bound from its shards of part_all)
    hipLaunchKernelGGL(k_p1a_fast, dim3(ctx->num_cus * 2), blk, 0, ctx->stream, ctx->g, ctx->st,
                       ctx->pending_fin.nblk ? ctx->pending_fin.par : -1, group, round, watermark, d_tgt, d_outp, d_outn);
    ctx->lz_min_from = std::min(ctx->lz_min_from, wm);
    ctx->lazy_active = true;
  }
  if ((rc = launch_check(ctx))) return rc;
  return FPX_OK;
}

int32_t fpx_acceptor_phase1a_dev(fpx_ctx* ctx, int32_t group, int32_t round, int32_t chosen_watermark,
                                 const uint64_t* d_target_mask, uint64_t* d_promised_bits, uint64_t* d_nack_bits) {
  if (!ctx || group < 0 || group >= ctx->g.ngroups || round < 0 || round > MAX_ROUND) return FPX_EINVAL;
  DeviceGuard _dg(ctx->cfg.device);  // (a pending fold stays pending: enqueue_phase1a)
  int rc;
  if ((rc = grow(ctx, &ctx->d_scratch, 128))) return rc;
  uint64_t* d_out = (uint64_t*)ctx->d_scratch.p;  // [8]: promised bits, nack bits the caller does not want
  return enqueue_phase1a(ctx, group, round, chosen_watermark, d_target_mask, d_promised_bits ? d_promised_bits : d_out,
                         d_nack_bits ? d_nack_bits : d_out + 4);
}

int32_t fpx_acceptor_phase1a(fpx_ctx* ctx, int32_t group, int32_t round, int32_t chosen_watermark,
                             const uint64_t* target_mask, uint64_t* promised_bits, uint64_t* nack_bits) {
  if (!ctx || group < 0 || group >= ctx->g.ngroups || round < 0 || round > MAX_ROUND) return FPX_EINVAL;
  DeviceGuard _dg(ctx->cfg.device);
  int rc;
  if ((rc = grow(ctx, &ctx->d_scratch, 128))) return rc;
  uint64_t* d_out = (uint64_t*)ctx->d_scratch.p;       // [8]: promised bits, nack bits
  uint64_t* d_tgt = target_mask ? d_out + 8 : nullptr;  // [4]
  if (target_mask) HIPCHK(ctx, hipMemcpyAsync(d_tgt, target_mask, 32, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = enqueue_phase1a(ctx, group, round, chosen_watermark, d_tgt, d_out, d_out + 4))) return rc;
  uint64_t h[8];
  HIPCHK(ctx, hipMemcpyAsync(h, d_out, 64, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (promised_bits) memcpy(promised_bits, h, 32);
  if (nack_bits) memcpy(nack_bits, h + 4, 32);
  return FPX_OK;
}

// PER_SLOT: every outstanding lazy promise is written into the cells it covers (one sweep over the ballot array)
// and forgotten; k_phase2 goes back to its lean form.  Readback and digests do this implicitly.
static int flush_promises(fpx_ctx* ctx) {
  if (!ctx->g.per_slot || !ctx->lazy_active) return FPX_OK;
  hipLaunchKernelGGL(k_lazy_flush, dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, ctx->g, ctx->st);
  const int nsc = ctx->g.ngroups * ctx->g.R;
  hipLaunchKernelGGL(k_lazy_clear, dim3((nsc + 255) / 256), dim3(256), 0, ctx->stream, ctx->g, ctx->st);
  int rc = launch_check(ctx);
  if (rc) return rc;
  ctx->lazy_active = false, ctx->lz_min_from = 0x7fffffff;
  return FPX_OK;
}

int32_t fpx_acceptor_flush_promises(fpx_ctx* ctx) {
  DeviceGuard _dg(ctx);
  if (!ctx) return FPX_EINVAL;
  return flush_promises(ctx);
}

int32_t fpx_proxy_forget(fpx_ctx* ctx, int32_t first_slot, int32_t count) {
  DeviceGuard _dg(ctx);
  if (!ctx || first_slot < 0 || count < 0 || (int64_t)first_slot + count > ctx->g.S) return FPX_EINVAL;
  if (count == 0) return FPX_OK;
  // an empty key word is all a tally entry needs to be free again (values / bitmaps are rewritten on open)
  if (ctx->g.lg_rows) {  // the slots of the range are not neighbours in memory: one thread per slot
    hipLaunchKernelGGL(k_clear_slots, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, ctx->g, ctx->st, first_slot, count, 0);
  } else {
    HIPCHK(ctx, hipMemsetAsync(ctx->st.pl_key + (size_t)first_slot * ctx->g.wp, 0, (size_t)count * ctx->g.wp * 4,
                               ctx->stream));
  }
  if (ctx->rt[0].key) {
    // the noop-range tallies that lie inside the window go too: the survivors move to the other table buffer
    const RangeTable& from = ctx->rt[ctx->rt_cur];
    const RangeTable& to = ctx->rt[ctx->rt_cur ^ 1];
    int rc = clear_range_table(ctx, to);
    if (rc) return rc;
    hipLaunchKernelGGL(k_ranges_rehash, dim3((from.cap + 255) / 256), dim3(256), 0, ctx->stream, from, to,
                       ctx->g.num_groups * 4, first_slot, count);
    if ((rc = launch_check(ctx))) return rc;
    ctx->rt_cur ^= 1;
  }
  return FPX_OK;
}

int32_t fpx_recycle_slots(fpx_ctx* ctx, int32_t first_slot, int32_t count) {
  DeviceGuard _dg(ctx);
  if (!ctx || first_slot < 0 || count < 0 || (int64_t)first_slot + count > ctx->g.S) return FPX_EINVAL;
  if (count == 0) return FPX_OK;
  if (ctx->g.lg_rows) {
    hipLaunchKernelGGL(k_clear_slots, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, ctx->g, ctx->st, first_slot, count, 1);
  } else {
    const size_t row = (size_t)ctx->g.VS * 4, at = (size_t)first_slot * row, len = (size_t)count * row;
    HIPCHK(ctx, hipMemsetAsync((char*)ctx->st.vote_round + at, 0xFF, len, ctx->stream));  // -1: no vote
    if (ctx->g.VS == ctx->g.RS) HIPCHK(ctx, hipMemsetAsync((char*)ctx->st.vote_value + at, 0xFF, len, ctx->stream));
    if (ctx->st.row_voted) HIPCHK(ctx, hipMemsetAsync(ctx->st.row_voted + first_slot, 0, (size_t)count, ctx->stream));
  }

  return fpx_proxy_forget(ctx, first_slot, count);
}

// ---- K4: Mencius noop ranges (batched; kernels in fpx_ranges.hpp) -----------------------------------------
static int32_t ranges_ctx_ok(fpx_ctx* ctx, int32_t n) {
  if (!ctx || n < 0) return FPX_EINVAL;
  if (ctx->g.per_slot) return FPX_EINVAL;  // ranges act on the acceptor's `round` scalar (mencius/Acceptor.scala:245-260)
  return FPX_OK;
}

enum { RANGES_FUSED = 0, RANGES_ACCEPTORS = 1, RANGES_OPEN = 2, RANGES_TALLY = 3 };

// one device run of n ranges; vote_bits must be present for every mode but RANGES_OPEN
static int enqueue_ranges(fpx_ctx* ctx, RangeBatch& b, int mode) {
  if (b.n == 0) return FPX_OK;
  const Geom& g = ctx->g;
  const size_t words = (size_t)b.n * g.num_groups * 4;
  b.run_id = ++ctx->run_id;
  b.fused = mode == RANGES_FUSED;
  b.quorum = ctx->cfg.f + 1;
  const int gn = (b.n + 255) / 256;
  const bool acceptors = mode == RANGES_FUSED || mode == RANGES_ACCEPTORS;
  const bool validate = !(((ctx->cfg.flags & FPX_F_TRUSTED) && !ctx->force_validate) || ctx->host_validated);
  if (validate) {
    if (acceptors) fill32(ctx, ctx->st.run_round, -1, (size_t)g.ngroups);
    hipLaunchKernelGGL(k_ranges_validate, dim3(gn), dim3(256), 0, ctx->stream, g, ctx->st, b, acceptors ? 1 : 0);
  }
  const RangeTable& rt = ctx->rt[ctx->rt_cur];
  // a few ranges: one workgroup walks clear -> open -> resolve -> acceptors -> tally (fpx_ranges.hpp: k_ranges_chain)
  const long long chain_words = ranges_chain_words(b.n, g.num_groups);
  const bool chain = mode == RANGES_FUSED && b.n <= RANGES_CHAIN_MAX && chain_words <= RANGES_CHAIN_LDS_WORDS &&
                     (long long)b.n * g.num_groups * g.R <= 8 * 1024 && !getenv("FPX_RANGES_NO_CHAIN");
  if (chain) {
    allow_lds(k_ranges_chain, (size_t)chain_words * 4 + 64);
    hipLaunchKernelGGL(k_ranges_chain, dim3(1), dim3(1024), (size_t)chain_words * 4, ctx->stream, g, ctx->st, rt, b);
  }
  if (acceptors && !chain) {
    fill32(ctx, b.vote_bits, 0, words * 2);
    if (b.nack_bits) fill32(ctx, b.nack_bits, 0, words * 2);
    if (b.nack_round) fill32(ctx, b.nack_round, -1, (size_t)b.n);
  }
  if (!chain && (mode == RANGES_FUSED || mode == RANGES_OPEN)) {
    hipLaunchKernelGGL(k_ranges_open, dim3(gn), dim3(256), 0, ctx->stream, g, ctx->st, rt, b, 0);
    hipLaunchKernelGGL(k_ranges_resolve, dim3(gn), dim3(256), 0, ctx->stream, g, ctx->st, rt, b);
  }
  if (acceptors) {
    const long long threads = (long long)b.n * g.num_groups * g.R;
    if (!chain)
      hipLaunchKernelGGL(k_ranges_acceptors, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, g, ctx->st, b);
    if (g.lg_rows) {  // leader-group-major rows: a range is one run of rows
      const int gy = std::min(b.n, 4096);
      const int gx = std::max(1, std::min(ctx->num_cus * 16 / gy, 64));
      hipLaunchKernelGGL(k_ranges_fill_lg, dim3(gx, gy), dim3(256), 0, ctx->stream, g, ctx->st, b);
    } else if (b.n <= RF_MAXN && g.num_leader_groups <= RF_MAXL && !getenv("FPX_RANGES_FILL_V1")) {
      // (RF_JB rows of L ownership words + the kernel's own 12 static bytes: beyond the default 64 KiB of dynamic LDS
      // the kernel needs the opt-in -- ADVICE r03: L close to 2048 failed every launch with FPX_EHIP)
      allow_lds(k_ranges_fill_rows, (size_t)RF_JB * g.num_leader_groups * 4 + 64);
      // sweep the log rows the ranges touch in memory order (fpx_ranges.hpp)
      hipLaunchKernelGGL(k_ranges_fill_rows, dim3(ctx->num_cus * 8), dim3(256), (size_t)RF_JB * g.num_leader_groups * 4,
                         ctx->stream, g, ctx->st, b);
    } else {
      const int gy = std::min(b.n, 4096);
      const int gx = std::max(1, std::min(ctx->num_cus * 16 / gy, 64));
      hipLaunchKernelGGL(k_ranges_fill, dim3(gx, gy), dim3(256), 0, ctx->stream, g, ctx->st, b);
    }
  }
  if (mode == RANGES_TALLY) hipLaunchKernelGGL(k_ranges_open, dim3(gn), dim3(256), 0, ctx->stream, g, ctx->st, rt, b, 1);
  if (!chain && (mode == RANGES_FUSED || mode == RANGES_TALLY))
    hipLaunchKernelGGL(k_ranges_tally, dim3(gn), dim3(256), 0, ctx->stream, g, ctx->st, rt, b);
  return launch_check(ctx);
}

int32_t fpx_noop_ranges_fused_dev(fpx_ctx* ctx, int32_t n, const int32_t* d_slot_start, const int32_t* d_slot_end,
                                  const int32_t* d_round, const uint64_t* d_target_masks, uint64_t* d_vote_bits,
                                  uint64_t* d_nack_bits, int32_t* d_nack_round, uint8_t* d_is_new, uint8_t* d_chosen) {
  DeviceGuard _dg(ctx);
  int rc = ranges_ctx_ok(ctx, n);
  if (rc) return rc;
  if (n == 0) return FPX_OK;
  const size_t words = (size_t)n * ctx->g.num_groups * 4;
  // scratch: entry[n], and the vote bitmaps when the caller does not want them
  if ((rc = grow(ctx, &ctx->d_rng, (size_t)n * 4 + 64 + (d_vote_bits ? 0 : words * 8)))) return rc;
  RangeBatch b;
  memset(&b, 0, sizeof(b));
  b.n = n, b.start = d_slot_start, b.end = d_slot_end, b.round = d_round, b.target = d_target_masks;
  b.entry = (int32_t*)ctx->d_rng.p;
  b.vote_bits = d_vote_bits ? d_vote_bits : (uint64_t*)((char*)ctx->d_rng.p + (((size_t)n * 4 + 63) & ~(size_t)63));
  b.nack_bits = d_nack_bits, b.nack_round = d_nack_round, b.is_new = d_is_new, b.chosen = d_chosen;
  return enqueue_ranges(ctx, b, RANGES_FUSED);
}

// A band whose halves are independent, in TWO launches instead of four (profiles/r05_cfg5.md): the vote kernel with the
// range chain as its first workgroup (k_phase2_band), then the fill with the vote kernel's k_finalize in further rows of
// its grid (k_ranges_fill_lg_fin).  Only for the everyday shape -- groups of at most 32 acceptors on leader-group-major
// rows, dense delivery, a chain that fits the vote kernel's own LDS, more than one workgroup of commands; *done = false
// with nothing enqueued otherwise (the caller takes the serial order).
static int enqueue_band_merged(fpx_ctx* ctx, Batch& b, RangeBatch& rb, bool* done) {
  *done = false;
  const Geom& g = ctx->g;
  const int G = ctx->lanes_per_slot;
  if (G > 8 || b.target || !g.lg_rows || g.per_slot || getenv("FPX_BAND_SERIAL")) return FPX_OK;
  const long long chain_words = ranges_chain_words(rb.n, g.num_groups);
  if (rb.n > RANGES_CHAIN_MAX || chain_words > RANGES_CHAIN_LDS_WORDS || (long long)rb.n * g.num_groups * g.R > 8 * 1024 ||
      getenv("FPX_RANGES_NO_CHAIN"))
    return FPX_OK;
  b.chunk = chunk_for(ctx, b.n);
  if ((b.n + b.chunk - 1) / b.chunk <= 8) return FPX_OK;  // one workgroup of commands: it finalises itself (`solo`)
  // (the launch's LDS is every workgroup's: the chain may raise it to what still lets six workgroups share a CU)
  const size_t lds = std::max(phase2_lds(ctx, b, true, false, true), (size_t)chain_words * 4);
  if (lds > 26000) return FPX_OK;
  // -- from here on as enqueue_phase2 and enqueue_ranges (FPX_F_TRUSTED: no validation passes) --
  b.index_base = ctx->index_base;
  int rc = enqueue_validate(ctx, b, true);
  if (rc) return rc;
  const int grid = grid_for(ctx, b.n);
  b.solo = 0;
  begin_phase2_launch(ctx, b);
  if (++ctx->launch_seq == 0) ctx->launch_seq = 1;
  b.launch_seq = ctx->launch_seq;
  rb.run_id = ++ctx->run_id;
  rb.fused = 1;
  rb.quorum = ctx->cfg.f + 1;
  const bool prof = ctx->profiling && ctx->ev_used + 2 <= ctx->ev.size();
  ctx->ev_start = prof ? ctx->ev[ctx->ev_used] : nullptr;
  ctx->ev_stop = prof ? ctx->ev[ctx->ev_used + 1] : nullptr;
  const RangeTable& rt = ctx->rt[ctx->rt_cur];
  launch_band(ctx, b, lds, grid, rt, rb);
  ctx->ev_start = ctx->ev_stop = nullptr;
  if (prof) ctx->ev_used += 2;
  if ((rc = launch_check(ctx))) return rc;
  const int ntab = g.ngroups * g.R;
  const int slices = std::max(FINALIZE_SLICES, std::min(256, grid / 32)), fgx = (ntab + 63) / 64;
  const int gy = std::min(rb.n, 4096);
  const int gx = std::max(1, std::min(ctx->num_cus * 16 / gy, 64));
  const int fin_rows = (fgx * slices + gx - 1) / gx;
  hipLaunchKernelGGL(k_ranges_fill_lg_fin, dim3(gx, gy + fin_rows), dim3(256), 0, ctx->stream, g, ctx->launch_st, rb, gy, fin_rows,
                     (int)b.parity, grid, b.launch_seq, fgx, slices);
  *done = true;
  ++ctx->band_merged_steps;
  return launch_check(ctx);
}

// One proxy-leader step of Mencius: fpx_phase2_fused_dev on the commands, then fpx_noop_ranges_fused_dev on the ranges --
// and, when the caller says that no leader group has both (`independent`), the two halves in two launches
// (enqueue_band_merged) or side by side: the ranges on the context's second stream between a fork and a join event
// (profiles/r05_cfg5.md).
int32_t fpx_mencius_band_fused_dev(fpx_ctx* ctx, int32_t n, const int32_t* d_slot, const int32_t* d_round, const int32_t* d_value_id,
                                   const uint64_t* d_target_mask, uint8_t* d_chosen, int32_t* d_chosen_round, int32_t* d_chosen_value,
                                   int32_t* d_nack_round, int32_t n_ranges, const int32_t* d_slot_start, const int32_t* d_slot_end,
                                   const int32_t* d_range_round, const uint64_t* d_range_target_masks, uint64_t* d_range_vote_bits,
                                   uint64_t* d_range_nack_bits, int32_t* d_range_nack_round, uint8_t* d_range_is_new,
                                   uint8_t* d_range_chosen, int32_t independent) {
  DeviceGuard _dg(ctx);
  if (!ctx || n < 0 || n_ranges < 0) return FPX_EINVAL;
  int rc;
  if (!independent || n == 0 || n_ranges == 0 || ctx->g.per_slot) {  // the serial order (what the two calls would do)
    if ((rc = fpx_phase2_fused_dev(ctx, n, d_slot, d_round, d_value_id, d_target_mask, d_chosen, d_chosen_round, d_chosen_value, d_nack_round)))
      return rc;
    return fpx_noop_ranges_fused_dev(ctx, n_ranges, d_slot_start, d_slot_end, d_range_round, d_range_target_masks, d_range_vote_bits,
                                     d_range_nack_bits, d_range_nack_round, d_range_is_new, d_range_chosen);
  }
  if (!ctx->band_stream) {
    if (hipStreamCreateWithFlags(&ctx->band_stream, hipStreamNonBlocking) != hipSuccess) return FPX_EHIP;
    if (hipEventCreateWithFlags(&ctx->band_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->band_join, hipEventDisableTiming) != hipSuccess)
      return FPX_EHIP;
  }
  // FPX_DEBUG_CHECKS in the environment: a trusted caller's `independent` is checked all the same (a false claim would let
  // the range chain's plain store of an acceptor's round race with k_finalize's atomicMax: silently wrong state, ADVICE
  // r05).  (Not FPX_DEBUG, which only logs: the check sends the step down the validating path, 0.076 -> 0.087 ms per band.)
  const bool debug_env = getenv("FPX_DEBUG_CHECKS") != nullptr;
  const bool validate = !((ctx->cfg.flags & FPX_F_TRUSTED) && !ctx->force_validate) || debug_env;
  if (validate) {
    // is the caller's word good?  Checked before anything is applied (FPX_EORDER, nothing applied); the halves themselves
    // then run one after the other: their run-contract checks share one scratch table (State::run_round)
    const int L = ctx->g.num_leader_groups;
    if ((rc = grow(ctx, &ctx->d_band, (size_t)L * 4))) return rc;
    fill32(ctx, (int32_t*)ctx->d_band.p, 0, (size_t)L);
    hipLaunchKernelGGL(k_band_mark, dim3((n_ranges + 255) / 256), dim3(256), 0, ctx->stream, ctx->g, n_ranges, d_slot_start, (int32_t*)ctx->d_band.p);
    hipLaunchKernelGGL(k_band_check, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->g, ctx->st, n, d_slot, d_round,
                       (const int32_t*)ctx->d_band.p, ctx->index_base);
    // a launch that failed here would skip the one check this path exists for, silently
    if ((rc = launch_check(ctx))) return rc;
    if ((rc = fpx_phase2_fused_dev(ctx, n, d_slot, d_round, d_value_id, d_target_mask, d_chosen, d_chosen_round, d_chosen_value, d_nack_round)))
      return rc;
    return fpx_noop_ranges_fused_dev(ctx, n_ranges, d_slot_start, d_slot_end, d_range_round, d_range_target_masks, d_range_vote_bits,
                                     d_range_nack_bits, d_range_nack_round, d_range_is_new, d_range_chosen);
  }
  // the ranges' scratch is sized on the main stream (grow may reallocate: nothing of the side stream is in flight here,
  // the previous step joined)
  const size_t words = (size_t)n_ranges * ctx->g.num_groups * 4;
  if ((rc = ranges_ctx_ok(ctx, n_ranges))) return rc;
  if ((rc = grow(ctx, &ctx->d_rng, (size_t)n_ranges * 4 + 64 + (d_range_vote_bits ? 0 : words * 8)))) return rc;
  {
    Batch b;
    memset(&b, 0, sizeof(b));
    b.n = n, b.slot = d_slot, b.round = d_round, b.value = d_value_id, b.target = d_target_mask;
    b.chosen = d_chosen, b.chosen_round = d_chosen_round, b.chosen_value = d_chosen_value, b.nack_round = d_nack_round;
    RangeBatch rb;
    memset(&rb, 0, sizeof(rb));
    rb.n = n_ranges, rb.start = d_slot_start, rb.end = d_slot_end, rb.round = d_range_round, rb.target = d_range_target_masks;
    rb.entry = (int32_t*)ctx->d_rng.p;
    rb.vote_bits = d_range_vote_bits ? d_range_vote_bits : (uint64_t*)((char*)ctx->d_rng.p + (((size_t)n_ranges * 4 + 63) & ~(size_t)63));
    rb.nack_bits = d_range_nack_bits, rb.nack_round = d_range_nack_round, rb.is_new = d_range_is_new, rb.chosen = d_range_chosen;
    bool done = false;
    if ((rc = enqueue_band_merged(ctx, b, rb, &done)) || done) return rc;
  }
  HIPCHK(ctx, hipEventRecord(ctx->band_fork, ctx->stream));
  HIPCHK(ctx, hipStreamWaitEvent(ctx->band_stream, ctx->band_fork, 0));
  hipStream_t main_stream = ctx->stream;
  ctx->stream = ctx->band_stream;
  rc = fpx_noop_ranges_fused_dev(ctx, n_ranges, d_slot_start, d_slot_end, d_range_round, d_range_target_masks, d_range_vote_bits,
                                 d_range_nack_bits, d_range_nack_round, d_range_is_new, d_range_chosen);
  const hipError_t je = hipEventRecord(ctx->band_join, ctx->band_stream);
  ctx->stream = main_stream;
  int rc2 = fpx_phase2_fused_dev(ctx, n, d_slot, d_round, d_value_id, d_target_mask, d_chosen, d_chosen_round, d_chosen_value, d_nack_round);
  if (je == hipSuccess) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->band_join, 0));
  if (je != hipSuccess) return FPX_EHIP;
  return rc ? rc : rc2;
}

int64_t fpx_band_merged_steps(fpx_ctx* ctx) { return ctx ? ctx->band_merged_steps : 0; }

namespace {
struct RangeKey {
  int32_t s, e, r;
  bool operator<(const RangeKey& o) const { return s != o.s ? s < o.s : (e != o.e ? e < o.e : r < o.r); }
};

// host staging of a range batch: [start | end | round | entry | nack_round] int32, [is_new | chosen] u8, then the
// bitmaps [target | votes | nacks] u64
struct RangeStage {
  int32_t *start, *end, *round, *entry, *nack_round;
  uint8_t *is_new, *chosen;
  uint64_t *target, *votes, *nacks;
};

int stage_ranges(fpx_ctx* ctx, int n, RangeStage* st) {
  const size_t words = (size_t)n * ctx->g.num_groups * 4;
  const size_t ints = ((size_t)n * 5 * 4 + 63) & ~(size_t)63, bytes = ((size_t)n * 2 + 63) & ~(size_t)63;
  int rc = grow(ctx, &ctx->d_rng, ints + bytes + 3 * words * 8);
  if (rc) return rc;
  char* p = (char*)ctx->d_rng.p;
  st->start = (int32_t*)p, st->end = st->start + n, st->round = st->end + n, st->entry = st->round + n;
  st->nack_round = st->entry + n;
  st->is_new = (uint8_t*)(p + ints), st->chosen = st->is_new + n;
  st->target = (uint64_t*)(p + ints + bytes), st->votes = st->target + words, st->nacks = st->votes + words;
  return FPX_OK;
}

// host batches: argument check (the first offender for fpx_error_detail) and the cuts that make every piece a run
int check_ranges(fpx_ctx* ctx, int n, const int32_t* start, const int32_t* end, const int32_t* round) {
  for (int i = 0; i < n; ++i)
    if (start[i] < 0 || end[i] < start[i] || end[i] > ctx->g.S || round[i] < 0 || round[i] > MAX_ROUND) {
      ctx->err_index = i, ctx->err_slot = start[i], ctx->err_round = round[i];
      return FPX_EINVAL;
    }
  return FPX_OK;
}

void split_range_runs(fpx_ctx* ctx, int n, const int32_t* start, const int32_t* end, const int32_t* round,
                      bool one_round_per_group, bool distinct_keys, std::vector<int>* cuts) {
  cuts->clear();
  cuts->push_back(0);
  std::vector<int32_t> cur((size_t)ctx->g.num_leader_groups, -1);
  std::vector<int> touched;
  std::set<RangeKey> keys;
  for (int i = 0; i < n; ++i) {
    const int lg = start[i] % ctx->g.num_leader_groups;
    bool cut = one_round_per_group && cur[lg] != -1 && cur[lg] != round[i];
    const RangeKey k{start[i], end[i], round[i]};
    if (distinct_keys && !cut) cut = keys.count(k) != 0;
    if (cut) {
      cuts->push_back(i);
      for (int t : touched) cur[t] = -1;
      touched.clear();
      keys.clear();
    }
    if (cur[lg] == -1) touched.push_back(lg);
    cur[lg] = round[i];
    if (distinct_keys) keys.insert(k);
  }
  cuts->push_back(n);
}
}  // namespace

// host-pointer driver shared by the four host entry points
static int32_t host_ranges(fpx_ctx* ctx, int mode, int32_t n, const int32_t* start, const int32_t* end,
                           const int32_t* round, const uint64_t* target_masks, const uint64_t* votes_in,
                           uint64_t* vote_bits, uint64_t* nack_bits, int32_t* nack_round, uint8_t* is_new,
                           uint8_t* chosen) {
  int rc = ranges_ctx_ok(ctx, n);
  if (rc) return rc;
  if (n == 0) return FPX_OK;
  if (!start || !end || !round || (mode == RANGES_TALLY && !votes_in)) return FPX_EINVAL;
  if ((rc = check_ranges(ctx, n, start, end, round))) return rc;
  const size_t words = (size_t)n * ctx->g.num_groups * 4, per = (size_t)ctx->g.num_groups * 4;
  RangeStage sg;
  if ((rc = stage_ranges(ctx, n, &sg))) return rc;
  HIPCHK(ctx, hipMemcpyAsync(sg.start, start, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(sg.end, end, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(sg.round, round, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  if (target_masks) HIPCHK(ctx, hipMemcpyAsync(sg.target, target_masks, words * 8, hipMemcpyHostToDevice, ctx->stream));
  if (mode == RANGES_TALLY) HIPCHK(ctx, hipMemcpyAsync(sg.votes, votes_in, words * 8, hipMemcpyHostToDevice, ctx->stream));
  HostRun host_run(ctx);
  std::vector<int> cuts;
  // one round per leader group in every run: the acceptors' scalar needs it, and the table insert relies on
  // "same (start, end) in one launch => same key"
  split_range_runs(ctx, n, start, end, round, true, mode == RANGES_TALLY, &cuts);
  for (size_t k = 0; k + 1 < cuts.size(); ++k) {
    const int lo = cuts[k], len = cuts[k + 1] - cuts[k];
    RangeBatch b;
    memset(&b, 0, sizeof(b));
    b.n = len, b.start = sg.start + lo, b.end = sg.end + lo, b.round = sg.round + lo;
    b.target = target_masks ? sg.target + (size_t)lo * per : nullptr;
    b.entry = sg.entry + lo, b.nack_round = sg.nack_round + lo;
    b.vote_bits = sg.votes + (size_t)lo * per, b.nack_bits = sg.nacks + (size_t)lo * per;
    b.is_new = sg.is_new + lo, b.chosen = sg.chosen + lo;
    if (mode == RANGES_TALLY) b.votes_in = sg.votes + (size_t)lo * per;
    if ((rc = enqueue_ranges(ctx, b, mode))) return rc;
  }
  if (vote_bits) HIPCHK(ctx, hipMemcpyAsync(vote_bits, sg.votes, words * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (nack_bits) HIPCHK(ctx, hipMemcpyAsync(nack_bits, sg.nacks, words * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (nack_round) HIPCHK(ctx, hipMemcpyAsync(nack_round, sg.nack_round, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (is_new) HIPCHK(ctx, hipMemcpyAsync(is_new, sg.is_new, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  if (chosen) HIPCHK(ctx, hipMemcpyAsync(chosen, sg.chosen, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  return fetch_status(ctx);
}

int32_t fpx_noop_ranges_fused(fpx_ctx* ctx, int32_t n, const int32_t* slot_start, const int32_t* slot_end,
                              const int32_t* round, const uint64_t* target_masks, uint64_t* vote_bits,
                              uint64_t* nack_bits, int32_t* nack_round, uint8_t* is_new, uint8_t* chosen) {
  DeviceGuard _dg(ctx);
  return host_ranges(ctx, RANGES_FUSED, n, slot_start, slot_end, round, target_masks, nullptr, vote_bits, nack_bits,
                     nack_round, is_new, chosen);
}

int32_t fpx_acceptor_phase2a_noop_ranges(fpx_ctx* ctx, int32_t n, const int32_t* slot_start, const int32_t* slot_end,
                                         const int32_t* round, const uint64_t* target_masks, uint64_t* vote_bits,
                                         uint64_t* nack_bits, int32_t* nack_round) {
  DeviceGuard _dg(ctx);
  return host_ranges(ctx, RANGES_ACCEPTORS, n, slot_start, slot_end, round, target_masks, nullptr, vote_bits, nack_bits,
                     nack_round, nullptr, nullptr);
}

int32_t fpx_proxy_open_noop_ranges(fpx_ctx* ctx, int32_t n, const int32_t* slot_start, const int32_t* slot_end,
                                   const int32_t* round, uint8_t* is_new) {
  DeviceGuard _dg(ctx);
  return host_ranges(ctx, RANGES_OPEN, n, slot_start, slot_end, round, nullptr, nullptr, nullptr, nullptr, nullptr,
                     is_new, nullptr);
}

int32_t fpx_proxy_phase2b_noop_ranges(fpx_ctx* ctx, int32_t n, const int32_t* slot_start, const int32_t* slot_end,
                                      const int32_t* round, const uint64_t* vote_bits, uint8_t* newly_chosen) {
  DeviceGuard _dg(ctx);
  return host_ranges(ctx, RANGES_TALLY, n, slot_start, slot_end, round, nullptr, vote_bits, nullptr, nullptr, nullptr,
                     nullptr, newly_chosen);
}

// the single-range entry points: batches of one
int32_t fpx_acceptor_phase2a_noop_range(fpx_ctx* ctx, int32_t slot_start, int32_t slot_end, int32_t round,
                                        const uint64_t* target_masks, uint64_t* vote_bits, uint64_t* nack_bits,
                                        int32_t* nack_round) {
  return fpx_acceptor_phase2a_noop_ranges(ctx, 1, &slot_start, &slot_end, &round, target_masks, vote_bits, nack_bits,
                                          nack_round);
}

int32_t fpx_proxy_open_noop_range(fpx_ctx* ctx, int32_t slot_start, int32_t slot_end, int32_t round, uint8_t* is_new) {
  return fpx_proxy_open_noop_ranges(ctx, 1, &slot_start, &slot_end, &round, is_new);
}

int32_t fpx_proxy_phase2b_noop_range(fpx_ctx* ctx, int32_t slot_start, int32_t slot_end, int32_t round,
                                     const uint64_t* vote_bits, uint8_t* newly_chosen) {
  return fpx_proxy_phase2b_noop_ranges(ctx, 1, &slot_start, &slot_end, &round, vote_bits, newly_chosen);
}

// the proxy leader's tally of one range (parity): state 0 = unknown, 1 = Pending, 2 = Done; votes num_groups x 4 words
int32_t fpx_read_range_tally(fpx_ctx* ctx, int32_t slot_start, int32_t slot_end, int32_t round, int32_t* state,
                             uint64_t* vote_bits) {
  DeviceGuard _dg(ctx);
  int rc = ranges_ctx_ok(ctx, 0);
  if (rc) return rc;
  const size_t words = (size_t)ctx->g.num_groups * 4;
  if ((rc = grow(ctx, &ctx->d_scratch, (words + 1) * 8))) return rc;
  hipLaunchKernelGGL(k_ranges_read, dim3(1), dim3(64), 0, ctx->stream, ctx->g, ctx->rt[ctx->rt_cur], slot_start, slot_end,
                     round, (uint64_t*)ctx->d_scratch.p);
  if ((rc = launch_check(ctx))) return rc;
  std::vector<uint64_t> h(words + 1);
  HIPCHK(ctx, hipMemcpyAsync(h.data(), ctx->d_scratch.p, h.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (state) *state = (int32_t)h[0];
  if (vote_bits) memcpy(vote_bits, h.data() + 1, words * 8);
  return FPX_OK;
}

// ---- f1: replica log ------------------------------------------------------------------------------------
int32_t fpx_replica_chosen_dev(fpx_ctx* ctx, int32_t n, const int32_t* d_slot, const int32_t* d_value_id,
                               const uint8_t* d_mask) {
  DeviceGuard _dg(ctx);
  if (!ctx || n < 0) return FPX_EINVAL;
  if (n == 0) return FPX_OK;
  Batch b;
  memset(&b, 0, sizeof(b));
  b.n = n, b.slot = d_slot, b.value = d_value_id, b.mask = d_mask;
  int rc = enqueue_validate(ctx, b, false);
  if (rc) return rc;
  const int ingest_grid = std::max(1, std::min({(n + 255) / 256, ctx->num_cus * 8, LG_MAX_PARTS}));
  hipLaunchKernelGGL(k_log_ingest, dim3(ingest_grid), dim3(256), 0, ctx->stream, ctx->g, ctx->st, b);
  hipLaunchKernelGGL(k_log_prep, dim3(1), dim3(256), 0, ctx->stream, ctx->g, ctx->st, ingest_grid);
  hipLaunchKernelGGL(k_log_scan, dim3(ctx->num_cus * 4), dim3(256), 0, ctx->stream, ctx->g, ctx->st);
  hipLaunchKernelGGL(k_log_commit, dim3(1), dim3(1), 0, ctx->stream, ctx->st);
  return launch_check(ctx);
}

int32_t fpx_replica_state(fpx_ctx* ctx, int32_t* executed_watermark, int32_t* num_chosen) {
  DeviceGuard _dg(ctx);
  if (!ctx) return FPX_EINVAL;
  int32_t h[2] = {0, 0};
  HIPCHK(ctx, hipMemcpyAsync(h, ctx->st.log_scalars, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (executed_watermark) *executed_watermark = h[0];
  if (num_chosen) *num_chosen = h[1];
  return FPX_OK;
}

int32_t fpx_replica_chosen(fpx_ctx* ctx, int32_t n, const int32_t* slot, const int32_t* value_id, const uint8_t* mask,
                           int32_t* executed_watermark, int32_t* num_chosen) {
  DeviceGuard _dg(ctx);
  if (!ctx || n < 0 || (n > 0 && (!slot || !value_id))) return FPX_EINVAL;
  for (int i = 0; i < n; ++i) {
    if ((!mask || mask[i]) && (slot[i] < 0 || slot[i] >= ctx->g.S)) {
      ctx->err_index = i, ctx->err_slot = slot[i], ctx->err_round = -1;
      return FPX_EINVAL;
    }
  }
  int rc;
  if (n > 0) {
    // masked-out messages must not trip the slot checks on the device: compact them away
    std::vector<int32_t> s, v;
    s.reserve(n), v.reserve(n);
    for (int i = 0; i < n; ++i)
      if (!mask || mask[i]) s.push_back(slot[i]), v.push_back(value_id[i]);
    const int m = (int)s.size();
    if (m > 0) {
      if ((rc = h2d(ctx, &ctx->d_slot, s.data(), m))) return rc;
      if ((rc = h2d(ctx, &ctx->d_value, v.data(), m))) return rc;
      std::vector<int> cuts;
      split_runs(ctx, m, s.data(), nullptr, false, &cuts);
      for (size_t k = 0; k + 1 < cuts.size(); ++k) {
        const int lo = cuts[k], len = cuts[k + 1] - cuts[k];
        rc = fpx_replica_chosen_dev(ctx, len, (int32_t*)ctx->d_slot.p + lo, (int32_t*)ctx->d_value.p + lo, nullptr);
        if (rc) return rc;
      }
    }
  }
  if ((rc = fpx_replica_state(ctx, executed_watermark, num_chosen))) return rc;
  return fetch_status(ctx);
}

int32_t fpx_replica_chosen_noop_range(fpx_ctx* ctx, int32_t slot_start, int32_t slot_end, int32_t* executed_watermark,
                                      int32_t* num_chosen) {
  DeviceGuard _dg(ctx);
  if (!ctx || slot_start < 0 || slot_end > ctx->g.S) return FPX_EINVAL;
  const int stride = ctx->cfg.num_leader_groups;
  const int count = slot_start < slot_end ? (slot_end - slot_start + stride - 1) / stride : 0;
  int32_t first = count;
  if (count > 0) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->st.log_scalars + LG_RANGE_FIRST, &first, 4, hipMemcpyHostToDevice, ctx->stream));
    const int grid = std::max(1, std::min((count + 255) / 256, ctx->num_cus * 8));
    hipLaunchKernelGGL(k_log_range_first, dim3(grid), dim3(256), 0, ctx->stream, ctx->st, slot_start, stride, count);
    hipLaunchKernelGGL(k_log_range_fill, dim3(grid), dim3(256), 0, ctx->stream, ctx->st, slot_start, stride);
    int rc = launch_check(ctx);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(&first, ctx->st.log_scalars + LG_RANGE_FIRST, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (first == count) {  // the handler ran to its end: executeLog (mencius/Replica.scala:485)
    hipLaunchKernelGGL(k_log_prep, dim3(1), dim3(256), 0, ctx->stream, ctx->g, ctx->st, 0);
    hipLaunchKernelGGL(k_log_scan, dim3(ctx->num_cus * 4), dim3(256), 0, ctx->stream, ctx->g, ctx->st);
    hipLaunchKernelGGL(k_log_commit, dim3(1), dim3(1), 0, ctx->stream, ctx->st);
    int rc = launch_check(ctx);
    if (rc) return rc;
  }
  int rc = fpx_replica_state(ctx, executed_watermark, num_chosen);
  if (rc) return rc;
  return fetch_status(ctx);
}

int32_t fpx_replica_read_log(fpx_ctx* ctx, int32_t first, int32_t count, int32_t* values, uint8_t* present) {
  DeviceGuard _dg(ctx);
  if (!ctx || first < 0 || count < 0 || (int64_t)first + count > ctx->g.S) return FPX_EINVAL;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (values && count) HIPCHK(ctx, hipMemcpy(values, ctx->st.log_value + first, (size_t)count * 4, hipMemcpyDeviceToHost));
  if (present && count) HIPCHK(ctx, hipMemcpy(present, ctx->st.log_present + first, (size_t)count, hipMemcpyDeviceToHost));
  return FPX_OK;
}

// ---- f2: Phase-1 recovery scan ------------------------------------------------------------------------
int32_t fpx_leader_phase1b_scan(fpx_ctx* ctx, int32_t chosen_watermark, const uint64_t* quorum_masks, int32_t cap,
                                int32_t* max_slot, int32_t* safe_round, int32_t* safe_value) {
  DeviceGuard _dg(ctx);
  if (!ctx || !quorum_masks || chosen_watermark < 0 || cap < 0) return FPX_EINVAL;
  const int ng = ctx->g.ngroups;
  int rc;
  if ((rc = grow(ctx, &ctx->d_target, (size_t)ng * 32 + 64))) return rc;
  uint64_t* d_q = (uint64_t*)ctx->d_target.p;
  int32_t* d_max = (int32_t*)(d_q + (size_t)ng * 4);
  HIPCHK(ctx, hipMemcpyAsync(d_q, quorum_masks, (size_t)ng * 32, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(d_max, 0xFF, 4, ctx->stream));
  const int ntab = ng * ctx->g.R;
  hipLaunchKernelGGL(k_quorum_max_slot, dim3((ntab + 255) / 256), dim3(256), 0, ctx->stream, ctx->g, ctx->st, d_q,
                     chosen_watermark, d_max);
  if ((rc = launch_check(ctx))) return rc;
  int32_t hmax = -1;
  HIPCHK(ctx, hipMemcpyAsync(&hmax, d_max, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (max_slot) *max_slot = hmax;
  // for (slot <- chosenWatermark to maxSlot)   Leader.scala:553
  const int64_t want = (int64_t)hmax - chosen_watermark + 1;
  const int count = (int)std::max<int64_t>(0, std::min<int64_t>(want, cap));
  if (count == 0) return FPX_OK;
  if ((rc = grow(ctx, &ctx->d_i32_a, (size_t)count * 4))) return rc;
  if ((rc = grow(ctx, &ctx->d_i32_b, (size_t)count * 4))) return rc;
  int32_t *d_sr = (int32_t*)ctx->d_i32_a.p, *d_sv = (int32_t*)ctx->d_i32_b.p;
  switch (ctx->lanes_per_slot) {
    case 1: launch_p1b<1>(ctx, d_q, chosen_watermark, count, d_sr, d_sv); break;
    case 2: launch_p1b<2>(ctx, d_q, chosen_watermark, count, d_sr, d_sv); break;
    case 4: launch_p1b<4>(ctx, d_q, chosen_watermark, count, d_sr, d_sv); break;
    case 8: launch_p1b<8>(ctx, d_q, chosen_watermark, count, d_sr, d_sv); break;
    case 16: launch_p1b<16>(ctx, d_q, chosen_watermark, count, d_sr, d_sv); break;
    case 32: launch_p1b<32>(ctx, d_q, chosen_watermark, count, d_sr, d_sv); break;
    default: launch_p1b<64>(ctx, d_q, chosen_watermark, count, d_sr, d_sv); break;
  }
  if ((rc = launch_check(ctx))) return rc;
  if ((rc = d2h(ctx, safe_round, ctx->d_i32_a, (size_t)count))) return rc;
  if ((rc = d2h(ctx, safe_value, ctx->d_i32_b, (size_t)count))) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FPX_OK;
}

// ---- readback ----------------------------------------------------------------------------------------
int32_t fpx_read_state(fpx_ctx* ctx, int32_t* vote_round, int32_t* vote_value, int32_t* ballot) {
  DeviceGuard _dg(ctx);
  if (!ctx) return FPX_EINVAL;
  const size_t S = (size_t)ctx->g.S, R = (size_t)ctx->g.R, RS = (size_t)ctx->g.RS;
  if (ballot) {
    int rcf = flush_promises(ctx);
    if (rcf) return rcf;
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  // device rows are RS cells long, the caller's are R
  // leader-group-major rows: leader group lg's rows are contiguous on the device and every L-th row of the caller's array
  const size_t L = ctx->g.lg_rows ? (size_t)ctx->g.num_leader_groups : 1, rows = S / L;
  auto fetch = [&](int32_t* dst, const int32_t* src, size_t src_stride) -> int {
    for (size_t lg = 0; lg < L; ++lg)
      HIPCHK(ctx, hipMemcpy2D(dst + lg * R, L * R * 4, src + lg * rows * src_stride, src_stride * 4, R * 4, rows, hipMemcpyDeviceToHost));
    return FPX_OK;
  };
  int rc2;
  if (vote_round && (rc2 = fetch(vote_round, ctx->st.vote_round, (size_t)ctx->g.VS))) return rc2;
  if (vote_value && (rc2 = fetch(vote_value, ctx->st.vote_value, (size_t)ctx->g.VS))) return rc2;
  if (ballot) {
    if (ctx->st.ballot) {
      if ((rc2 = fetch(ballot, ctx->st.ballot, RS))) return rc2;
    } else {
      std::fill(ballot, ballot + S * R, -1);
    }
  }
  return FPX_OK;
}

int32_t fpx_read_scalars(fpx_ctx* ctx, int32_t* promised, int32_t* max_voted_slot) {
  DeviceGuard _dg(ctx);
  if (!ctx) return FPX_EINVAL;
  const size_t nsc = (size_t)ctx->g.ngroups * ctx->g.R;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (promised) HIPCHK(ctx, hipMemcpy(promised, ctx->st.promised, nsc * 4, hipMemcpyDeviceToHost));
  if (max_voted_slot) HIPCHK(ctx, hipMemcpy(max_voted_slot, ctx->st.max_voted, nsc * 4, hipMemcpyDeviceToHost));
  return FPX_OK;
}

int32_t fpx_read_acceptor(fpx_ctx* ctx, int32_t group, int32_t replica, int32_t* promised, int32_t* max_voted_slot,
                          int32_t* vote_round, int32_t* vote_value, int32_t* ballot) {
  DeviceGuard _dg(ctx);
  if (!ctx || group < 0 || group >= ctx->g.ngroups || replica < 0 || replica >= ctx->g.R) return FPX_EINVAL;
  const size_t e = (size_t)group * ctx->g.R + replica;
  const size_t S = (size_t)ctx->g.S;
  int rc;
  if (ballot && (rc = flush_promises(ctx))) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (promised) {
    if (ctx->g.per_slot) *promised = -1;
    else HIPCHK(ctx, hipMemcpy(promised, ctx->st.promised + e, 4, hipMemcpyDeviceToHost));
  }
  if (max_voted_slot) HIPCHK(ctx, hipMemcpy(max_voted_slot, ctx->st.max_voted + e, 4, hipMemcpyDeviceToHost));
  if (vote_round || vote_value || ballot) {
    if ((rc = grow(ctx, &ctx->d_scratch, S * 12 + 128))) return rc;
    int32_t* base = (int32_t*)((char*)ctx->d_scratch.p + 128);
    hipLaunchKernelGGL(k_gather_acceptor, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, ctx->stream, ctx->g, ctx->st,
                       group, replica, base, base + S, base + 2 * S);
    if ((rc = launch_check(ctx))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (vote_round) HIPCHK(ctx, hipMemcpy(vote_round, base, S * 4, hipMemcpyDeviceToHost));
    if (vote_value) HIPCHK(ctx, hipMemcpy(vote_value, base + S, S * 4, hipMemcpyDeviceToHost));
    if (ballot) HIPCHK(ctx, hipMemcpy(ballot, base + 2 * S, S * 4, hipMemcpyDeviceToHost));
  }
  return FPX_OK;
}

int32_t fpx_acceptor_max_voted_in(fpx_ctx* ctx, int32_t group, int32_t replica, int32_t first_slot, int32_t count,
                                  int32_t* max_slot) {
  DeviceGuard _dg(ctx);
  if (!ctx || !max_slot || group < 0 || group >= ctx->g.ngroups || replica < 0 || replica >= ctx->g.R || first_slot < 0 || count < 0 ||
      (int64_t)first_slot + count > ctx->g.S)
    return FPX_EINVAL;
  *max_slot = -1;
  if (count == 0) return FPX_OK;
  int rc;
  if ((rc = grow(ctx, &ctx->d_scratch, 128))) return rc;
  int32_t* d = (int32_t*)ctx->d_scratch.p;
  fill32(ctx, d, -1, 1);
  hipLaunchKernelGGL(k_max_voted_in, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, ctx->g, ctx->st, group, replica,
                     first_slot, count, d);
  if ((rc = launch_check(ctx))) return rc;
  HIPCHK(ctx, hipMemcpyAsync(max_slot, d, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FPX_OK;
}

// Acceptor.handlePhase1a's Phase1b.info (multipaxos/Acceptor.scala:166-178): the acceptor's votes in slots >=
// chosen_watermark, ascending (states.iteratorFrom).  Phase 1 is off the steady path: one gather of the acceptor's
// column, compacted on the host.
int32_t fpx_acceptor_phase1b_info(fpx_ctx* ctx, int32_t group, int32_t replica, int32_t chosen_watermark, int32_t cap,
                                  int32_t* count, int32_t* slot, int32_t* vote_round, int32_t* vote_value) {
  DeviceGuard _dg(ctx);
  if (!ctx || !count || cap < 0 || group < 0 || group >= ctx->g.ngroups || replica < 0 || replica >= ctx->g.R ||
      (cap > 0 && (!slot || !vote_round || !vote_value)))
    return FPX_EINVAL;
  *count = 0;
  const size_t e = (size_t)group * ctx->g.R + replica;
  const int64_t S = ctx->g.S;
  int rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  int32_t max_voted = -1;
  HIPCHK(ctx, hipMemcpy(&max_voted, ctx->st.max_voted + e, 4, hipMemcpyDeviceToHost));
  const int64_t lo = chosen_watermark < 0 ? 0 : chosen_watermark;
  if (max_voted < lo || lo >= S) return FPX_OK;  // no vote at or above the watermark
  if ((rc = grow(ctx, &ctx->d_scratch, (size_t)S * 12 + 128))) return rc;
  int32_t* base = (int32_t*)((char*)ctx->d_scratch.p + 128);
  hipLaunchKernelGGL(k_gather_acceptor, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, ctx->stream, ctx->g, ctx->st,
                     group, replica, base, base + S, base + 2 * S);
  if ((rc = launch_check(ctx))) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  const size_t len = (size_t)(max_voted - lo + 1);
  std::vector<int32_t> vr(len), vv(len);
  HIPCHK(ctx, hipMemcpy(vr.data(), base + lo, len * 4, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(vv.data(), base + S + lo, len * 4, hipMemcpyDeviceToHost));
  int32_t k = 0;
  for (size_t j = 0; j < len; ++j) {
    if (vr[j] < 0) continue;  // no vote in the slot (or another group's slot)
    if (k < cap) slot[k] = (int32_t)(lo + (int64_t)j), vote_round[k] = vr[j], vote_value[k] = vv[j];
    ++k;
  }
  *count = k;
  return FPX_OK;
}

// ---- multi-GPU: RCCL behind the C ABI ------------------------------------------------------------------
static RcclApi* rccl_bind();
// bound once per process, also when two contexts create their communicators from two threads (a function-local
// static's initialiser runs exactly once)
static RcclApi* rccl() {
  static RcclApi* const bound = rccl_bind();
  return bound;
}
static RcclApi* rccl_bind() {
  static RcclApi api;
  void* h = nullptr;
  const char* env = getenv("FPX_RCCL_LIB");
  if (env && *env) h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
  // an RCCL that is already mapped (PyTorch's bundled copy) wins over loading a second one
  const char* loaded[] = {"librccl.so", "librccl.so.1"};
  for (size_t i = 0; !h && i < 2; ++i) h = dlopen(loaded[i], RTLD_NOW | RTLD_NOLOAD);
  const char* fresh[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (size_t i = 0; !h && i < 3; ++i) h = dlopen(fresh[i], RTLD_NOW | RTLD_GLOBAL);
  if (!h) return nullptr;
  auto sym = [&](const char* name) { return dlsym(h, name); };
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
  api.ReduceScatter = reinterpret_cast<decltype(api.ReduceScatter)>(sym("ncclReduceScatter"));
  api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
  api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
  api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
  api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
  if (!api.GroupStart || !api.GroupEnd) api.GroupStart = api.GroupEnd = nullptr;
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.ReduceScatter || !api.AllGather) return nullptr;
  api.lib = h;
  return &api;
}

#define RCCLCHK(ctx, expr)                 \
  do {                                     \
    int _r = (expr);                       \
    if (_r != 0) {                         \
      if (ctx) (ctx)->last_rccl = _r;      \
      return FPX_ERCCL;                    \
    }                                      \
  } while (0)

int32_t fpx_comm_unique_id(uint8_t id[FPX_COMM_ID_BYTES]) {
  if (!id) return FPX_EINVAL;
  RcclApi* r = rccl();
  if (!r) return FPX_ERCCL;
  RcclUniqueId u;
  if (r->GetUniqueId(&u) != 0) return FPX_ERCCL;
  memcpy(id, u.internal, FPX_COMM_ID_BYTES);
  return FPX_OK;
}

int32_t fpx_comm_create(fpx_ctx* ctx, const uint8_t id[FPX_COMM_ID_BYTES], int32_t rank, int32_t world) {
  DeviceGuard _dg(ctx);
  if (!ctx || !id || world < 1 || rank < 0 || rank >= world || ctx->comm) return FPX_EINVAL;
  RcclApi* r = rccl();
  if (!r) return FPX_ERCCL;
  RcclUniqueId u;
  memcpy(u.internal, id, FPX_COMM_ID_BYTES);
  RcclComm c = nullptr;
  RCCLCHK(ctx, r->CommInitRank(&c, world, u, rank));  // binds to the current device = the context's
  ctx->comm = c;
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  return FPX_OK;
}

int32_t fpx_comm_destroy(fpx_ctx* ctx) {
  DeviceGuard _dg(ctx);
  if (!ctx) return FPX_EINVAL;
  if (!ctx->comm) return FPX_OK;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  RcclApi* r = rccl();
  if (r) RCCLCHK(ctx, r->CommDestroy(ctx->comm));
  ctx->comm = nullptr;
  ctx->comm_rank = 0, ctx->comm_world = 1;
  return FPX_OK;
}

int32_t fpx_comm_info(fpx_ctx* ctx, int32_t* rank, int32_t* world) {
  if (!ctx) return FPX_EINVAL;
  if (rank) *rank = ctx->comm_rank;
  if (world) *world = ctx->comm_world;
  return FPX_OK;
}

int32_t fpx_last_rccl_error(fpx_ctx* ctx) { return ctx ? ctx->last_rccl : 0; }

int32_t fpx_profile_read_collective(fpx_ctx* ctx, int32_t* launches, double* total_ms) {
  DeviceGuard _dg(ctx);
  if (!ctx) return FPX_EINVAL;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  double sum = 0;
  for (size_t i = 0; i + 1 < ctx->cev_used; i += 2) {
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->cev[i], ctx->cev[i + 1]));
    sum += ms;
  }
  if (launches) *launches = (int32_t)(ctx->cev_used / 2);
  if (total_ms) *total_ms = sum;
  ctx->cev_used = 0;
  return FPX_OK;
}

int32_t fpx_phase2_replica_sharded_dev(fpx_ctx* ctx, int32_t n, const int32_t* d_slot, const int32_t* d_round,
                                       const int32_t* d_value_id, const uint64_t* d_target_mask, uint8_t* d_chosen,
                                       int32_t* d_chosen_round, int32_t* d_chosen_value, int32_t* d_nack_round) {
  DeviceGuard _dg(ctx);
  if (!ctx || n < 0) return FPX_EINVAL;
  const int world = ctx->comm_world, rank = ctx->comm_rank;
  if (n % world) return FPX_EINVAL;
  if (n == 0) return FPX_OK;
  const int per = n / world, lo = rank * per;
  int rc;
  if ((rc = grow(ctx, &ctx->d_part, (size_t)n * 32))) return rc;
  if (ctx->comm && (rc = grow(ctx, &ctx->d_mine, (size_t)per * 32))) return rc;
  uint64_t* part = (uint64_t*)ctx->d_part.p;
  // K1: my acceptor columns vote on every slot of the batch; partial bitmaps (bits in my range only) to HBM
  Batch b;
  memset(&b, 0, sizeof(b));
  b.n = n, b.slot = d_slot, b.round = d_round, b.value = d_value_id, b.target = d_target_mask;
  b.vote_bits = part, b.nack_round = d_nack_round;
  if ((rc = enqueue_phase2(ctx, b, false))) return rc;
  const uint64_t* mine = part + (size_t)lo * 4;
  if (ctx->comm) {  // also on a communicator of one rank: the same call path, RCCL copies
    // the exchange step: sum (== OR) of the partial bitmaps, scattered so that I receive my slots' rows
    RcclApi* r = rccl();
    if (!r) return FPX_ERCCL;
    const bool prof = ctx->profiling && ctx->cev_used + 2 <= ctx->cev.size();
    if (prof) HIPCHK(ctx, hipEventRecord(ctx->cev[ctx->cev_used], ctx->stream));
    RCCLCHK(ctx, r->ReduceScatter(part, ctx->d_mine.p, (size_t)per * 4, RCCL_UINT64, RCCL_SUM, ctx->comm, ctx->stream));
    if (prof) {
      HIPCHK(ctx, hipEventRecord(ctx->cev[ctx->cev_used + 1], ctx->stream));
      ctx->cev_used += 2;
    }
    mine = (const uint64_t*)ctx->d_mine.p;
    // a Nack from ANY rank's acceptors reaches the caller (Leader.handleNack reacts to the largest round, Leader.scala:
    // 672-697): max over the ranks, 4 B per slot more on the wire
    if (d_nack_round && world > 1) {
      if (!r->AllReduce) return FPX_ERCCL;
      RCCLCHK(ctx, r->AllReduce(d_nack_round, d_nack_round, (size_t)n, RCCL_INT32, RCCL_MAX, ctx->comm, ctx->stream));
    }
  }
  // ProxyLeader.handlePhase2a bookkeeping + handlePhase2b for my slice of the slots; K1's validation pass
  // covered the whole batch (slots distinct), so the slice needs none
  HostRun trusted(ctx);
  Batch o;
  memset(&o, 0, sizeof(o));
  o.n = per, o.slot = d_slot + lo, o.round = d_round + lo, o.value = d_value_id + lo;
  if ((rc = enqueue_open(ctx, o))) return rc;
  Batch t;
  memset(&t, 0, sizeof(t));
  t.n = per, t.slot = d_slot + lo, t.round = d_round + lo, t.vote_bits = const_cast<uint64_t*>(mine);
  t.chosen = d_chosen, t.chosen_round = d_chosen_round, t.chosen_value = d_chosen_value;
  return enqueue_tally(ctx, t);
}

int32_t fpx_comm_allgather_chosen_dev(fpx_ctx* ctx, int32_t n_local, const uint8_t* d_chosen,
                                      const int32_t* d_chosen_round, const int32_t* d_chosen_value,
                                      uint8_t* d_all_chosen, int32_t* d_all_round, int32_t* d_all_value) {
  DeviceGuard _dg(ctx);
  if (!ctx || n_local < 0) return FPX_EINVAL;
  if (n_local == 0) return FPX_OK;
  if (ctx->comm_world == 1) {  // degenerate: a device copy
    if (d_chosen && d_all_chosen) HIPCHK(ctx, hipMemcpyAsync(d_all_chosen, d_chosen, (size_t)n_local, hipMemcpyDeviceToDevice, ctx->stream));
    if (d_chosen_round && d_all_round) HIPCHK(ctx, hipMemcpyAsync(d_all_round, d_chosen_round, (size_t)n_local * 4, hipMemcpyDeviceToDevice, ctx->stream));
    if (d_chosen_value && d_all_value) HIPCHK(ctx, hipMemcpyAsync(d_all_value, d_chosen_value, (size_t)n_local * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return FPX_OK;
  }
  RcclApi* r = rccl();
  if (!r || !ctx->comm) return FPX_ERCCL;
  const bool prof = ctx->profiling && ctx->cev_used + 2 <= ctx->cev.size();
  if (prof) HIPCHK(ctx, hipEventRecord(ctx->cev[ctx->cev_used], ctx->stream));
  // the three arrays of the Chosen records travel as ONE group: one launch and one ring set-up instead of three
  // (ncclGroupStart / ncclGroupEnd; an RCCL without them gets three calls)
  if (r->GroupStart) RCCLCHK(ctx, r->GroupStart());
  int grc = 0;
  if (d_chosen && d_all_chosen) grc = r->AllGather(d_chosen, d_all_chosen, (size_t)n_local, RCCL_UINT8, ctx->comm, ctx->stream);
  if (!grc && d_chosen_round && d_all_round) grc = r->AllGather(d_chosen_round, d_all_round, (size_t)n_local, RCCL_INT32, ctx->comm, ctx->stream);
  if (!grc && d_chosen_value && d_all_value) grc = r->AllGather(d_chosen_value, d_all_value, (size_t)n_local, RCCL_INT32, ctx->comm, ctx->stream);
  if (r->GroupEnd) {  // (always closed, also after a failed call inside the group)
    const int erc = r->GroupEnd();
    if (!grc) grc = erc;
  }
  RCCLCHK(ctx, grc);
  if (prof) {
    HIPCHK(ctx, hipEventRecord(ctx->cev[ctx->cev_used + 1], ctx->stream));
    ctx->cev_used += 2;
  }
  return FPX_OK;
}

int32_t fpx_state_digest(fpx_ctx* ctx, uint64_t out[8]) {
  DeviceGuard _dg(ctx);
  if (!ctx || !out) return FPX_EINVAL;
  int rc;
  if ((rc = grow(ctx, &ctx->d_scratch, 128))) return rc;
  uint64_t* d = (uint64_t*)ctx->d_scratch.p;
  if ((rc = flush_promises(ctx))) return rc;
  HIPCHK(ctx, hipMemsetAsync(d, 0, 64, ctx->stream));
  const Geom& g = ctx->g;
  const int big = ctx->num_cus * 16;
  const size_t n4 = (size_t)g.S * (size_t)(g.RS / 4);
  const int gc = (int)std::max<size_t>(1, std::min<size_t>((n4 + 255) / 256, (size_t)big));
  hipLaunchKernelGGL(k_digest_cells, dim3(gc), dim3(256), 0, ctx->stream, g, ctx->st.vote_round, g.VS, d + 0);
  hipLaunchKernelGGL(k_digest_cells, dim3(gc), dim3(256), 0, ctx->stream, g, ctx->st.vote_value, g.VS, d + 1);
  if (ctx->st.ballot) hipLaunchKernelGGL(k_digest_cells, dim3(gc), dim3(256), 0, ctx->stream, g, ctx->st.ballot, g.RS, d + 2);
  const int nsc = g.ngroups * g.R;
  const int gs = std::max(1, std::min((nsc + 255) / 256, big));
  hipLaunchKernelGGL(k_digest_1d, dim3(gs), dim3(256), 0, ctx->stream, ctx->st.promised, nsc, d + 3);
  hipLaunchKernelGGL(k_digest_1d, dim3(gs), dim3(256), 0, ctx->stream, ctx->st.max_voted, nsc, d + 4);
  const int gt = std::max(1, std::min((g.S + 255) / 256, big));
  hipLaunchKernelGGL(k_digest_tally, dim3(gt), dim3(256), 0, ctx->stream, g, ctx->st, d + 5);
  hipLaunchKernelGGL(k_digest_log, dim3(gt), dim3(256), 0, ctx->stream, g, ctx->st, d + 6);
  if (ctx->rt[0].key) {
    const RangeTable& rt = ctx->rt[ctx->rt_cur];
    hipLaunchKernelGGL(k_digest_ranges, dim3(std::max(1, std::min((rt.cap + 255) / 256, big))), dim3(256), 0, ctx->stream, rt,
                       g.num_groups * 4, d + 7);
  }
  if ((rc = launch_check(ctx))) return rc;
  HIPCHK(ctx, hipMemcpyAsync(out, d, 64, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FPX_OK;
}

int32_t fpx_read_tally(fpx_ctx* ctx, int32_t slot, int32_t* num_entries, int32_t* rounds, int32_t* states,
                       int32_t* values, uint64_t* vote_bits) {
  DeviceGuard _dg(ctx);
  if (!ctx || slot < 0 || slot >= ctx->g.S) return FPX_EINVAL;
  const int wp = ctx->g.wp;
  uint32_t keys[8];
  int32_t vals[8];
  uint64_t bits[32];
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  const size_t prow = host_phys_slot(ctx->g, slot);
  HIPCHK(ctx, hipMemcpy(keys, ctx->st.pl_key + prow * wp, (size_t)wp * 4, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(vals, ctx->st.pl_value + prow * wp, (size_t)wp * 4, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(bits, ctx->st.pl_bits + prow * wp * 4, (size_t)wp * 32, hipMemcpyDeviceToHost));
  int cnt = 0;
  for (int w = 0; w < ctx->g.ways; ++w) {
    if (keys[w] == 0 || (keys[w] & KEY_RANGE)) continue;  // range tallies are not per-slot tallies
    const bool done = keys[w] & KEY_DONE;
    if (rounds) rounds[cnt] = (int32_t)((keys[w] & KEY_ROUND_MASK) - 1u);
    if (states) states[cnt] = done ? 1 : 0;
    // a Done entry has dropped its Pending payload (ProxyLeader.scala:256)
    if (values) values[cnt] = done ? -1 : vals[w];
    if (vote_bits)
      for (int k = 0; k < 4; ++k) vote_bits[(size_t)cnt * 4 + k] = done ? 0ull : bits[w * 4 + k];
    ++cnt;
  }
  if (num_entries) *num_entries = cnt;
  return FPX_OK;
}

}  // extern "C"
