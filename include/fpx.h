/*
 * fpx.h -- C ABI of libfpx: the MI355X (gfx950) Phase-2 accept / quorum-tally engine.
 *
 * This is the drop-in boundary for ONE hot path of mwhittaker/frankenpaxos:
 *
 *   a1  multipaxos.Acceptor.handlePhase2a     shared/src/main/scala/frankenpaxos/multipaxos/Acceptor.scala:184-220
 *   a2  mencius.Acceptor.handlePhase2a        shared/src/main/scala/frankenpaxos/mencius/Acceptor.scala:202-235
 *   a3  multipaxos.ProxyLeader.handlePhase2b  shared/src/main/scala/frankenpaxos/multipaxos/ProxyLeader.scala:217-258
 *   a4  mencius.ProxyLeader.handlePhase2b     shared/src/main/scala/frankenpaxos/mencius/ProxyLeader.scala:305-353
 *   a5  quorums.*.isWriteQuorum / isSuperSetOfWriteQuorum
 *                                             shared/src/main/scala/frankenpaxos/quorums/{SimpleMajority,Grid,UnanimousWrites}.scala
 *   a6  multipaxos.ProxyLeader.handlePhase2a  shared/src/main/scala/frankenpaxos/multipaxos/ProxyLeader.scala:175-215
 *   a7  roundsystem.ClassicRoundRobin         shared/src/main/scala/frankenpaxos/roundsystem/RoundSystem.scala:60-87
 *   a8  simulator.FakeTransport (delivery)    shared/src/main/scala/frankenpaxos/FakeTransport.scala:89-159
 *
 * The reference has no FFI of its own (it is pure Scala); the seam is the Actor/Transport trait
 * surface.  A JNI shim marshals one event-loop tick worth of Phase2a / Phase2b
 * protobuf messages into the struct-of-arrays batches below and calls these entry points once per
 * batch.  Everything is plain pointers + sizes; no torch / HIP types appear in a signature (a HIP
 * stream crosses as void*).
 *
 * Conventions
 *   - Every function returns an int32 status (FPX_OK == 0).  FPX_EINVAL corresponds to a Scala
 *     require(...) failure; FPX_EFATAL_UNKNOWN_SLOTROUND to logger.fatal in
 *     ProxyLeader.handlePhase2b (ProxyLeader.scala:220-225).  A stale round is NOT an error: it
 *     produces a Nack exactly as in the reference.
 *   - A context is NOT thread-safe: one caller thread per context, mirroring "All Transport
 *     implementations MUST be single-threaded" (shared/src/main/scala/frankenpaxos/Transport.scala:37-39).
 *   - Entry points without a suffix take HOST pointers, are synchronous, and accept ANY batch: the
 *     library splits it into device "runs" so that the result equals message-at-a-time delivery in
 *     array order.  Entry points ending in _dev take DEVICE pointers (resident in HBM), enqueue on
 *     the context's stream and return immediately; the batch must already satisfy the run contract
 *     (below) -- a violation is detected on the device, nothing is applied, and the next fpx_sync()
 *     returns FPX_EORDER.
 *   - Run contract (one kernel launch): (1) slots in the batch are pairwise distinct; (2) in
 *     FPX_BALLOT_ACCEPTOR mode all messages addressed to one acceptor group carry the same round
 *     (different groups may be in different rounds).  Under (2) the acceptor's running-max `round`
 *     (Acceptor.scala:95,204) seen by message i equals max(round at batch start, round[i]) for every
 *     acceptor, which is what lets a batch be evaluated data-parallel and still be bit-exact.  A
 *     proxy leader's event-loop tick satisfies both in steady state; the host entry points cut any
 *     other batch at the offending message and launch the pieces back to back.
 *   - Bitmaps: one message's set of acceptors is 4 x uint64 (256 bits); bit j of the 256-bit
 *     little-endian integer is acceptor index j of the slot's acceptor group (for a Grid:
 *     j = row * grid_cols + col, row = groupIndex, col = acceptorIndex).
 *   - value_id is an int32 standing in for CommandBatchOrNoop (FPX_NOOP == -1 is Noop).
 */
#ifndef FPX_H
#define FPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPX_VERSION 100

#define FPX_MAX_REPLICAS 256
#define FPX_MASK_WORDS 4
#define FPX_NOOP (-1)

/* status codes.  What a non-OK status says about the state:
 *   FPX_EINVAL / FPX_EORDER   raised by the validation pass BEFORE anything of the batch is applied: the
 *                             batch -- and every later _dev call up to the next fpx_sync -- applied nothing.
 *   FPX_ECAPACITY             per message: that message's tally could not be opened, so it was neither
 *                             recorded nor forwarded to the acceptors (as if the proxy leader had never
 *                             received it); every other message of the batch was applied in full, acceptor
 *                             rounds and maxVotedSlot included.  The reference has no such limit
 *                             (ProxyLeader.states grows forever): free ways with fpx_proxy_forget or raise
 *                             tally_ways and resend the message.
 *   FPX_EFATAL_UNKNOWN_SLOTROUND  per message (the reference process would have died in logger.fatal): that
 *                             Phase2b was dropped, the others were applied.
 *   FPX_EHIP / FPX_ENOMEM     the HIP runtime failed; the context should be destroyed.
 * fpx_error_detail names the first offending message. */
enum {
  FPX_OK = 0,
  FPX_EINVAL = 1,                   /* == Scala require(...) / IllegalArgumentException            */
  FPX_EFATAL_UNKNOWN_SLOTROUND = 2, /* == logger.fatal, ProxyLeader.scala:220-225                    */
  FPX_EHIP = 3,                     /* a HIP runtime call failed (fpx_last_hip_error has the code)   */
  FPX_ENODEVICE = 4,                /* no gfx950 device / HIP runtime: there is NO CPU fallback      */
  FPX_ECAPACITY = 5,                /* more than tally_ways live (slot, round) tallies for one slot  */
  FPX_EORDER = 6,                   /* a _dev batch violated the run contract; nothing was applied   */
  FPX_ENOMEM = 7,
  FPX_ERCCL = 8,                    /* an RCCL call failed / RCCL is not loadable (fpx_last_rccl_error)  */
  FPX_EFATAL_PROTOCOL = 9           /* a logger.fatal / logger.check of the reference would have fired for a
                                       message (EPaxos: transitionToAcceptPhase on a committed instance or
                                       below an entry's ballot, Replica.scala:740-757): that message was not
                                       applied, the others were                                             */
};

typedef enum {
  FPX_Q_THRESHOLD = 0,       /* |X| >= f+1             ProxyLeader.scala:238 (non-flexible)          */
  FPX_Q_SIMPLE_MAJORITY = 1, /* |X| >= n/2+1           SimpleMajority.scala:30,41-49                 */
  FPX_Q_GRID = 2,            /* every row intersects X Grid.scala:43-50                              */
  FPX_Q_UNANIMOUS = 3        /* X == members           UnanimousWrites.scala:44-51                   */
} fpx_quorum_kind;

typedef enum {
  /* one promised `round` per acceptor, shared by all its slots: multipaxos/mencius Acceptor
   * (Acceptor.scala:95).  Faithful model. */
  FPX_BALLOT_ACCEPTOR = 0,
  /* one ballot per (slot, acceptor) cell, stored in HBM next to voteRound / voteValue: the
   * per-instance ballot of epaxos.Replica.handleAccept (epaxos/Replica.scala:1421-1510) and the
   * "generalised ballot[S x R]" model of SURVEY.md section 8(d). */
  FPX_BALLOT_PER_SLOT = 1
} fpx_ballot_mode;

/* flags */
#define FPX_F_TRUSTED 1u /* _dev entry points skip the run-contract validation pass (caller guarantees it) */
/* Hint: target masks are scattered subsets of the group -- the reference's thrifty default, a random f+1
 * of 2f+1 (ProxyLeader.scala:190-191).  K1 / K3 launches that carry target masks then write partially
 * voted 16-byte cells by read-modify-write instead of 4-byte stores.  Results are identical either way. */
#define FPX_F_SCATTERED_TARGETS 2u
/* Without that hint, on 256-cell rows of one acceptor group (R = 253 .. 256, threshold / majority / unanimous quorums),
 * launches that carry target masks run as TWO kernels: chunks of 32 messages that all go to a RUN of neighbouring
 * acceptors -- positions [start, start + len) of the row, cyclically, start a multiple of 16, len <= 128: what a proxy
 * leader sends that rotates a window of f + 1 acceptors over the group instead of shuffling (any f + 1 will do,
 * ProxyLeader.scala:190-191; jni/Native.scala's GpuProxyLeader does) -- of rows nobody voted in yet are walked two
 * rows per wavefront step (a row costs the same instructions whatever it moves, so half a row at a time ran at the
 * dense rate: 3.0e9 slots/s; now 4.4e9), every other chunk by the row-at-a-time walk behind it.  The hint skips the
 * first kernel.  Results are identical either way (FPX_NO_PACKED_RUNS in the environment switches the first kernel off). */
/* Mencius contexts (num_leader_groups L > 1, num_slots a multiple of L, num_replicas <= 32) keep the per-slot rows of the cell arrays and
 * tally tables LEADER-GROUP-MAJOR in HBM -- slot s lives in row (s % L) * (S / L) + s / L -- so that what one leader
 * group does (a noop range = every L-th slot; its batch of Phase2as in slot order) touches neighbouring rows and
 * every 128-byte line leaves the GPU whole.  A launch is fastest as the leader groups' batches back to back; one batch
 * in slot order across the leader groups is walked column by column by the kernel and is still faster than on
 * slot-ordered rows (profiles/r03_cfg5.md).  This flag keeps rows in slot order (round 2's layout).  Results are
 * identical either way; only speed differs. */
#define FPX_F_SLOT_MAJOR_ROWS 4u

typedef struct {
  int32_t num_slots;         /* S: log window held in HBM; slots are 0 .. S-1                          */
  int32_t num_replicas;      /* R: acceptors per acceptor group, 1 .. 256                               */
  int32_t num_groups;        /* acceptor groups per leader group; slot -> group as below                */
  int32_t num_leader_groups; /* 1 for MultiPaxos.  Mencius: leader group = slot % num_leader_groups,
                                acceptor group = (slot / num_leader_groups) % num_groups
                                (mencius/ProxyLeader.scala:231-234); group id = lg * num_groups + ag     */
  int32_t f;                 /* FPX_Q_THRESHOLD: quorum size f+1                                         */
  int32_t quorum_kind;       /* fpx_quorum_kind                                                          */
  int32_t grid_rows, grid_cols; /* FPX_Q_GRID: rows * cols == num_replicas                               */
  int32_t num_leaders;       /* ClassicRoundRobin(num_leaders): Nack routing, Acceptor.scala:197         */
  int32_t ballot_mode;       /* fpx_ballot_mode                                                          */
  int32_t tally_ways;        /* live (slot, round) tallies kept per slot, 1 .. 8 (ProxyLeader.states is
                                keyed by SlotRound, ProxyLeader.scala:87,135)                            */
  int32_t replica_base;      /* multi-GPU replica-axis sharding: this context owns acceptors
                                [replica_base, replica_base + num_replicas) of a group of
                                replicas_total acceptors; multiple of 4; 0 when not sharded             */
  int32_t replicas_total;    /* 0 => num_replicas.  Quorum predicates are over replicas_total.          */
  int32_t device;            /* HIP device ordinal.  Every entry point makes it current for the duration
                                of the call and restores the caller's device, so contexts on several GPUs
                                can be driven from one thread                                              */
  uint32_t flags;            /* FPX_F_*                                                                  */
} fpx_config;

typedef struct fpx_ctx fpx_ctx;

/* ---- lifecycle ------------------------------------------------------------------------------- */
int32_t fpx_version(void);
const char* fpx_strerror(int32_t status);
/* validates the config exactly like Config.checkValid-style require()s (FPX_EINVAL), no GPU needed */
int32_t fpx_config_check(const fpx_config* cfg);
/* allocates HBM state: acceptors start with round = -1 and no votes (Acceptor.scala:95-104),
 * the proxy leader with no tallies (ProxyLeader.scala:135). */
int32_t fpx_create(const fpx_config* cfg, fpx_ctx** out);
int32_t fpx_destroy(fpx_ctx* ctx);
/* re-initialises all state to the freshly-created state (device memset, async on the stream) */
int32_t fpx_reset(fpx_ctx* ctx);
/* hip_stream is a hipStream_t passed through as it is: NULL is the device's default (null) stream --
 * which is what torch.cuda.current_stream().cuda_stream is unless the caller switched streams --
 * and FPX_STREAM_OWN selects the context's private stream (the state after fpx_create).  The _dev
 * entry points enqueue on this stream: device inputs must be produced on it (or ordered before it). */
#define FPX_STREAM_OWN ((void*)(intptr_t)-1)
int32_t fpx_set_stream(fpx_ctx* ctx, void* hip_stream);
/* waits for the stream; returns the sticky device status of the _dev calls since the last sync
 * (FPX_OK, FPX_EORDER, FPX_ECAPACITY, FPX_EFATAL_UNKNOWN_SLOTROUND, FPX_EINVAL) and clears it */
int32_t fpx_sync(fpx_ctx* ctx);
/* index / slot / round of the first offending message of the last non-OK status */
int32_t fpx_error_detail(fpx_ctx* ctx, int32_t* index, int32_t* slot, int32_t* round);
int32_t fpx_last_hip_error(fpx_ctx* ctx);
/* HBM bytes held by the context */
int64_t fpx_device_bytes(fpx_ctx* ctx);
/* How the cell arrays (Acceptor.states of every acceptor, multipaxos/Acceptor.scala:98) were placed in HBM by fpx_create:
 * out[0] = 1 if the slab is built from 1 GiB physical chunks paired by measurement (vote_round's and vote_value's chunk of
 * a gigabyte of rows must be a pair the chip writes side by side at the fast rate; contexts of 2 GiB and more with groups
 * of 61+ acceptors), 0 if it is one allocation; out[1] = gigabyte windows probed; out[2..4] = min / median / max over the
 * windows of the hot access pattern's time on half a window, in ms (0 when nothing was probed).  Diagnostics only:
 * results never depend on the placement.  FPX_PLACEMENT_CHUNKS=0 in the environment keeps one allocation. */
int32_t fpx_placement_stats(fpx_ctx* ctx, float out[5]);
/* What the placement search of fpx_create cost: probes run, decisions taken WITHOUT a probe because the search's budget
 * was spent, and the wall clock of the search in ms.  The budget is FPX_PLACEMENT_BUDGET_MS in the environment (default
 * 300 ms; 0 = no probing at all: the slab is built from the chunks in allocation order): once it is spent every remaining
 * decision takes its first candidate, so N ranks creating contexts at once cannot stretch fpx_create without bound.  Any
 * of the three pointers may be NULL.  Diagnostics only. */
int32_t fpx_placement_search(fpx_ctx* ctx, int32_t* probes, int32_t* unprobed, float* ms);
/* the configuration the context was created with (replicas_total filled in): a binding sizes its buffers from the
 * handle, not from what its caller says the handle is */
int32_t fpx_get_config(fpx_ctx* ctx, fpx_config* out);
/* Page-locked host memory for the host-pointer entry points: batches that live in it cross PCIe by DMA at
 * link rate instead of through the runtime's pageable staging copies.  A JVM caller wraps the region in
 * a direct ByteBuffer (JNI NewDirectByteBuffer), the counterpart of the Netty direct buffers the
 * reference's transports hand to the handlers.  Any other host pointer stays legal, only slower. */
int32_t fpx_host_alloc(int64_t bytes, void** out);
int32_t fpx_host_free(void* p);
/* Kernel timing for roofline accounting: while enabled, every K1 / K3 call brackets its dominant
 * kernel (k_phase2) with HIP events on the context's stream.  fpx_profile_read waits for the stream
 * and returns the number of bracketed launches since the last read and the sum of their durations. */
int32_t fpx_profile_enable(fpx_ctx* ctx, int32_t on);
int32_t fpx_profile_read(fpx_ctx* ctx, int32_t* launches, double* total_ms);
/* the same read launch by launch: the durations of the first `cap` timed launches since the last read, in order, into
 * ms_out; *launches = how many there were (the variance of one kernel over the log's windows: profiles/r05_placement.md) */
int32_t fpx_profile_read_launches(fpx_ctx* ctx, int32_t cap, float* ms_out, int32_t* launches);

/* ---- a7: roundsystem.ClassicRoundRobin (RoundSystem.scala:60-87); pure host scalars ----------- */
int32_t fpx_round_leader(int32_t num_leaders, int32_t round);                         /* :63     */
int32_t fpx_next_classic_round(int32_t num_leaders, int32_t leader_index, int32_t round); /* :66-81 */

/* ---- a5: quorum predicates on the device ----------------------------------------------------- */
/* Evaluates isWriteQuorum (strict = 1: FPX_EINVAL if a bit outside the member set is present, the
 * require() of SimpleMajority.scala:42 / Grid.scala:44 / UnanimousWrites.scala:45) or
 * isSuperSetOfWriteQuorum (strict = 0: foreign bits ignored) for n node sets; nodes is n x 4 words,
 * out is n bytes (host pointers).  Only the quorum fields of cfg are read.  Runs the same device
 * function the tally kernels use. */
int32_t fpx_quorum_eval(const fpx_config* cfg, int32_t n, const uint64_t* nodes, int32_t strict,
                        uint8_t* out);
int32_t fpx_is_write_quorum(const fpx_config* cfg, const uint64_t nodes[4], int32_t strict,
                            uint8_t* out);
/* isReadQuorum / isSuperSetOfReadQuorum with the same conventions (SimpleMajority.scala:41-47,
 * Grid.scala:36-41,52-53, UnanimousWrites.scala:36-42,53-54) */
int32_t fpx_read_quorum_eval(const fpx_config* cfg, int32_t n, const uint64_t* nodes, int32_t strict,
                             uint8_t* out);

/* ---- a1/a2 (K1): acceptors handle Phase2a -------------------------------------------------------
 * Delivers Phase2a(slot[i], round[i], value_id[i]), i = 0..n-1 in array order, to the acceptors of
 * the slot's group selected by target_mask (n x 4 words; NULL = every acceptor of the group).  Per
 * targeted acceptor (Acceptor.scala:192-219): round[i] < its round -> Nack(its round); otherwise
 * its round := round[i], states[slot] := (round[i], value_id[i]), maxVotedSlot := max(.., slot),
 * Phase2b.  Outputs (each may be NULL): vote_bits n x 4 words = acceptors that sent Phase2b;
 * nack_bits n x 4 words = acceptors that sent Nack; nack_round n ints = the largest round carried
 * by a Nack for message i, -1 if none (the only field Leader.handleNack, Leader.scala:672-697,
 * reacts to; destination leader = fpx_round_leader(num_leaders, round[i])). */
int32_t fpx_acceptor_phase2a(fpx_ctx* ctx, int32_t n, const int32_t* slot, const int32_t* round,
                             const int32_t* value_id, const uint64_t* target_mask,
                             uint64_t* vote_bits, uint64_t* nack_bits, int32_t* nack_round);
int32_t fpx_acceptor_phase2a_dev(fpx_ctx* ctx, int32_t n, const int32_t* d_slot,
                                 const int32_t* d_round, const int32_t* d_value_id,
                                 const uint64_t* d_target_mask, uint64_t* d_vote_bits,
                                 uint64_t* d_nack_bits, int32_t* d_nack_round);

/* Phase1a (Acceptor.scala:148-182), needed so that rounds can move: acceptors of `group` selected
 * by target_mask[4] (NULL = all) with round > their round promise it (round := round), the others
 * Nack.  In FPX_BALLOT_PER_SLOT mode the promise applies to every cell of the group with
 * slot >= chosen_watermark.  promised_bits / nack_bits: 4 words each (may be NULL).
 * NOTE multipaxos uses `phase1a.round < round -> Nack` (Acceptor.scala:155), i.e. an EQUAL round is
 * promised again; that is what is implemented. */
int32_t fpx_acceptor_phase1a(fpx_ctx* ctx, int32_t group, int32_t round, int32_t chosen_watermark,
                             const uint64_t* target_mask, uint64_t* promised_bits,
                             uint64_t* nack_bits);
/* the same on device-resident arguments (each 4 words in HBM, 8-byte aligned, may be NULL), asynchronous on the context's
 * stream: a leader change in the middle of a device-resident stream costs no host round trip.  With a ballot per cell it is
 * ONE launch that writes the reply bits straight into the caller's words and does not wait for the fold of the fused step
 * before it (fpx_deferred_folds): a vote launch leaves a bound on what its fold can raise, the decision uses that
 * (csrc/fpx_kernels.hpp, k_p1a_fast) -- unless the watermark is above the smallest one handed in since the promises were
 * last flushed, which takes the three launches of rounds 2 - 5 (as does FPX_P1A_SPLIT=1 in the environment). */
int32_t fpx_acceptor_phase1a_dev(fpx_ctx* ctx, int32_t group, int32_t round, int32_t chosen_watermark,
                                 const uint64_t* d_target_mask, uint64_t* d_promised_bits,
                                 uint64_t* d_nack_bits);
/* FPX_BALLOT_PER_SLOT keeps a Phase1a that no cell is ahead of as ONE record per acceptor ("every cell from
 * the watermark on is at least `round`") instead of rewriting S x R ballots, and honours the records in the
 * vote kernels; results are identical.  This writes all outstanding records into the cells (one sweep over the
 * ballot array) and drops them, which puts the vote kernels back into their leanest form -- worth calling once
 * after the last Phase1a before a long steady stretch.  Readback and digests do it implicitly.  Asynchronous. */
int32_t fpx_acceptor_flush_promises(fpx_ctx* ctx);

/* ---- a6: ProxyLeader.handlePhase2a bookkeeping ---------------------------------------------------
 * Opens the tally Pending(phase2a, {}) for (slot[i], round[i]) (ProxyLeader.scala:213).  A (slot,
 * round) that is already known is ignored (:177-184) and reported as is_new[i] = 0.  The reference
 * picks a thrifty random quorum here (:190-196, unseeded RNG, F12 in SURVEY.md); which acceptors a
 * Phase2a goes to is the caller's target_mask in fpx_acceptor_phase2a. */
int32_t fpx_proxy_open(fpx_ctx* ctx, int32_t n, const int32_t* slot, const int32_t* round,
                       const int32_t* value_id, uint8_t* is_new);
int32_t fpx_proxy_open_dev(fpx_ctx* ctx, int32_t n, const int32_t* d_slot, const int32_t* d_round,
                           const int32_t* d_value_id, uint8_t* d_is_new);

/* ---- a3/a4 (K2): ProxyLeader.handlePhase2b ----------------------------------------------------------
 * Message i carries the Phase2b's of the acceptors in vote_bits[i] for (slot[i], round[i]) (an
 * all-zero row is "no message").  Unknown (slot, round) -> FPX_EFATAL_UNKNOWN_SLOTROUND (:220-225);
 * Done -> ignored (:227-232); Pending -> votes are recorded keyed by acceptor (a duplicate vote
 * changes nothing, :237) and, when the quorum predicate holds (:238-243), Chosen(slot,
 * pending.phase2a.value) is emitted exactly once (:246-256): newly_chosen[i] = 1,
 * chosen_round[i] = round[i], chosen_value[i] = the value given to fpx_proxy_open; otherwise
 * newly_chosen[i] = 0, chosen_round[i] = chosen_value[i] = -1. */
int32_t fpx_proxy_phase2b(fpx_ctx* ctx, int32_t n, const int32_t* slot, const int32_t* round,
                          const uint64_t* vote_bits, uint8_t* newly_chosen, int32_t* chosen_round,
                          int32_t* chosen_value);
int32_t fpx_proxy_phase2b_dev(fpx_ctx* ctx, int32_t n, const int32_t* d_slot, const int32_t* d_round,
                              const uint64_t* d_vote_bits, uint8_t* d_newly_chosen,
                              int32_t* d_chosen_round, int32_t* d_chosen_value);

/* Garbage collection of the proxy leader (NOT in the reference, whose ProxyLeader.states grows forever,
 * ProxyLeader.scala:135): forgets every tally -- Pending or Done -- of the slots
 * [first_slot, first_slot + count) and every noop-range tally whose range lies inside that window, so that a long-running simulation can re-propose a chosen-and-executed
 * window of the log in further rounds without exhausting tally_ways.  Acceptor state is untouched.
 * Asynchronous on the context's stream.  After it, a Phase2b for a forgotten (slot, round) is "unknown". */
int32_t fpx_proxy_forget(fpx_ctx* ctx, int32_t first_slot, int32_t count);

/* Recycling of log-window rows (NOT in the reference, whose Acceptor.states and ProxyLeader.states grow forever,
 * multipaxos/Acceptor.scala:98, ProxyLeader.scala:135): the rows [first_slot, first_slot + count) of the window
 * become fresh -- every acceptor's vote in them is dropped (voteRound = voteValue = -1, as if `states` had no
 * entry) and their tallies are forgotten (fpx_proxy_forget).  What an acceptor PROMISED stays: its scalar round
 * (FPX_BALLOT_ACCEPTOR) is untouched, and in FPX_BALLOT_PER_SLOT mode the cells keep their ballots, so a recycled
 * row never accepts a round the old row would have refused.  maxVotedSlot is left as it is (an upper bound).  A
 * host that maps an unbounded log onto the window as row = slot % num_slots calls this for the rows whose slots are
 * chosen and no longer needed before it lets the log wrap onto them (frankenpaxos_amd/jni/Native.scala, GpuPhase2).
 * Asynchronous on the context's stream. */
int32_t fpx_recycle_slots(fpx_ctx* ctx, int32_t first_slot, int32_t count);

/* ---- K3: fused step = fpx_proxy_open + fpx_acceptor_phase2a + fpx_proxy_phase2b ------------------
 * For each message in order: open (slot, round) (duplicates are ignored and NOT forwarded to the
 * acceptors, :177-184), deliver the Phase2a to the targeted acceptors (target_mask NULL = all: the
 * dense schedule of SURVEY.md section 8(d)), feed their Phase2b's to the tally.  The vote bitmap never
 * leaves the chip.  nack_round may be NULL. */
int32_t fpx_phase2_fused(fpx_ctx* ctx, int32_t n, const int32_t* slot, const int32_t* round,
                         const int32_t* value_id, const uint64_t* target_mask, uint8_t* chosen,
                         int32_t* chosen_round, int32_t* chosen_value, int32_t* nack_round);
/* The same, asynchronous, for batches in PAGE-LOCKED host memory (fpx_host_alloc or any memory mapped into the GPU's
 * address space -- FPX_EINVAL otherwise): submit returns at once with a ticket, wait(ticket) blocks until that call's
 * outputs are in the caller's arrays and returns its status.  Up to 3 calls may be in flight (FPX_ECAPACITY: wait for
 * the oldest first); they execute in submission order.  The inputs go up by the copy engine on a stream of their own
 * into staging buffers in HBM; validation and the fused step follow on the context's stream; the vote kernel writes the
 * Chosen records (and Nack rounds) STRAIGHT into the caller's page-locked arrays -- there is no copy down.  With calls
 * submitted back to back the upload of call k + 1 hides behind the fused step of call k, which runs ~3 % longer for its
 * posted writes over PCIe (profiles/r06_host_path.md has what every other way of crossing PCIe cost the vote kernel);
 * see profiles/r06_host_path.md for the rate per 2^20 x 256 call.  A call must be ONE device run (the run
 * contract above): a violation is FPX_EORDER from wait with nothing applied -- pass that batch to fpx_phase2_fused,
 * which cuts it into runs.  After an error the calls queued behind the failed one have applied nothing either and
 * report the same status.  fpx_phase2_fused itself uses submit + wait when it is handed page-locked arrays. */
int32_t fpx_phase2_fused_submit(fpx_ctx* ctx, int32_t n, const int32_t* slot, const int32_t* round,
                                const int32_t* value_id, const uint64_t* target_mask, uint8_t* chosen,
                                int32_t* chosen_round, int32_t* chosen_value, int32_t* nack_round,
                                int32_t* ticket);
int32_t fpx_phase2_fused_wait(fpx_ctx* ctx, int32_t ticket);
int32_t fpx_phase2_fused_dev(fpx_ctx* ctx, int32_t n, const int32_t* d_slot, const int32_t* d_round,
                             const int32_t* d_value_id, const uint64_t* d_target_mask,
                             uint8_t* d_chosen, int32_t* d_chosen_round, int32_t* d_chosen_value,
                             int32_t* d_nack_round);

/* ---- a2/a4 (K4): Mencius noop ranges -----------------------------------------------------------------
 * Needs FPX_BALLOT_ACCEPTOR.  Rounds must be < 2^30 - 1 everywhere (one key bit marks the per-slot shadow of
 * a length-1 range).  A leader with nothing to propose skips a stretch of its slots with ONE message per
 * stretch; with hundreds of leader groups a tick carries hundreds of them, so every entry point takes n
 * ranges (slot_start[i], slot_end[i], round[i]), processed as if delivered in array order.  Bitmaps are
 * num_groups x 4 words PER RANGE (bit = acceptor index within its acceptor group), so a batch's are
 * n x num_groups x 4.
 *
 * mencius.Acceptor.handlePhase2aNoopRange (mencius/Acceptor.scala:237-291) delivered to the acceptors of
 * EVERY acceptor group of the leader group that owns slot_start (the proxy leader relays a range to each
 * group, mencius/ProxyLeader.scala:276-291), selected by target_masks (NULL = all).  An acceptor with
 * round > `round` Nacks; otherwise round := `round` and every slot of [slot_start, slot_end) owned by its
 * acceptor group (slot = slot_start + k * num_leader_groups with (slot / num_leader_groups) % num_groups ==
 * its group) votes (round, Noop).  Outputs (may be NULL): vote_bits / nack_bits, nack_round[i] = largest round
 * carried by a Nack for range i or -1. */
int32_t fpx_acceptor_phase2a_noop_ranges(fpx_ctx* ctx, int32_t n, const int32_t* slot_start,
                                         const int32_t* slot_end, const int32_t* round,
                                         const uint64_t* target_masks, uint64_t* vote_bits,
                                         uint64_t* nack_bits, int32_t* nack_round);
/* mencius.ProxyLeader.handlePhase2aNoopRange bookkeeping (mencius/ProxyLeader.scala:255-303): opens
 * PendingPhase2aNoopRange for (slot_start, slot_end, round); a known key -- also an earlier message of the
 * same batch -- is ignored (is_new = 0).  The tallies live in a hash table of the context (capacity: a few
 * hundred ranges in flight per leader group; FPX_ECAPACITY beyond it); fpx_proxy_forget reclaims those of
 * a slot window, Pending and Done alike. */
int32_t fpx_proxy_open_noop_ranges(fpx_ctx* ctx, int32_t n, const int32_t* slot_start, const int32_t* slot_end,
                                   const int32_t* round, uint8_t* is_new);
/* mencius.ProxyLeader.handlePhase2bNoopRange (mencius/ProxyLeader.scala:355-411).  Unknown key ->
 * FPX_EFATAL_UNKNOWN_SLOTROUND; Done or a single-slot tally under the same key -> ignored; ChosenNoopRange
 * (newly_chosen = 1) once every acceptor group has f+1 votes. */
int32_t fpx_proxy_phase2b_noop_ranges(fpx_ctx* ctx, int32_t n, const int32_t* slot_start, const int32_t* slot_end,
                                      const int32_t* round, const uint64_t* vote_bits, uint8_t* newly_chosen);
/* The fused step for ranges = open + acceptors + tally (the K3 of noop ranges): a range that is already
 * known is neither forwarded nor tallied again.  Outputs may be NULL.  The _dev form takes device pointers,
 * enqueues on the context's stream and needs one round per leader group within the batch (the run contract;
 * FPX_EORDER otherwise, nothing applied); the host form cuts any batch into such runs itself. */
int32_t fpx_noop_ranges_fused(fpx_ctx* ctx, int32_t n, const int32_t* slot_start, const int32_t* slot_end,
                              const int32_t* round, const uint64_t* target_masks, uint64_t* vote_bits,
                              uint64_t* nack_bits, int32_t* nack_round, uint8_t* is_new, uint8_t* chosen);
int32_t fpx_noop_ranges_fused_dev(fpx_ctx* ctx, int32_t n, const int32_t* d_slot_start,
                                  const int32_t* d_slot_end, const int32_t* d_round,
                                  const uint64_t* d_target_masks, uint64_t* d_vote_bits, uint64_t* d_nack_bits,
                                  int32_t* d_nack_round, uint8_t* d_is_new, uint8_t* d_chosen);
/* One step of a Mencius proxy leader: the Phase2as of the leader groups that have commands AND the Phase2aNoopRanges
 * of the leader groups that skip their slots (mencius/ProxyLeader.scala:231-234 slot -> leader group; :216-303 the two
 * handlers) = fpx_phase2_fused_dev(the first ten arguments) followed by fpx_noop_ranges_fused_dev(the next ten), with
 * exactly their outputs and errors.  independent != 0: the caller states that no leader group has both a command and a
 * range in this step (a leader either proposes in its slots or skips them).  Then the two halves touch disjoint rows,
 * tallies and acceptors, no order between them is observable, and under FPX_F_TRUSTED the step is TWO launches instead
 * of four where its shape allows (groups of at most 32 acceptors on leader-group-major rows, commands delivered to every
 * acceptor -- d_target_mask NULL --, more than 512 commands, a range chain that fits the vote kernel's LDS): the vote
 * kernel with the ranges' chain of dependent steps as its first workgroup, then the ranges' fill with the vote kernel's
 * fold of maxima as further rows of its grid (profiles/r05_cfg5.md); other shapes run the halves side by side, the
 * ranges on a second stream of the context between a fork and a join event.  A context that validates its batches
 * checks the statement first: a leader group with both makes the step FPX_EORDER with nothing applied (call again with
 * independent = 0), and the halves run one after the other.  FPX_BAND_SERIAL=1 in the environment keeps the two-launch
 * form off.  HAZARD: under FPX_F_TRUSTED the statement is NOT checked -- a leader group with both a command and a range
 * in an `independent` step makes the range chain's store of an acceptor's round race with the vote kernel's fold of
 * maxima, and the state is silently wrong.  FPX_DEBUG_CHECKS=1 in the environment checks the statement on trusted contexts too
 * (FPX_EORDER, nothing applied; the halves then run one after the other). */
int32_t fpx_mencius_band_fused_dev(fpx_ctx* ctx, int32_t n, const int32_t* d_slot, const int32_t* d_round,
                                   const int32_t* d_value_id, const uint64_t* d_target_mask, uint8_t* d_chosen,
                                   int32_t* d_chosen_round, int32_t* d_chosen_value, int32_t* d_nack_round,
                                   int32_t n_ranges, const int32_t* d_slot_start, const int32_t* d_slot_end,
                                   const int32_t* d_range_round, const uint64_t* d_range_target_masks,
                                   uint64_t* d_range_vote_bits, uint64_t* d_range_nack_bits,
                                   int32_t* d_range_nack_round, uint8_t* d_range_is_new, uint8_t* d_range_chosen,
                                   int32_t independent);
/* diagnostic: the fused steps since fpx_create whose launch carried the fold of the step before in its own grid.  Every K1 /
 * K3 launch is followed by a fold of the maxima its workgroups left (Acceptor.round / maxVotedSlot, multipaxos/Acceptor.scala:
 * 204-209, as scalars per acceptor); with a ballot per cell (FPX_BALLOT_PER_SLOT) no vote kernel reads those scalars, so the
 * fold of one fpx_phase2_fused_dev call waits for the next one and rides in its launch (or is launched by whatever entry
 * point other than a Phase1a touches the context first: the deferral is not observable through this ABI).  FPX_NO_DEFER_FINALIZE=1 in the
 * environment launches every fold at once, as rounds 1 - 5 did. */
int64_t fpx_deferred_folds(fpx_ctx* ctx);
/* diagnostic: the steps of fpx_mencius_band_fused_dev that ran in the two-launch form since fpx_create */
int64_t fpx_band_merged_steps(fpx_ctx* ctx);
/* batches of one (bitmaps num_groups x 4 words) */
int32_t fpx_acceptor_phase2a_noop_range(fpx_ctx* ctx, int32_t slot_start, int32_t slot_end,
                                        int32_t round, const uint64_t* target_masks,
                                        uint64_t* vote_bits, uint64_t* nack_bits, int32_t* nack_round);
int32_t fpx_proxy_open_noop_range(fpx_ctx* ctx, int32_t slot_start, int32_t slot_end, int32_t round,
                                  uint8_t* is_new);
int32_t fpx_proxy_phase2b_noop_range(fpx_ctx* ctx, int32_t slot_start, int32_t slot_end, int32_t round,
                                     const uint64_t* vote_bits, uint8_t* newly_chosen);
/* readback (parity): state 0 = unknown key, 1 = Pending, 2 = Done; vote_bits num_groups x 4 words (zero
 * unless Pending) */
int32_t fpx_read_range_tally(fpx_ctx* ctx, int32_t slot_start, int32_t slot_end, int32_t round, int32_t* state,
                             uint64_t* vote_bits);

/* ---- a9 (K5): EPaxos pre-accept fast path -------------------------------------------------------------
 * One tick of FRESH instances through the pre-accept phase of n = 2f+1 EPaxos replicas with the
 * key-value store's top-one conflict index (epaxos/Replica.scala:569-600, 633-729, 1159-1419;
 * statemachine/KeyValueStore.scala:225-302; util/TopOne.scala; Util.scala:19-21).
 *
 * Message i: instance (leader[i], number[i]) with a single-key command on key[i] (is_set[i] = 1:
 * SetRequest, 0: GetRequest); the leader sends PreAccept to the n-2 other replicas in resp_mask[i]
 * (thrifty fast quorum, Replica.scala:705-706).  rank is n x m: rank[r * m + i] = the position of
 * message i in replica r's processing order (a permutation of 0..m-1 per replica; only replicas that
 * take part in message i -- its leader and resp_mask[i] -- matter; a rank row that is not a permutation
 * is FPX_EINVAL and nothing is applied; values outside 0..m-1 are always caught, and a row whose values are in
 * range but repeat is told from a permutation by comparing two independent 64-bit additive fingerprints of its
 * multiset of values with those of 0..m-1 -- no scatter / gather of the row is needed for it).  Every participating replica
 * computes the command's conflicts against ITS conflict index in ITS order (getTopOneConflicts), the
 * leader's become the PreAccept's dependencies, the others answer PreAcceptOk with the union; the
 * leader takes the fast path iff the n-2 answers are identical (popularItems(..., n-2)), otherwise
 * the slow path proposes the union of all answers (preAcceptingSlowPath, :796-813).  After the tick
 * the commits reach every replica's conflict index (commit -> updateConflictIndex, :815-828).
 * seen_mask (NULL = resp_mask, the thrifty deployment): the OTHER replicas that receive and process the
 * PreAccept at all.  With the reference's default ThriftySystem.NotThrifty (Replica.scala:83, 556-562) that is
 * every other replica (n-1 of them): all of them compute conflicts and update their index in their order,
 * and the leader decides on the first n-2 answers to arrive (fastQuorumSize responses including its own,
 * :1376) -- resp_mask, a subset of seen_mask.
 * Outputs (may be NULL): fast[i]; deps = the committed (fast) or Accept-phase (slow) dependencies and
 * leader_deps = the PreAccept's, as InstancePrefixSets (epaxos/InstancePrefixSet.scala): deps[i * n + l] is
 * the IntPrefixSet watermark of leader l's column (every instance of l below it is a dependency).  The
 * reference removes the instance itself from its dependencies (dependencies.subtractOne(instance),
 * Replica.scala:582; compact/IntPrefixSet.scala:388-398), which only ever changes the column of the
 * instance's OWN leader: when a replica already holds a higher-numbered conflicting instance of that leader
 * (it processed (L, 5) before (L, 4)), that column's watermark drops to number[i] and the ids above it become
 * the set's explicit `values` -- always the run number[i] + 1 .. end - 1, reported as
 * own_values_end[i * 2 + 0] (deps) and own_values_end[i * 2 + 1] (leader_deps); 0 = no explicit values (the
 * case on every FIFO channel).  Sequence numbers are the constant 0 the reference uses with top-k
 * dependencies (:575-578). */
typedef struct fpx_epx fpx_epx;
typedef struct {
  int32_t num_replicas; /* n: 3, 5 or 7 */
  int32_t num_keys;
  int32_t device;
  uint32_t flags;
  int32_t num_instances; /* > 0: every replica keeps its command log (Replica.cmdLog) for the instances
                            (leader, number < num_instances): fpx_epx_preaccept records and checks it,
                            fpx_epx_prepare / fpx_epx_accept run on it.  0: no command log (pre-accept only) */
} fpx_epx_config;
int32_t fpx_epx_create(const fpx_epx_config* cfg, fpx_epx** out);
int32_t fpx_epx_destroy(fpx_epx* epx);
/* n, num_keys, num_instances of the context (any pointer may be NULL) */
int32_t fpx_epx_info(fpx_epx* epx, int32_t* num_replicas, int32_t* num_keys, int32_t* num_instances);
int32_t fpx_epx_set_stream(fpx_epx* epx, void* hip_stream);
int32_t fpx_epx_preaccept(fpx_epx* epx, int32_t m, const int32_t* leader, const int32_t* number,
                          const int32_t* key, const uint8_t* is_set, const uint8_t* resp_mask,
                          const uint8_t* seen_mask, const int32_t* rank, const int32_t* triple_id,
                          uint8_t* fast, int32_t* deps, int32_t* leader_deps, int32_t* own_values_end);
/* device-resident inputs / outputs; the work is enqueued on the context's stream and fpx_epx_sync returns the sticky
 * status.  One host wait is inside: for ticks of at most 2048 keys the library partitions the tick by key and runs
 * one kernel per key group on chip, which needs every key's commands to fit the on-chip tables (1152 per key at
 * n = 5); whether they do is known once the tick's key histogram is -- a few microseconds of device work after the
 * stream reaches this call -- and is read from a page-locked word while the next kernels (already enqueued) run.  The
 * call therefore returns when the stream has reached the tick's first two small kernels, not before; a tick with a
 * hotter key is then enqueued again in the general form (radix sort of all (key, message) pairs).  Not capturable
 * into a HIP graph.  FPX_EPX_V1 in the environment at fpx_epx_create selects the general form always.
 * Any alignment of the input arrays is accepted; arrays that all start at 16-byte boundaries, with m % 4 == 0, are read
 * with 16-byte loads (~1 % of a 2^20-command tick). */
int32_t fpx_epx_preaccept_dev(fpx_epx* epx, int32_t m, const int32_t* d_leader, const int32_t* d_number,
                              const int32_t* d_key, const uint8_t* d_is_set, const uint8_t* d_resp_mask,
                              const uint8_t* d_seen_mask, const int32_t* d_rank, const int32_t* d_triple_id,
                              uint8_t* d_fast, int32_t* d_deps, int32_t* d_leader_deps,
                              int32_t* d_own_values_end);
/* The same with ONE packed line per command instead of the four output arrays (what the kernel that decides a key
 * writes most cheaply: a command's outputs are contiguous and sector-aligned, where the four arrays take four
 * partial sectors at every message index).  d_packed: m x fpx_epx_packed_stride(n) ints; line i =
 *   [0, n) deps   [n, 2n) leader_deps   [2n] own_values_end (deps)   [2n + 1] own_values_end (leader_deps)
 *   [2n + 2] fast (0 / 1)   the rest 0            (stride = 12 / 16 / 20 ints for n = 3 / 5 / 7) */
int32_t fpx_epx_packed_stride(int32_t num_replicas);
int32_t fpx_epx_preaccept_packed_dev(fpx_epx* epx, int32_t m, const int32_t* d_leader, const int32_t* d_number,
                                     const int32_t* d_key, const uint8_t* d_is_set, const uint8_t* d_resp_mask,
                                     const uint8_t* d_seen_mask, const int32_t* d_rank, const int32_t* d_triple_id,
                                     int32_t* d_packed);
int32_t fpx_epx_sync(fpx_epx* epx);
/* Dependency-graph execution of one tick's commits ON THE DEVICE (the general, host-side graph is
 * include/fpx_depgraph.h): what Replica.execute does with the committed triples (epaxos/Replica.scala:859-917,
 * depgraph/TarjanDependencyGraph.scala:225-276) -- strongly connected components of the committed instances, components
 * in reverse topological order, inside a component by (leader, id) (sequence numbers are 0 with top-k dependencies).
 * Message i = instance (leader[i], number[i]) with the dependencies of line i of d_packed (fpx_epx_preaccept_packed_dev's
 * output: deps watermarks + the explicit ids of the own column); d_committed (may be NULL = all) 0 = not committed yet:
 * it and whatever reaches it wait.  The columns must be DENSE: leader l's instances first[l] .. first[l] + count[l] - 1
 * each exactly once (sum of count = m), everything below first[l] executed earlier (FPX_EINVAL otherwise).
 * Outputs: d_order[p] = the message executed p-th, d_component[p] = its component's number (consecutive from 0; equal
 * numbers = one strongly connected component), for p < *num_executed; *num_components.  Where the reference leaves the
 * order open (components that do not depend on each other) this path takes its own; the SET of components and the
 * validity of the order equal fpx_depgraph's.  *needs_host_path != 0: the members of a strongly connected component could
 * not be made neighbours of the order -- two DIFFERENT closures of vertices on cycles have the same closure sum AND the same
 * 22-bit closure hash, so their members may interleave (detected where component starts are found; components of any size
 * are handled on the device, this is a hash collision, probability ~2^-22 per pair of such closures), or a closure sum of
 * 2^29 and more (not reachable with m < 2^21) -- the outputs are not valid, run the tick through fpx_depgraph_commit_epx.
 * Device pointers, the context's stream; ONE host wait on a page-locked word at the end (round 4: one per closure round),
 * so the call returns when the order is on the device: not capturable into a HIP graph.  n <= 5 with columns of fewer
 * than 2^21 - 2 instances runs on 16-byte rows (csrc/fpx_depgraph_pk.hpp), everything else on 32-byte rows;
 * FPX_DG_WIDE=1 in the environment forces the latter (the results do not depend on it); FPX_DG_HASH_BITS=b (2 .. 22) shortens
 * the closure hash -- a test hook that makes the collisions needs_host_path reports happen (tests/test_depgraph_dev.py). */
int32_t fpx_epx_execute_dev(fpx_epx* epx, int32_t m, const int32_t* d_leader, const int32_t* d_number,
                            const int32_t* d_packed, const uint8_t* d_committed, const int32_t* first,
                            const int32_t* count, int32_t* d_order, int32_t* d_component, int64_t* num_executed,
                            int64_t* num_components, int32_t* needs_host_path);
/* The same on HOST arrays (what a JNI caller holds: frankenpaxos_amd/jni/EPaxosNative.scala, `deviceExecution`): message i
 * = instance (leader[i], number[i]) committed with the watermarks deps[i * n .. i * n + n) and, if deps_values_end is given
 * and deps_values_end[i] > 0, the explicit ids number[i] + 1 .. deps_values_end[i] - 1 of its own column
 * (dependencies.subtractOne, Replica.scala:582; then deps[i * n + leader[i]] must equal number[i]); committed (may be NULL =
 * all) as above.  order / component: m entries each, filled for p < *num_executed.  Builds the packed lines, uploads,
 * runs fpx_epx_execute_dev, downloads; synchronous. */
int32_t fpx_epx_execute(fpx_epx* epx, int32_t m, const int32_t* leader, const int32_t* number, const int32_t* deps,
                        const int32_t* deps_values_end, const uint8_t* committed, const int32_t* first,
                        const int32_t* count, int32_t* order, int32_t* component, int64_t* num_executed,
                        int64_t* num_components, int32_t* needs_host_path);

/* ---- EPaxos beyond fresh instances: the per-instance Paxos on the command log (num_instances > 0) --------------
 * Ballots are (ordering, replicaIndex), compared lexicographically (epaxos/BallotHelpers.scala:11-21); where one
 * int32 carries a ballot it is ordering * 8 + replicaIndex, the null ballot (-1, -1) (Replica.scala:256) is -1.
 * Command-log entry kinds: 0 none, 1 NoCommandEntry, 2 PreAcceptedEntry, 3 AcceptedEntry, 4 CommittedEntry
 * (Replica.scala:303-330).  A CommandTriple is the caller's int32 triple_id (the command; pre-accept: the optional
 * triple_id argument) plus its dependencies, which the command log keeps with every entry a pre-accept wrote (what
 * THAT replica answered, :1259-1271; the agreed dependencies after a fast-path commit): n watermarks + the end of
 * the own-leader column's explicit values (fpx_epx_read_cmdlog_deps).  An Accept names its triple by id alone:
 * the entries it writes have dependency column 0 = -1 ("look the triple up by its id").
 * fpx_epx_preaccept with a command log: every participating replica must not know the instance yet (the
 * `cmdLog.get == None` branch of handlePreAccept, Replica.scala:1169-1172; anything else is FPX_EINVAL, nothing
 * applied); it records PreAcceptedEntry(Ballot(0, leader), Ballot(0, leader), triple) at the participants, and a
 * fast-path commit turns the entry into CommittedEntry at every replica.
 *
 * The three calls below deliver message i, in array order, to the replicas in target_mask[i] (bit r); the instances
 * (leader[i], number[i]) of one call must be pairwise distinct (FPX_EINVAL otherwise, nothing applied).  Replies
 * per message (host pointers, may be NULL): ok_bits / nack_bits / commit_bits (the replica answered with the Commit
 * it already holds), nack_ballot = the largest `largestBallot` carried by a Nack (Nack(instance, largestBallot),
 * Replica.scala:1424, 1652), -1 if none.
 *
 * fpx_epx_prepare: Replica.handlePrepare (Replica.scala:1632-1757) -- phase 1 of an instance's recovery.  reply_*
 *   are m x n: the PrepareOk of replica r = (status: 0 NotSeen / 2 PreAccepted / 3 Accepted, voteBallot, triple id);
 *   -1 where r sent no PrepareOk.
 * fpx_epx_accept: the Accept phase of message i proposed by replica ballot_replica[i] in ballot
 *   (ballot_ordering[i], ballot_replica[i]): transitionToAcceptPhase at the proposer (:732-792; target_mask must
 *   not contain it), handleAccept at the targets (:1421-1511), handleAcceptOk (:1513-1565): with f + 1 responses,
 *   the proposer's own included, the instance is committed -- CommittedEntry at every replica (commit :815-860 and
 *   Commit to the others).  A proposer that holds a CommittedEntry, or an entry with a larger ballot, would have
 *   died in logger.fatal / logger.check: FPX_EFATAL_PROTOCOL, that message is skipped.
 *   key[i] / is_set[i] = the triple's command (key -1 = Noop): wherever the reference stores the triple it also calls
 *   updateConflictIndex(instance, commandOrNoop) (:602-614) -- at the proposer (:763), at every replica that takes
 *   the Accept in (:1503) and, on commit, at every replica (:828) -- so a replica that first hears of an instance
 *   through an Accept or a Commit reports it as a conflict from then on. */
int32_t fpx_epx_prepare(fpx_epx* epx, int32_t m, const int32_t* leader, const int32_t* number,
                        const int32_t* ballot_ordering, const int32_t* ballot_replica, const uint8_t* target_mask,
                        uint8_t* ok_bits, uint8_t* nack_bits, uint8_t* commit_bits, int32_t* nack_ballot,
                        int32_t* reply_status, int32_t* reply_vote_ballot, int32_t* reply_triple);
/* K8: Replica.handlePrepareOk (epaxos/Replica.scala:1759-1884) -- the decision of the replica that recovers instance i
 * in ballot (ballot_ordering[i], ballot_replica[i]) once it holds the PrepareOks of the replicas in resp_mask[i]
 * (fpx_epx_prepare's ok_bits) with the contents reply_* (fpx_epx_prepare's outputs, m x n):
 *   action 0 = fewer than f + 1 responses, wait (:1799-1801)
 *   action 1 = transitionToAcceptPhase(instance, ballot, the triple replica `source` reported)     -> fpx_epx_accept
 *   action 2 = transitionToPreAcceptPhase(instance, ballot, the command of `source`'s triple, avoidFastPath = true)
 *   action 3 = transitionToPreAcceptPhase(instance, ballot, Noop, avoidFastPath = true)  -> fpx_epx_handle_preaccept
 * triple[i] = that triple's id (-1 for action 0 / 3); source = the lowest replica index among the eligible responses
 * (the reference takes whichever its hash map yields first; they carry the same command).
 * as_intended = 0 reproduces the reference AS IT EVALUATES: its test for an Accepted response compares the required
 * enum field `status` with an Option (:1810, always false) and its default-ballot filter looks at the ballot of the
 * Prepare that was answered instead of the vote's (:1831, never the default ballot during a recovery), so only
 * actions 0, 2 and 3 can come out.  as_intended = 1 evaluates the two tests the way the surrounding comments describe
 * them (an Accepted response at the highest voteBallot wins; f identical PreAccepted triples voted in
 * Ballot(0, leader), the recovering replica's own excluded, win: Util.popularItems(.., f) with the triples compared as
 * the command log stores them).  Reads the command log, changes nothing.  The PrepareOks' triples are compared through
 * the command-log entries they were answered from (reply_triple names a triple, the entry holds its dependencies): the
 * caller must not let another call touch these instances between fpx_epx_prepare and this one -- a decision taken on
 * entries that moved meanwhile is taken on other data than the PrepareOks carried (the reference's replies carry the
 * triples themselves, Replica.scala:1717-1731). */
int32_t fpx_epx_handle_prepare_oks(fpx_epx* epx, int32_t m, const int32_t* leader, const int32_t* number,
                                   const int32_t* ballot_ordering, const int32_t* ballot_replica,
                                   const uint8_t* resp_mask, const int32_t* reply_status,
                                   const int32_t* reply_vote_ballot, const int32_t* reply_triple, int32_t as_intended,
                                   int32_t* action, int32_t* source, int32_t* triple);
int32_t fpx_epx_accept(fpx_epx* epx, int32_t m, const int32_t* leader, const int32_t* number,
                       const int32_t* ballot_ordering, const int32_t* ballot_replica, const int32_t* triple_id,
                       const int32_t* key, const uint8_t* is_set, const uint8_t* target_mask, uint8_t* ok_bits,
                       uint8_t* nack_bits, uint8_t* commit_bits, int32_t* nack_ballot, uint8_t* committed);
/* K7: Replica.handlePreAccept in full (epaxos/Replica.scala:1159-1289) -- what fpx_epx_preaccept's tick-at-once form
 * leaves out: PreAccepts for instances a replica already knows (a leader's re-sent PreAccept, a recovering replica
 * pre-accepting again in a higher ballot).  Message i = PreAccept(instance (leader, number), ballot
 * (ballot_ordering, ballot_replica), single-key get / set on key[i] or Noop (key[i] = -1), sequenceNumber 0,
 * dependencies deps_in[i * n ..] = n watermarks + deps_in_values_end[i] (explicit values number + 1 .. end - 1 of
 * the instance's own-leader column, 0 = none; the array may be NULL; a PreAccept that depends on its own instance is
 * FPX_EINVAL)), delivered in array order to the replicas of target_mask[i]; instances pairwise distinct per call.
 * At each replica, cmdLog.get(instance) decides (:1169-1238):
 *   none                                            -> processed
 *   NoCommandEntry(b):       ballot < b             -> Nack(instance, largestBallot)
 *   PreAcceptedEntry(b, vb): ballot < b -> Nack;  ballot == vb -> the PreAcceptOk again, from the stored triple
 *   AcceptedEntry(b, vb):    ballot < b -> Nack;  ballot == vb -> ignored
 *   CommittedEntry(triple)                          -> the Commit back
 *   otherwise processed: largestBallot = max(largestBallot, ballot) (:1251); dependencies = the command's conflicts
 *   in THIS replica's index minus the instance itself (computeSequenceNumberAndDependencies :569-600; none for a
 *   Noop) U deps_in (:1257-1262); PreAcceptedEntry(ballot, ballot, triple) (:1265-1276); updateConflictIndex (:1279,
 *   a Noop leaves the index alone); PreAcceptOk(dependencies).  (The leaderStates / timer bookkeeping of :1243-1254
 *   is leader-side state outside this path.)
 * Replies (host pointers, may be NULL): ok_bits (processed), resend_bits, nack_bits, commit_bits -- the other replicas
 * of target_mask ignored the message; nack_ballot as above; reply_deps (m x n x n) / reply_values_end (m x n) /
 * reply_triple (m x n): the dependencies and triple id of the PreAcceptOk or Commit replica r sent (zeros / 0 / -1
 * where it sent neither). */
int32_t fpx_epx_handle_preaccept(fpx_epx* epx, int32_t m, const int32_t* leader, const int32_t* number,
                                 const int32_t* ballot_ordering, const int32_t* ballot_replica, const int32_t* key,
                                 const uint8_t* is_set, const int32_t* triple_id, const int32_t* deps_in,
                                 const int32_t* deps_in_values_end, const uint8_t* target_mask, uint8_t* ok_bits,
                                 uint8_t* resend_bits, uint8_t* nack_bits, uint8_t* commit_bits, int32_t* nack_ballot,
                                 int32_t* reply_deps, int32_t* reply_values_end, int32_t* reply_triple);
/* Replica.handleCommit (epaxos/Replica.scala:1567-1575 -> commit, :815-830) at every replica of target_mask: whatever the
 * replica's command log held for the instance is replaced by CommittedEntry(triple) -- a Commit is final, no ballot is
 * compared (:826-827) -- and its conflict index learns the command (:828; key -1 = Noop).  deps: n watermarks per
 * message + deps_values_end (the explicit ids number + 1 .. end - 1 of the own-leader column, 0 = none), or NULL: the
 * triple is known by its id alone, as after an Accept.  Instances of one call that repeat must carry the same triple.
 * Timers, leaderStates and the dependency graph (:822, :831, :859-875) are the caller's.  FPX_EINVAL, nothing applied:
 * an instance outside the command log, a key outside the index, a replica outside target_mask's n bits, negative
 * watermarks, explicit ids that do not lie above the instance. */
int32_t fpx_epx_handle_commit(fpx_epx* epx, int32_t m, const int32_t* leader, const int32_t* number,
                              const int32_t* triple_id, const int32_t* key, const uint8_t* is_set, const int32_t* deps,
                              const int32_t* deps_values_end, const uint8_t* target_mask);
/* one command-log entry: out[0..4] = kind, ballot, voteBallot, triple id, the replica's largestBallot */
int32_t fpx_epx_read_cmdlog(fpx_epx* epx, int32_t replica, int32_t leader, int32_t number, int32_t out[5]);
/* the dependencies kept with that entry: deps[n] watermarks (deps[0] = -1: known by triple id only), *values_end */
int32_t fpx_epx_read_cmdlog_deps(fpx_epx* epx, int32_t replica, int32_t leader, int32_t number, int32_t* deps,
                                 int32_t* values_end);
/* replica's conflict-index entry of one key: gets[n], sets[n] (TopOne vectors) */
int32_t fpx_epx_read_index(fpx_epx* epx, int32_t replica, int32_t key, int32_t* gets, int32_t* sets);

/* ---- next rows of SURVEY.md section 8(f) ----------------------------------------------------------
 *
 * f1  Replica.handleChosen + executeLog (multipaxos/Replica.scala:572-590, 394-447;
 *     util/BufferMap.scala:29-51): the receiver of Chosen.  One replica log per context (every replica
 *     receives the same Chosen stream).  For i in order with mask[i] != 0 (mask NULL = all):
 *     log.get(slot) defined -> ignored (redundantly chosen); else log.put(slot, value), numChosen += 1;
 *     then the contiguous prefix executes: executedWatermark advances while log.get(it) is defined.
 *     Outputs (may be NULL): the new executedWatermark and numChosen. */
int32_t fpx_replica_chosen(fpx_ctx* ctx, int32_t n, const int32_t* slot, const int32_t* value_id,
                           const uint8_t* mask, int32_t* executed_watermark, int32_t* num_chosen);
/* device-resident inputs; the result is read with fpx_replica_state after fpx_sync */
int32_t fpx_replica_chosen_dev(fpx_ctx* ctx, int32_t n, const int32_t* d_slot,
                               const int32_t* d_value_id, const uint8_t* d_mask);
int32_t fpx_replica_state(fpx_ctx* ctx, int32_t* executed_watermark, int32_t* num_chosen);
/* mencius.Replica.handleChosenNoopRange (mencius/Replica.scala:464-485): the slots slot_start,
 * slot_start + num_leader_groups, ... below slot_end are put as Noop in order until the first one that is
 * ALREADY in the log -- there the reference handler returns: the rest of the range is dropped and
 * executeLog does not run (kept as is).  Otherwise the contiguous prefix executes.  Synchronous. */
int32_t fpx_replica_chosen_noop_range(fpx_ctx* ctx, int32_t slot_start, int32_t slot_end,
                                      int32_t* executed_watermark, int32_t* num_chosen);
/* log entries [first, first + count): value (-1 where absent) and present flag */
int32_t fpx_replica_read_log(fpx_ctx* ctx, int32_t first, int32_t count, int32_t* values,
                             uint8_t* present);

/* f2  Leader.handlePhase1b recovery scan (multipaxos/Leader.scala:306-329 safeValue, :543-566).
 *     quorum_masks: one 4-word set per acceptor group (num_leader_groups * num_groups x 4): the
 *     acceptors whose Phase1b the leader holds.  maxSlot = the largest slot >= chosen_watermark in
 *     which any of them voted (-1 if none).  For slot = chosen_watermark .. maxSlot (at most cap
 *     entries are written): safe_round = the highest voteRound among the quorum's acceptors of the
 *     slot's group that voted in the slot (-1 if none voted), safe_value = the value voted in that
 *     round (FPX_NOOP if none: "everything is safe, we return Noop"). */
int32_t fpx_leader_phase1b_scan(fpx_ctx* ctx, int32_t chosen_watermark, const uint64_t* quorum_masks,
                                int32_t cap, int32_t* max_slot, int32_t* safe_round,
                                int32_t* safe_value);

/* Acceptor.maxVotedSlot (multipaxos/Acceptor.scala:104, 208) as the read path reports it (handleMaxSlotRequest /
 * handleBatchMaxSlotRequest, :222-254), restricted to the slots [first_slot, first_slot + count) of the acceptor's group:
 * the largest of them in which the acceptor holds a vote, -1 if there is none.  With first_slot = 0 and count = num_slots
 * this is the scalar fpx_read_acceptor returns; a caller that maps an unbounded log onto the context's rows (row =
 * slot % num_slots: jni/Native.scala) asks lap by lap, because the maximum over rows is not the maximum over slots once
 * the window has wrapped.  Synchronous; one strided read of the acceptor's column. */
int32_t fpx_acceptor_max_voted_in(fpx_ctx* ctx, int32_t group, int32_t replica, int32_t first_slot, int32_t count,
                                  int32_t* max_slot);
/* Acceptor.handlePhase1a's reply (multipaxos/Acceptor.scala:163-181; mencius/Acceptor.scala:181-199): the
 * Phase1b.info of acceptor `replica` of `group` -- one Phase1bSlotInfo(slot, voteRound, voteValue) per slot >=
 * chosen_watermark in which the acceptor has voted, in ascending slot order (states.iteratorFrom).  *count = how
 * many there are; the first min(*count, cap) are written (call with cap = 0 to size the arrays).  Together with
 * fpx_acceptor_phase1a (the round movement) this is everything an acceptor-side actor needs to answer a Leader's
 * Phase1a with the Phase1b the unchanged Leader.handlePhase1b expects (multipaxos/Leader.scala:504-577). */
int32_t fpx_acceptor_phase1b_info(fpx_ctx* ctx, int32_t group, int32_t replica, int32_t chosen_watermark,
                                  int32_t cap, int32_t* count, int32_t* slot, int32_t* vote_round,
                                  int32_t* vote_value);

/* ---- state readback (parity) ------------------------------------------------------------------- */
/* acceptor `replica` of `group`: its round (FPX_BALLOT_ACCEPTOR; -1 in PER_SLOT mode),
 * maxVotedSlot, and for every slot s in [0, S): vote_round[s] / vote_value[s] (-1 / -1 when the
 * acceptor has no vote in s or s belongs to another group) and, in PER_SLOT mode, ballot[s].
 * Array arguments may be NULL. */
int32_t fpx_read_acceptor(fpx_ctx* ctx, int32_t group, int32_t replica, int32_t* promised,
                          int32_t* max_voted_slot, int32_t* vote_round, int32_t* vote_value,
                          int32_t* ballot);
/* whole-array readback, slot-major [S][R] int32 (may be NULL each) */
int32_t fpx_read_state(fpx_ctx* ctx, int32_t* vote_round, int32_t* vote_value, int32_t* ballot);
/* per-acceptor scalars, [num_leader_groups * num_groups][R] */
int32_t fpx_read_scalars(fpx_ctx* ctx, int32_t* promised, int32_t* max_voted_slot);
/* proxy-leader tallies of one slot: up to tally_ways entries; state 0 = Pending, 1 = Done */
int32_t fpx_read_tally(fpx_ctx* ctx, int32_t slot, int32_t* num_entries, int32_t* rounds,
                       int32_t* states, int32_t* values, uint64_t* vote_bits /* ways x 4 */);

/* ---- multi-GPU: one context per GPU, one RCCL communicator over them ----------
 *
 * The Phase-2 path shards two ways.
 * (1) By acceptor group -- no exchange step: slot -> group is fixed (multipaxos/ProxyLeader.scala:190,
 *     mencius/ProxyLeader.scala:231-234) and groups never interact in Phase 2, so rank r simply owns the
 *     groups g with g % world == r and runs the single-GPU entry points on its own slots.  Only Chosen
 *     records leave a GPU; fpx_comm_allgather_chosen_dev hands every rank all of them when a device-side
 *     replica log is kept on each GPU (13 B per slot per rank over xGMI).
 * (2) By the acceptor axis of ONE big group (flexible mode: "the log is not partitioned",
 *     multipaxos/Config.scala:16-21) -- one exchange step, replacing the fan-in of every acceptor's Phase2b
 *     to one proxy leader (multipaxos/ProxyLeader.scala:217-258): the context of rank r is created with
 *     replica_base = r * R / world, num_replicas = R / world, replicas_total = R;
 *     fpx_phase2_replica_sharded_dev runs K1 on its acceptor columns for all n slots (partial 256-bit vote
 *     bitmaps with bits only in its own range), ONE ncclReduceScatter(ncclSum, ncclUint64) of the 4 n words
 *     (disjoint bit ranges: sum == OR and never carries) gives rank r the full bitmaps of ITS n / world
 *     slots, and the proxy-leader open + tally (K2) runs on that slice.  xGMI traffic per GPU and call:
 *     a ring reduce-scatter sends and receives (world - 1) / world x 32 B x n (28 MiB for n = 2^20 at
 *     world = 8, ~0.2 ms at one 153 GB/s link); an all-reduce would move twice that and make every GPU tally
 *     all n slots.
 * RCCL is loaded at run time (an RCCL already in the process -- PyTorch's -- is reused; else librccl.so.1 /
 * $FPX_RCCL_LIB): single-GPU callers never need it.  The id comes from fpx_comm_unique_id on one rank and
 * reaches the others out of band (the JVM actors' own transport, MPI, a torch.distributed broadcast ...). */
#define FPX_COMM_ID_BYTES 128
int32_t fpx_comm_unique_id(uint8_t id[FPX_COMM_ID_BYTES]);
/* collective: every rank calls it with the same id and world, its own rank; binds the communicator to ctx */
int32_t fpx_comm_create(fpx_ctx* ctx, const uint8_t id[FPX_COMM_ID_BYTES], int32_t rank, int32_t world);
int32_t fpx_comm_destroy(fpx_ctx* ctx);
int32_t fpx_comm_info(fpx_ctx* ctx, int32_t* rank, int32_t* world); /* (0, 1) without a communicator */
int32_t fpx_last_rccl_error(fpx_ctx* ctx);
/* (2) above.  n must be a multiple of world; all ranks pass the same n messages (device pointers,
 * asynchronous on the context's stream, run contract as for the other _dev calls).  d_target_mask: n x 4
 * words over the WHOLE group (bit j = acceptor j of replicas_total) or NULL.  Outputs cover this rank's
 * slice of the batch, messages [rank * n / world, (rank + 1) * n / world): d_chosen / d_chosen_round /
 * d_chosen_value have n / world entries (as fpx_proxy_phase2b_dev); d_nack_round (n entries or NULL) is the
 * largest round Nacked by ANY rank's acceptors (ncclAllReduce(ncclMax) of the ranks' own, in place: the same n
 * values on every rank).  Without a communicator (world 1) it is K1 + open + K2. */
int32_t fpx_phase2_replica_sharded_dev(fpx_ctx* ctx, int32_t n, const int32_t* d_slot, const int32_t* d_round,
                                       const int32_t* d_value_id, const uint64_t* d_target_mask,
                                       uint8_t* d_chosen, int32_t* d_chosen_round, int32_t* d_chosen_value,
                                       int32_t* d_nack_round);
/* (1) above: all-gather of the Chosen records of n_local messages per rank; the d_all_* arrays have
 * world x n_local entries, rank-major.  Any of the three pairs may be NULL. */
int32_t fpx_comm_allgather_chosen_dev(fpx_ctx* ctx, int32_t n_local, const uint8_t* d_chosen,
                                      const int32_t* d_chosen_round, const int32_t* d_chosen_value,
                                      uint8_t* d_all_chosen, int32_t* d_all_round, int32_t* d_all_value);
/* with fpx_profile_enable: number and summed duration of the collectives since the last read (HIP events
 * on the context's stream around each RCCL call) */
int32_t fpx_profile_read_collective(fpx_ctx* ctx, int32_t* launches, double* total_ms);

/* Whole-state digests for parity checks at sizes where a readback is gigabytes: out[0..6] =
 * vote_round cells, vote_value cells, ballot cells (0 in FPX_BALLOT_ACCEPTOR mode), acceptors' rounds,
 * acceptors' maxVotedSlot, the proxy leader's single-slot tallies, the replica's log (+ executedWatermark,
 * numChosen), the proxy leader's noop-range tallies.  Each is an order-independent wrapping sum of a 64-bit hash per element (per cell
 * (slot, acceptor): splitmix64-finalise((slot * R + acceptor) * 0x9E3779B97F4A7C15 + (uint32)value + 1)), so
 * equal digests <=> equal state up to 2^-64.  Waits for the stream. */
int32_t fpx_state_digest(fpx_ctx* ctx, uint64_t out[8]);

#ifdef __cplusplus
}
#endif
#endif /* FPX_H */
