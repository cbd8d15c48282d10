/*
 * fpx_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, single-threaded, message-at-a-time) of the frankenpaxos Phase-2 hot
 * path, used as the parity oracle for the HIP library and as the timed `cpu_baseline` ("port") in
 * bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this;
 * libfpx never links or calls it.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference/shared/src/main/scala/frankenpaxos/).
 *
 * Parity pinning: the quorum predicates, the round system and the replica log
 * container are pinned by the reference's own known-answer tests, transcribed in
 * tests/test_oracle_golden.py.  The vote (a1/a2) and the tally (a3/a4) have NO golden vectors in the
 * reference (its protocol tests are randomized invariant checks seeded from the wall clock) and the
 * JVM reference cannot run in this environment (no JDK): for those two functions parity is UNPINNED;
 * the restatement is anchored line-by-line on the citations below, on hand-computed micro-traces
 * (tests/test_oracle_traces.py) and on the reference's safety invariant re-implemented in
 * tests/test_oracle_invariants.py.
 */
#ifndef FPX_ORACLE_H
#define FPX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- roundsystem.ClassicRoundRobin  roundsystem/RoundSystem.scala:60-87 ----------------------- */
int fpo_round_leader(int n, int round);                      /* :63     */
int fpo_next_classic_round(int n, int leader_index, int round); /* :66-81 */

/* ---- quorums.*  quorums/{QuorumSystem,SimpleMajority,Grid,UnanimousWrites}.scala --------------
 * Node sets are arrays of arbitrary int ids (Set[Int]).  Predicates return 1 / 0, or -1 when the
 * Scala code would throw IllegalArgumentException from require(...). */
typedef struct fpo_qs fpo_qs;
fpo_qs* fpo_qs_simple_majority(const int* members, int n); /* SimpleMajority.scala:19-31 ; NULL on require */
fpo_qs* fpo_qs_unanimous_writes(const int* members, int n); /* UnanimousWrites.scala:17-26             */
fpo_qs* fpo_qs_grid(const int* grid, int rows, int cols);  /* Grid.scala:5-18 (row-major rows x cols)  */
void fpo_qs_free(fpo_qs* qs);
int fpo_qs_nodes(const fpo_qs* qs, int* out);              /* returns |nodes| */
int fpo_qs_is_read_quorum(const fpo_qs* qs, const int* xs, int n);
int fpo_qs_is_write_quorum(const fpo_qs* qs, const int* xs, int n);
int fpo_qs_is_superset_of_read_quorum(const fpo_qs* qs, const int* xs, int n);
int fpo_qs_is_superset_of_write_quorum(const fpo_qs* qs, const int* xs, int n);
/* random quorums; the reference uses scala.util.Random, which never reaches the chosen values
 * (SURVEY.md F12), so the oracle uses splitmix64 on *rng.  Return the quorum size. */
int fpo_qs_random_read_quorum(const fpo_qs* qs, uint64_t* rng, int* out);
int fpo_qs_random_write_quorum(const fpo_qs* qs, uint64_t* rng, int* out);

uint64_t fpo_splitmix64(uint64_t* state);

/* ---- util.BufferMap + Replica log prefix  util/BufferMap.scala:8-62, multipaxos/Replica.scala:394-447,572-590 */
typedef struct fpo_log fpo_log;
fpo_log* fpo_log_new(int grow_size);
void fpo_log_free(fpo_log* log);
int fpo_log_get(const fpo_log* log, int key, int* value); /* 1 = Some, 0 = None          BufferMap.scala:29-35 */
void fpo_log_put(fpo_log* log, int key, int value);       /*                              BufferMap.scala:37-51 */
void fpo_log_garbage_collect(fpo_log* log, int watermark); /*                             BufferMap.scala:55-62 */
/* Replica.handleChosen: ignore if already present, else put and execute the contiguous prefix;
 * returns the new executedWatermark (Replica.scala:572-590, 394-447) */
int fpo_log_chosen(fpo_log* log, int slot, int value);
int fpo_log_executed_watermark(const fpo_log* log);
int fpo_log_largest_key(const fpo_log* log);                 /* BufferMap.scala:17 */

/* ---- the Phase-2 system: acceptor groups + one proxy leader -------------------------------------
 * Same configuration fields and the same batch semantics as include/fpx.h, evaluated
 * message-at-a-time in array order by the per-actor handlers below. */
typedef struct {
  int32_t num_slots, num_replicas, num_groups, num_leader_groups;
  int32_t f, quorum_kind, grid_rows, grid_cols, num_leaders, ballot_mode, tally_ways;
  int32_t replica_base, replicas_total, device;
  uint32_t flags;
} fpo_config; /* layout-identical to fpx_config */

enum { FPO_OK = 0, FPO_EINVAL = 1, FPO_EFATAL_UNKNOWN_SLOTROUND = 2, FPO_ECAPACITY = 5 };

typedef struct fpo_sys fpo_sys;
int fpo_config_check(const fpo_config* cfg);
fpo_sys* fpo_sys_new(const fpo_config* cfg);
void fpo_sys_free(fpo_sys* sys);
void fpo_sys_reset(fpo_sys* sys);
int fpo_group_of_slot(const fpo_config* cfg, int slot);

/* single-message handlers (the restatement proper) */
/* multipaxos.Acceptor.handlePhase2a  multipaxos/Acceptor.scala:184-220 (mencius/Acceptor.scala:202-235).
 * Returns 1 = Phase2b(slot, round) sent, 0 = Nack(*reply_round) sent. */
int fpo_acceptor_handle_phase2a(fpo_sys* sys, int group, int replica, int slot, int round, int value,
                                int* reply_round);
/* multipaxos.Acceptor.handlePhase1a  multipaxos/Acceptor.scala:148-182.  1 = Phase1b, 0 = Nack */
int fpo_acceptor_handle_phase1a(fpo_sys* sys, int group, int replica, int round, int chosen_watermark,
                                int* reply_round);
/* multipaxos.ProxyLeader.handlePhase2a bookkeeping  multipaxos/ProxyLeader.scala:175-215.
 * 1 = new Pending created, 0 = ignored (already known) */
int fpo_proxy_handle_phase2a(fpo_sys* sys, int slot, int round, int value);
/* multipaxos.ProxyLeader.handlePhase2b  multipaxos/ProxyLeader.scala:217-258.
 * returns 1 = Chosen(slot, *chosen_value) emitted, 0 = waiting, 2 = ignored (Done),
 * -1 = logger.fatal (unknown slot/round) */
int fpo_proxy_handle_phase2b(fpo_sys* sys, int acceptor_bit, int slot, int round, int* chosen_value);
/* the quorum predicate the proxy leader applies to a 256-bit acceptor set (strict: -1 on foreign bit) */
int fpo_sys_is_write_quorum(const fpo_config* cfg, const uint64_t nodes[4], int strict);
int fpo_sys_is_read_quorum(const fpo_config* cfg, const uint64_t nodes[4], int strict);

/* batch entry points: identical signatures/semantics to the host-pointer entry points of fpx.h */
int fpo_acceptor_phase2a(fpo_sys* sys, int32_t n, const int32_t* slot, const int32_t* round,
                         const int32_t* value_id, const uint64_t* target_mask, uint64_t* vote_bits,
                         uint64_t* nack_bits, int32_t* nack_round);
int fpo_acceptor_phase1a(fpo_sys* sys, int32_t group, int32_t round, int32_t chosen_watermark,
                         const uint64_t* target_mask, uint64_t* promised_bits, uint64_t* nack_bits);
int fpo_proxy_open(fpo_sys* sys, int32_t n, const int32_t* slot, const int32_t* round,
                   const int32_t* value_id, uint8_t* is_new);
int fpo_proxy_phase2b(fpo_sys* sys, int32_t n, const int32_t* slot, const int32_t* round,
                      const uint64_t* vote_bits, uint8_t* newly_chosen, int32_t* chosen_round,
                      int32_t* chosen_value);
int fpo_phase2_fused(fpo_sys* sys, int32_t n, const int32_t* slot, const int32_t* round,
                     const int32_t* value_id, const uint64_t* target_mask, uint8_t* chosen,
                     int32_t* chosen_round, int32_t* chosen_value, int32_t* nack_round);
/* the same fused step delivered through a strict FIFO message pump (FakeTransport.scala:89-159
 * with FIFO instead of random delivery): all Phase2a's reach the proxy leader first, then the
 * acceptors, then the Phase2b's come back.  Must agree with fpo_phase2_fused. */
int fpo_phase2_fifo_pump(fpo_sys* sys, int32_t n, const int32_t* slot, const int32_t* round,
                         const int32_t* value_id, const uint64_t* target_mask, uint8_t* chosen,
                         int32_t* chosen_round, int32_t* chosen_value, int32_t* nack_round);

/* K4: Mencius noop ranges (mencius/Acceptor.scala:237-291, mencius/ProxyLeader.scala:255-303, 355-411) */
int fpo_acceptor_handle_phase2a_noop_range(fpo_sys* sys, int group, int replica, int slot_start, int slot_end,
                                           int round, int* reply_round);
int fpo_proxy_handle_phase2a_noop_range(fpo_sys* sys, int slot_start, int slot_end, int round);
int fpo_proxy_handle_phase2b_noop_range(fpo_sys* sys, int acceptor_group, int acceptor_index, int slot_start,
                                        int slot_end, int round);
int fpo_acceptor_phase2a_noop_range(fpo_sys* sys, int32_t slot_start, int32_t slot_end, int32_t round,
                                    const uint64_t* target_masks, uint64_t* vote_bits, uint64_t* nack_bits,
                                    int32_t* nack_round);
int fpo_proxy_open_noop_range(fpo_sys* sys, int32_t slot_start, int32_t slot_end, int32_t round, uint8_t* is_new);
int fpo_proxy_phase2b_noop_range(fpo_sys* sys, int32_t slot_start, int32_t slot_end, int32_t round,
                                 const uint64_t* vote_bits, uint8_t* newly_chosen);

/* batched / fused forms (fpx_*_noop_ranges, fpx_noop_ranges_fused of fpx.h): n ranges in array order */
int fpo_acceptor_phase2a_noop_ranges(fpo_sys* sys, int32_t n, const int32_t* start, const int32_t* end,
                                     const int32_t* round, const uint64_t* target_masks, uint64_t* vote_bits,
                                     uint64_t* nack_bits, int32_t* nack_round);
int fpo_proxy_open_noop_ranges(fpo_sys* sys, int32_t n, const int32_t* start, const int32_t* end,
                               const int32_t* round, uint8_t* is_new);
int fpo_proxy_phase2b_noop_ranges(fpo_sys* sys, int32_t n, const int32_t* start, const int32_t* end,
                                  const int32_t* round, const uint64_t* vote_bits, uint8_t* newly_chosen);
int fpo_noop_ranges_fused(fpo_sys* sys, int32_t n, const int32_t* start, const int32_t* end, const int32_t* round,
                          const uint64_t* target_masks, uint64_t* vote_bits, uint64_t* nack_bits, int32_t* nack_round,
                          uint8_t* is_new, uint8_t* chosen);
int fpo_read_range_tally(fpo_sys* sys, int32_t start, int32_t end, int32_t round, int32_t* state, uint64_t* vote_bits);
int fpo_recycle_slots(fpo_sys* sys, int32_t first_slot, int32_t count);
int fpo_proxy_forget(fpo_sys* sys, int32_t first_slot, int32_t count);

/* f1: Replica.handleChosen per message (multipaxos/Replica.scala:572-590) on the system's replica log */
int fpo_replica_chosen(fpo_sys* sys, int32_t n, const int32_t* slot, const int32_t* value_id,
                       const uint8_t* mask, int32_t* executed_watermark, int32_t* num_chosen);
/* mencius.Replica.handleChosenNoopRange, mencius/Replica.scala:464-485 (early `return` kept) */
int fpo_replica_chosen_noop_range(fpo_sys* sys, int32_t slot_start, int32_t slot_end,
                                  int32_t* executed_watermark, int32_t* num_chosen);
int fpo_replica_read_log(fpo_sys* sys, int32_t first, int32_t count, int32_t* values, uint8_t* present);
/* f2: Leader.handlePhase1b recovery (multipaxos/Leader.scala:306-329 safeValue, :543-566) */
int fpo_leader_phase1b_scan(fpo_sys* sys, int32_t chosen_watermark, const uint64_t* quorum_masks,
                            int32_t cap, int32_t* max_slot, int32_t* safe_round, int32_t* safe_value);

/* Phase1b.info of one acceptor: multipaxos/Acceptor.scala:166-178 */
int fpo_acceptor_phase1b_info(fpo_sys* sys, int32_t group, int32_t replica, int32_t chosen_watermark, int32_t cap,
                              int32_t* count, int32_t* slot, int32_t* vote_round, int32_t* vote_value);
int fpo_error_detail(fpo_sys* sys, int32_t* index, int32_t* slot, int32_t* round);
int fpo_read_acceptor(fpo_sys* sys, int32_t group, int32_t replica, int32_t* promised,
                      int32_t* max_voted_slot, int32_t* vote_round, int32_t* vote_value,
                      int32_t* ballot);
int fpo_read_state(fpo_sys* sys, int32_t* vote_round, int32_t* vote_value, int32_t* ballot);
int fpo_read_scalars(fpo_sys* sys, int32_t* promised, int32_t* max_voted_slot);
/* the CPU twin of fpx_state_digest (include/fpx.h) */
int fpo_state_digest(fpo_sys* sys, uint64_t out[8]);
int fpo_read_tally(fpo_sys* sys, int32_t slot, int32_t* num_entries, int32_t* rounds,
                   int32_t* states, int32_t* values, uint64_t* vote_bits);

#ifdef __cplusplus
}
#endif
#endif
