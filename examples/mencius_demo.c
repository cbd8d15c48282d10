/* mencius_demo.c -- the C ABI of include/fpx.h from plain C on the shape of BASELINE.json configs[4], small:
 * Mencius with 4 leader groups x one group of 3 acceptors, a band of 4096 slots (of a window of 8192).  The leader groups that have commands
 * propose them (the leader groups' batches back to back, each in slot order: what the leader-group-major rows of a
 * Mencius context like best), the others skip their slots with one Phase2aNoopRange each; everything is chosen; the
 * replica's log then executes the whole band.  A stale leader of one group is Nacked after a leader change there.
 *
 *   gcc -std=c11 -O2 examples/mencius_demo.c -Iinclude -Lfrankenpaxos_amd/csrc -lfpx \
 *       -Wl,-rpath,$PWD/frankenpaxos_amd/csrc -Wl,-rpath,/opt/rocm/lib -o /tmp/mencius_demo && /tmp/mencius_demo
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fpx.h"

#define L 4     /* leader groups: slot s belongs to leader group s % L (mencius/ProxyLeader.scala:169-176) */
#define S 4096  /* the band */
#define ROWS (S / L)

int main(void) {
  fpx_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.num_slots = 2 * S, cfg.num_replicas = 3, cfg.num_groups = 1, cfg.num_leader_groups = L;
  cfg.f = 1, cfg.quorum_kind = FPX_Q_THRESHOLD, cfg.num_leaders = 2;
  cfg.ballot_mode = FPX_BALLOT_ACCEPTOR, cfg.tally_ways = 4;
  fpx_ctx* ctx = NULL;
  int32_t st = fpx_create(&cfg, &ctx);
  if (st != FPX_OK) {
    printf("fpx_create: %s\n", fpx_strerror(st));  /* FPX_ENODEVICE without an MI355X: there is no CPU path */
    return st == FPX_ENODEVICE ? 77 : 1;
  }
  /* every leader group's leader of round 0 runs Phase 1 on its acceptor group (group id = lg * num_groups + ag) */
  for (int lg = 0; lg < L; ++lg)
    if ((st = fpx_acceptor_phase1a(ctx, lg, 0, 0, NULL, NULL, NULL)) != FPX_OK) return 1;

  /* leader groups 0 and 2 have commands for all their slots of the band: group 0's batch, then group 2's */
  static int32_t slot[2 * ROWS], round[2 * ROWS], value[2 * ROWS], chosen_round[2 * ROWS], chosen_value[2 * ROWS], nack_round[2 * ROWS];
  static uint8_t chosen[2 * ROWS];
  int n = 0;
  for (int lg = 0; lg < L; lg += 2)
    for (int j = 0; j < ROWS; ++j, ++n) slot[n] = j * L + lg, round[n] = 0, value[n] = 500000 + slot[n];
  st = fpx_phase2_fused(ctx, n, slot, round, value, NULL, chosen, chosen_round, chosen_value, nack_round);
  int n_chosen = 0;
  for (int i = 0; i < n; ++i) n_chosen += chosen[i] && chosen_value[i] == value[i];
  printf("commands: status %d, %d of %d chosen\n", st, n_chosen, n);

  /* leader groups 1 and 3 have nothing to propose: one Phase2aNoopRange each covers their slots of the band
   * (mencius/Leader.scala:342-345; [start, end) stands for start, start + L, ... below end) */
  int32_t start[2] = {1, 3}, end[2] = {S - L + 1 + 1, S - L + 3 + 1}, rround[2] = {0, 0}, rnack[2];
  uint8_t is_new[2], rchosen[2];
  st = fpx_noop_ranges_fused(ctx, 2, start, end, rround, NULL, NULL, NULL, rnack, is_new, rchosen);
  printf("noop ranges: status %d, new %d %d, chosen %d %d\n", st, is_new[0], is_new[1], rchosen[0], rchosen[1]);

  /* the replica: Chosen for every command, ChosenNoopRange for the two ranges -> the whole band executes */
  int32_t watermark = -1, num = -1;
  if ((st = fpx_replica_chosen(ctx, n, slot, chosen_value, chosen, &watermark, &num)) != FPX_OK) return 1;
  for (int k = 0; k < 2; ++k)
    if ((st = fpx_replica_chosen_noop_range(ctx, start[k], end[k], &watermark, &num)) != FPX_OK) return 1;
  printf("replica: executed watermark %d of %d, %d slots in the log\n", watermark, S, num);

  /* leader group 2 changes leaders (round 1); its old leader's Phase2a is Nacked with that round, group 0 goes on */
  const int32_t r1 = fpx_next_classic_round(2, 1, 0);
  if ((st = fpx_acceptor_phase1a(ctx, 2, r1, 0, NULL, NULL, NULL)) != FPX_OK) return 1;
  int32_t s2[2] = {S + 2, S}, rr2[2] = {0, 0}, v2[2] = {7, 8}, cr2[2], cv2[2], nr2[2];  /* the next band's first slots */
  uint8_t ch2[2];
  st = fpx_phase2_fused(ctx, 2, s2, rr2, v2, NULL, ch2, cr2, cv2, nr2);
  printf("after the change in leader group 2: status %d; slot %d (group 2) in round 0: chosen %d, Nack round %d; slot %d (group 0) in round 0: chosen %d value %d\n",
         st, s2[0], ch2[0], nr2[0], s2[1], ch2[1], cv2[1]);
  const int ok = n_chosen == n && rchosen[0] && rchosen[1] && watermark == S && num == S && !ch2[0] && nr2[0] == r1 && ch2[1] && cv2[1] == 8;
  fpx_destroy(ctx);
  return ok ? 0 : 1;
}
