/* phase2_demo.c -- the C ABI of include/fpx.h from plain C: BASELINE.json configs[0] in 60 lines.
 * MultiPaxos f = 1 (one group of 3 acceptors, 2 leaders): the leader of round 0 runs Phase 1, proposes 1000
 * commands, every one is chosen; a stale leader is Nacked after a leader change.
 *
 *   gcc -std=c11 -O2 examples/phase2_demo.c -Iinclude -Lfrankenpaxos_amd/csrc -lfpx \
 *       -Wl,-rpath,$PWD/frankenpaxos_amd/csrc -Wl,-rpath,/opt/rocm/lib -o /tmp/phase2_demo && /tmp/phase2_demo
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fpx.h"

#define N 1000

int main(void) {
  fpx_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.num_slots = 4096, cfg.num_replicas = 3, cfg.num_groups = 1, cfg.num_leader_groups = 1;
  cfg.f = 1, cfg.quorum_kind = FPX_Q_THRESHOLD, cfg.num_leaders = 2;
  cfg.ballot_mode = FPX_BALLOT_ACCEPTOR, cfg.tally_ways = 4;
  fpx_ctx* ctx = NULL;
  int32_t st = fpx_create(&cfg, &ctx);
  if (st != FPX_OK) {
    printf("fpx_create: %s\n", fpx_strerror(st));  /* FPX_ENODEVICE without an MI355X: there is no CPU path */
    return st == FPX_ENODEVICE ? 77 : 1;
  }
  /* Acceptor.handlePhase1a: the leader of round 0 (fpx_round_leader(2, 0) == 0) takes over */
  if ((st = fpx_acceptor_phase1a(ctx, 0, 0, 0, NULL, NULL, NULL)) != FPX_OK) return 1;
  static int32_t slot[N], round[N], value[N], chosen_round[N], chosen_value[N], nack_round[N];
  static uint8_t chosen[N];
  for (int i = 0; i < N; ++i) slot[i] = i, round[i] = 0, value[i] = 100000 + i;
  /* ProxyLeader.handlePhase2a -> Acceptor.handlePhase2a -> ProxyLeader.handlePhase2b for the whole tick */
  st = fpx_phase2_fused(ctx, N, slot, round, value, NULL, chosen, chosen_round, chosen_value, nack_round);
  int n_chosen = 0;
  for (int i = 0; i < N; ++i) n_chosen += chosen[i] && chosen_value[i] == value[i];
  printf("round 0: status %d, %d of %d commands chosen\n", st, n_chosen, N);
  /* leader 1 takes over in its next round; the old leader's Phase2a's are Nacked with that round */
  const int32_t r1 = fpx_next_classic_round(2, 1, 0);
  if ((st = fpx_acceptor_phase1a(ctx, 0, r1, N, NULL, NULL, NULL)) != FPX_OK) return 1;
  for (int i = 0; i < 10; ++i) slot[i] = N + i;
  st = fpx_phase2_fused(ctx, 10, slot, round, value, NULL, chosen, chosen_round, chosen_value, nack_round);
  printf("stale leader after the change to round %d: status %d, chosen %d, Nack carries round %d -> leader %d\n", r1, st,
         chosen[0], nack_round[0], fpx_round_leader(2, nack_round[0]));
  const int ok = n_chosen == N && !chosen[0] && nack_round[0] == r1;
  fpx_destroy(ctx);
  return ok ? 0 : 1;
}
