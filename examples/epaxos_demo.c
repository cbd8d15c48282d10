/* epaxos_demo.c -- the EPaxos rows of include/fpx.h + include/fpx_depgraph.h from plain C: BASELINE.json configs[3] in
 * small.  n = 5 replicas, 8 keys, one tick of 400 fresh single-key commands: every replica scans its conflicts in its
 * own delivery order (fpx_epx_preaccept), fast-path commands are committed at once, the others go through the Accept
 * phase (fpx_epx_accept: f + 1 AcceptOks commit), every committed triple enters the dependency graph
 * (fpx_depgraph_commit_epx) and the graph executes them: conflicting commands in the same order at every replica.
 *
 *   gcc -std=c11 -O2 examples/epaxos_demo.c -Iinclude -Lfrankenpaxos_amd/csrc -lfpx \
 *       -Wl,-rpath,$PWD/frankenpaxos_amd/csrc -Wl,-rpath,/opt/rocm/lib -o /tmp/epaxos_demo && /tmp/epaxos_demo
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fpx.h"
#include "fpx_depgraph.h"

#define N 5
#define M 400
#define KEYS 8

static unsigned long long rng_state = 42;
static unsigned rnd(unsigned mod) {  /* splitmix64 */
  unsigned long long z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull, z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (unsigned)((z ^ (z >> 31)) % mod);
}

int main(void) {
  fpx_epx_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.num_replicas = N, cfg.num_keys = KEYS, cfg.num_instances = 1024;  /* a command log: the Accept phase needs it */
  fpx_epx* epx = NULL;
  int32_t st = fpx_epx_create(&cfg, &epx);
  if (st != FPX_OK) {
    printf("fpx_epx_create: %s\n", fpx_strerror(st));  /* FPX_ENODEVICE without an MI355X: there is no CPU path */
    return st == FPX_ENODEVICE ? 77 : 1;
  }
  static int32_t leader[M], number[M], key[M], triple[M], rank[N * M], deps[M * N], ldeps[M * N], own[M * 2];
  static uint8_t is_set[M], resp[M], fast[M];
  int32_t next[N] = {0, 0, 0, 0, 0};
  for (int i = 0; i < M; ++i) {
    leader[i] = (int32_t)rnd(N), number[i] = next[leader[i]]++, key[i] = (int32_t)rnd(KEYS), is_set[i] = (uint8_t)rnd(2);
    triple[i] = 1000 + i;
    /* thrifty fast quorum: the leader asks n - 2 of the other replicas (Replica.scala:705-706) */
    const int skip = (leader[i] + 1 + (int)rnd(N - 1)) % N;
    resp[i] = (uint8_t)(((1u << N) - 1u) & ~(1u << leader[i]) & ~(1u << skip));
  }
  /* every replica's delivery order: replica r sees the tick rotated by 37 r positions (FIFO per leader is kept: a
   * leader numbers its instances in tick order, and a rotation preserves each leader's order except at the wrap) */
  for (int r = 0; r < N; ++r)
    for (int i = 0; i < M; ++i) rank[r * M + i] = (i + 37 * r) % M;
  st = fpx_epx_preaccept(epx, M, leader, number, key, is_set, resp, NULL, rank, triple, fast, deps, ldeps, own);
  if (st != FPX_OK) return 1;
  int n_fast = 0;
  for (int i = 0; i < M; ++i) n_fast += fast[i];
  printf("tick: status %d, %d of %d commands committed on the fast path\n", st, n_fast, M);
  /* the slow path: the leader proposes the union of the answers in its default ballot to f = 2 other replicas */
  static int32_t s_leader[M], s_number[M], s_zero[M], s_triple[M], s_key[M];
  static uint8_t s_set[M], s_tgt[M], s_done[M];
  int ns = 0;
  for (int i = 0; i < M; ++i)
    if (!fast[i]) {
      s_leader[ns] = leader[i], s_number[ns] = number[i], s_zero[ns] = 0, s_triple[ns] = triple[i], s_key[ns] = key[i];
      s_set[ns] = is_set[i];
      s_tgt[ns] = (uint8_t)((1u << ((leader[i] + 1) % N)) | (1u << ((leader[i] + 2) % N)));
      ++ns;
    }
  if (ns > 0 && (st = fpx_epx_accept(epx, ns, s_leader, s_number, s_zero, s_leader, s_triple, s_key, s_set, s_tgt, NULL,
                                     NULL, NULL, NULL, s_done)) != FPX_OK)
    return 1;
  int n_slow = 0;
  for (int i = 0; i < ns; ++i) n_slow += s_done[i];
  printf("accept phase: %d of %d slow-path commands committed\n", n_slow, ns);
  /* Replica.commit -> dependencyGraph.commit, then appendExecute (Replica.scala:859-917): every command is committed
   * by now with the dependencies of the tick's decision */
  fpx_depgraph_config gc = {FPX_DG_ZIGZAG, N, 0};
  fpx_depgraph* graph = NULL;
  if (fpx_depgraph_create(&gc, &graph) != FPX_OK) return 1;
  if (fpx_depgraph_commit_epx(graph, M, leader, number, NULL, deps, own, 2, NULL) != FPX_OK) return 1;
  int64_t n_exec = 0, n_comp = 0, n_block = 0;
  if (fpx_depgraph_execute(graph, -1, &n_exec, &n_comp, &n_block) != FPX_OK) return 1;
  static int32_t ex_leader[M], ex_id[M];
  if (fpx_depgraph_read_result(graph, ex_leader, ex_id, NULL, NULL, NULL) != FPX_OK) return 1;
  /* every dependency of a command that is not in its own strongly connected component runs before it: check the
   * watermark part directly -- position of (l, x) in the execution order */
  static int pos[N][M];
  for (int e = 0; e < (int)n_exec; ++e) pos[ex_leader[e]][ex_id[e]] = e + 1;
  int all_there = n_exec == M;
  for (int l = 0; l < N && all_there; ++l)
    for (int x = 0; x < next[l]; ++x) all_there = all_there && pos[l][x] > 0;
  /* (the zigzag variant reports the next, not yet committed instance of every leader column as a blocker) */
  printf("dependency graph: %lld of %d commands executed in %lld components, %lld blockers (one per leader: the next instance)\n",
         (long long)n_exec, M, (long long)n_comp, (long long)n_block);
  fpx_depgraph_destroy(graph);
  fpx_epx_destroy(epx);
  return (st == FPX_OK && n_fast + n_slow == M && all_there && n_block == N) ? 0 : 1;
}
