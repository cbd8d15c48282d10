/*
 * fpx_oracle.c -- TEST INFRASTRUCTURE ONLY (see fpx_oracle.h).
 *
 * Plain C, single-threaded, message-at-a-time restatement of the reference handlers.  Paths in
 * comments are relative to /root/reference/shared/src/main/scala/frankenpaxos/.
 *
 * Parity status: quorums / round system / BufferMap pinned by the reference's known-answer tests
 * (tests/test_oracle_golden.py); vote + tally "parity unpinned" by the reference (no golden vectors
 * exist, JVM cannot run here) -- anchored on citations, micro-traces and the safety invariant.
 */
#include "fpx_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ============================================================================================== */
/* roundsystem.ClassicRoundRobin                                                                   */
/* ============================================================================================== */

/* RoundSystem.scala:63  `override def leader(round: Round): LeaderIndex = round % n` */
int fpo_round_leader(int n, int round) { return round % n; }

/* RoundSystem.scala:66-81 */
int fpo_next_classic_round(int n, int leader_index, int round) {
  if (round < 0) {
    return leader_index;
  } else {
    int smallest_multiple_of_n = n * (round / n);
    int offset = leader_index % n;
    if (smallest_multiple_of_n + offset > round) {
      return smallest_multiple_of_n + offset;
    } else {
      return smallest_multiple_of_n + n + offset;
    }
  }
}

uint64_t fpo_splitmix64(uint64_t* state) {
  uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

/* ============================================================================================== */
/* quorum systems over Set[Int]                                                                    */
/* ============================================================================================== */

enum { QS_SIMPLE_MAJORITY = 1, QS_GRID = 2, QS_UNANIMOUS = 3 };

struct fpo_qs {
  int kind;
  int n;        /* |members| (deduplicated) */
  int* members; /* set semantics: unique ids */
  int rows, cols;
  int* grid; /* rows x cols */
};

static int contains(const int* xs, int n, int x) {
  for (int i = 0; i < n; ++i)
    if (xs[i] == x) return 1;
  return 0;
}

/* dedup into a fresh array (Scala Set construction) */
static int* to_set(const int* xs, int n, int* out_n) {
  int* s = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  int m = 0;
  for (int i = 0; i < n; ++i)
    if (!contains(s, m, xs[i])) s[m++] = xs[i];
  *out_n = m;
  return s;
}

static int subset_of(const int* xs, int n, const int* ys, int m) {
  for (int i = 0; i < n; ++i)
    if (!contains(ys, m, xs[i])) return 0;
  return 1;
}

fpo_qs* fpo_qs_simple_majority(const int* members, int n) {
  /* SimpleMajority.scala:23-26 require(!members.isEmpty) */
  if (n <= 0) return NULL;
  fpo_qs* qs = (fpo_qs*)calloc(1, sizeof(fpo_qs));
  qs->kind = QS_SIMPLE_MAJORITY;
  qs->members = to_set(members, n, &qs->n);
  return qs;
}

fpo_qs* fpo_qs_unanimous_writes(const int* members, int n) {
  /* UnanimousWrites.scala:21-24 */
  if (n <= 0) return NULL;
  fpo_qs* qs = (fpo_qs*)calloc(1, sizeof(fpo_qs));
  qs->kind = QS_UNANIMOUS;
  qs->members = to_set(members, n, &qs->n);
  return qs;
}

fpo_qs* fpo_qs_grid(const int* grid, int rows, int cols) {
  /* Grid.scala:9-17: non-empty, equal sized rows (guaranteed by the rows x cols layout) */
  if (rows <= 0 || cols < 0) return NULL;
  fpo_qs* qs = (fpo_qs*)calloc(1, sizeof(fpo_qs));
  qs->kind = QS_GRID;
  qs->rows = rows;
  qs->cols = cols;
  qs->grid = (int*)malloc(sizeof(int) * (size_t)(rows * cols > 0 ? rows * cols : 1));
  memcpy(qs->grid, grid, sizeof(int) * (size_t)(rows * cols));
  qs->members = to_set(grid, rows * cols, &qs->n); /* Grid.scala:26 nodes = gridSetSet.flatten */
  return qs;
}

void fpo_qs_free(fpo_qs* qs) {
  if (!qs) return;
  free(qs->members);
  free(qs->grid);
  free(qs);
}

int fpo_qs_nodes(const fpo_qs* qs, int* out) {
  if (out) memcpy(out, qs->members, sizeof(int) * (size_t)qs->n);
  return qs->n;
}

static int count_members(const fpo_qs* qs, const int* xs, int n) {
  /* nodes.count(members.contains) over a Set => count unique ids that are members */
  int m;
  int* s = to_set(xs, n, &m);
  int c = 0;
  for (int i = 0; i < m; ++i)
    if (contains(qs->members, qs->n, s[i])) ++c;
  free(s);
  return c;
}

/* Grid.scala:36-41 / :52-53  exists(row => row.subsetOf(xs)) */
static int grid_some_row_subset(const fpo_qs* qs, const int* xs, int n) {
  for (int r = 0; r < qs->rows; ++r)
    if (subset_of(qs->grid + r * qs->cols, qs->cols, xs, n)) return 1;
  return 0;
}

/* Grid.scala:43-50 / :55-56  forall(row => row.exists(x => xs.contains(x))) */
static int grid_every_row_hit(const fpo_qs* qs, const int* xs, int n) {
  for (int r = 0; r < qs->rows; ++r) {
    int hit = 0;
    for (int c = 0; c < qs->cols; ++c)
      if (contains(xs, n, qs->grid[r * qs->cols + c])) hit = 1;
    if (!hit) return 0;
  }
  return 1;
}

int fpo_qs_is_superset_of_read_quorum(const fpo_qs* qs, const int* xs, int n) {
  switch (qs->kind) {
    case QS_SIMPLE_MAJORITY: /* SimpleMajority.scala:51-52, quorumSize = members.size / 2 + 1 (:30) */
      return count_members(qs, xs, n) >= qs->n / 2 + 1;
    case QS_UNANIMOUS: /* UnanimousWrites.scala:53-54 nodes.exists(members.contains) */
      return count_members(qs, xs, n) >= 1;
    case QS_GRID:
      return grid_some_row_subset(qs, xs, n);
  }
  return -1;
}

int fpo_qs_is_superset_of_write_quorum(const fpo_qs* qs, const int* xs, int n) {
  switch (qs->kind) {
    case QS_SIMPLE_MAJORITY: /* SimpleMajority.scala:54-55 */
      return fpo_qs_is_superset_of_read_quorum(qs, xs, n);
    case QS_UNANIMOUS: /* UnanimousWrites.scala:56-57 members.subsetOf(nodes) */
      return subset_of(qs->members, qs->n, xs, n);
    case QS_GRID:
      return grid_every_row_hit(qs, xs, n);
  }
  return -1;
}

int fpo_qs_is_read_quorum(const fpo_qs* qs, const int* xs, int n) {
  /* require(nodes.subsetOf(members)): SimpleMajority.scala:42-45, Grid.scala:37-40,
   * UnanimousWrites.scala:37-40 */
  if (!subset_of(xs, n, qs->members, qs->n)) return -1;
  switch (qs->kind) {
    case QS_SIMPLE_MAJORITY: /* :46 nodes.size >= quorumSize */
      return count_members(qs, xs, n) >= qs->n / 2 + 1;
    case QS_UNANIMOUS: /* :41 !nodes.isEmpty */
      return n > 0;
    case QS_GRID:
      return grid_some_row_subset(qs, xs, n);
  }
  return -1;
}

int fpo_qs_is_write_quorum(const fpo_qs* qs, const int* xs, int n) {
  if (!subset_of(xs, n, qs->members, qs->n)) return -1;
  switch (qs->kind) {
    case QS_SIMPLE_MAJORITY: /* SimpleMajority.scala:49 = isReadQuorum */
      return count_members(qs, xs, n) >= qs->n / 2 + 1;
    case QS_UNANIMOUS: /* UnanimousWrites.scala:50 nodes == members */
      return count_members(qs, xs, n) == qs->n;
    case QS_GRID:
      return grid_every_row_hit(qs, xs, n);
  }
  return -1;
}

static void shuffle(int* xs, int n, uint64_t* rng) {
  for (int i = n - 1; i > 0; --i) {
    int j = (int)(fpo_splitmix64(rng) % (uint64_t)(i + 1));
    int t = xs[i];
    xs[i] = xs[j];
    xs[j] = t;
  }
}

int fpo_qs_random_read_quorum(const fpo_qs* qs, uint64_t* rng, int* out) {
  switch (qs->kind) {
    case QS_SIMPLE_MAJORITY: { /* SimpleMajority.scala:36-37 shuffle(members).take(quorumSize) */
      int* tmp = (int*)malloc(sizeof(int) * (size_t)qs->n);
      memcpy(tmp, qs->members, sizeof(int) * (size_t)qs->n);
      shuffle(tmp, qs->n, rng);
      int k = qs->n / 2 + 1;
      memcpy(out, tmp, sizeof(int) * (size_t)k);
      free(tmp);
      return k;
    }
    case QS_UNANIMOUS: /* UnanimousWrites.scala:32 shuffle(members).take(1) */
      out[0] = qs->members[fpo_splitmix64(rng) % (uint64_t)qs->n];
      return 1;
    case QS_GRID: { /* Grid.scala:28 a random row */
      int r = (int)(fpo_splitmix64(rng) % (uint64_t)qs->rows);
      memcpy(out, qs->grid + r * qs->cols, sizeof(int) * (size_t)qs->cols);
      return qs->cols;
    }
  }
  return 0;
}

int fpo_qs_random_write_quorum(const fpo_qs* qs, uint64_t* rng, int* out) {
  switch (qs->kind) {
    case QS_SIMPLE_MAJORITY: /* SimpleMajority.scala:39 */
      return fpo_qs_random_read_quorum(qs, rng, out);
    case QS_UNANIMOUS: /* UnanimousWrites.scala:34 members */
      memcpy(out, qs->members, sizeof(int) * (size_t)qs->n);
      return qs->n;
    case QS_GRID: { /* Grid.scala:30-33 a random column */
      int c = (int)(fpo_splitmix64(rng) % (uint64_t)qs->cols);
      for (int r = 0; r < qs->rows; ++r) out[r] = qs->grid[r * qs->cols + c];
      return qs->rows;
    }
  }
  return 0;
}

/* ============================================================================================== */
/* util.BufferMap + the replica's log prefix                                                        */
/* ============================================================================================== */

struct fpo_log {
  int grow_size;
  int watermark;   /* BufferMap.scala:14 */
  int largest_key; /* :17 */
  int size;        /* buffer.size */
  int cap;
  int* val;
  unsigned char* some;
  int executed_watermark; /* Replica.scala:214 */
  int num_chosen;         /* Replica.scala:219 */
};

fpo_log* fpo_log_new(int grow_size) {
  fpo_log* l = (fpo_log*)calloc(1, sizeof(fpo_log));
  l->grow_size = grow_size;
  l->largest_key = -1;
  l->size = grow_size; /* Buffer.fill(growSize)(None) :10-11 */
  l->cap = grow_size > 0 ? grow_size : 1;
  l->val = (int*)calloc((size_t)l->cap, sizeof(int));
  l->some = (unsigned char*)calloc((size_t)l->cap, 1);
  return l;
}

void fpo_log_free(fpo_log* l) {
  if (!l) return;
  free(l->val);
  free(l->some);
  free(l);
}

static void log_pad(fpo_log* l, int len) { /* :23-27 */
  if (len > l->cap) {
    int cap = l->cap;
    while (cap < len) cap *= 2;
    l->val = (int*)realloc(l->val, sizeof(int) * (size_t)cap);
    l->some = (unsigned char*)realloc(l->some, (size_t)cap);
    memset(l->some + l->cap, 0, (size_t)(cap - l->cap));
    l->cap = cap;
  }
  if (l->size < len) l->size = len;
}

int fpo_log_get(const fpo_log* l, int key, int* value) { /* :29-35 */
  int k = key - l->watermark;
  if (k < 0 || k >= l->size) return 0;
  if (!l->some[k]) return 0;
  if (value) *value = l->val[k];
  return 1;
}

void fpo_log_put(fpo_log* l, int key, int value) { /* :37-51 */
  if (key > l->largest_key) l->largest_key = key;
  int k = key - l->watermark;
  if (k < 0) return;
  if (k >= l->size) log_pad(l, k + 1 + l->grow_size);
  l->val[k] = value;
  l->some[k] = 1;
}

void fpo_log_garbage_collect(fpo_log* l, int watermark) { /* :55-62 */
  if (watermark <= l->watermark) return;
  int drop = watermark - l->watermark;
  if (drop > l->size) drop = l->size;
  memmove(l->val, l->val + drop, sizeof(int) * (size_t)(l->size - drop));
  memmove(l->some, l->some + drop, (size_t)(l->size - drop));
  memset(l->some + (l->size - drop), 0, (size_t)drop);
  l->size -= drop;
  l->watermark = watermark;
}

int fpo_log_chosen(fpo_log* l, int slot, int value) {
  /* Replica.scala:580-590: already present -> ignore; else put, numChosen += 1, executeLog() */
  if (fpo_log_get(l, slot, NULL)) return l->executed_watermark;
  fpo_log_put(l, slot, value);
  l->num_chosen += 1;
  /* Replica.scala:394-404: execute the contiguous prefix */
  while (fpo_log_get(l, l->executed_watermark, NULL)) l->executed_watermark += 1;
  return l->executed_watermark;
}

int fpo_log_executed_watermark(const fpo_log* l) { return l->executed_watermark; }
int fpo_log_largest_key(const fpo_log* l) { return l->largest_key; }

/* ============================================================================================== */
/* the Phase-2 system                                                                               */
/* ============================================================================================== */

/* ProxyLeader.states: Map[SlotRound, State]  ProxyLeader.scala:87-99,135 */
typedef struct {
  uint64_t key;  /* slot(Start) << 32 | round ; valid iff state != 0 */
  int end;       /* slotEndExclusive: slot + 1 for a single-slot tally (mencius SlotRound, ProxyLeader.scala:86-90) */
  int is_range;  /* PendingPhase2aNoopRange (mencius/ProxyLeader.scala:99-104) */
  int state;     /* 0 empty, 1 Pending, 2 Done */
  int value;     /* pending.phase2a.commandBatchOrNoop */
  uint64_t v[4]; /* phase2bs.keys as a set of acceptor bits */
  uint64_t* rv;  /* PendingPhase2aNoopRange.phase2bNoopRanges: one 256-bit set of acceptor indices per acceptor
                    group (num_groups x 4 words), mencius/ProxyLeader.scala:99-104, 295-301 */
  int seq;       /* insertion order (for fpo_read_tally) */
} tally_t;

struct fpo_sys {
  fpo_config cfg;
  int ngroups; /* num_leader_groups * num_groups */
  /* acceptors: scalars [ngroups][R]; cells [S][R] (the row of slot s belongs to group(s)) */
  int* promised;       /* Acceptor.scala:95  var round = -1 */
  int* max_voted_slot; /* Acceptor.scala:104 */
  int* vote_round;     /* states(slot).voteRound, -1 = no entry (Acceptor.scala:98) */
  int* vote_value;
  int* ballot;         /* PER_SLOT mode only */
  /* proxy leader */
  tally_t* tab;
  size_t tab_cap, tab_n;
  int seq;
  int err_index, err_slot, err_round;
  fpo_log* log; /* the replica's log (f1), created on first use */
};

int fpo_config_check(const fpo_config* c) {
  if (!c) return FPO_EINVAL;
  if (c->num_slots < 1) return FPO_EINVAL;
  if (c->num_replicas < 1 || c->num_replicas > 256) return FPO_EINVAL;
  if (c->num_groups < 1 || c->num_leader_groups < 1) return FPO_EINVAL;
  if (c->num_leaders < 1) return FPO_EINVAL;
  if (c->tally_ways < 1 || c->tally_ways > 8) return FPO_EINVAL;
  if (c->ballot_mode != 0 && c->ballot_mode != 1) return FPO_EINVAL;
  int total = c->replicas_total ? c->replicas_total : c->num_replicas;
  if (total < 1 || total > 256) return FPO_EINVAL;
  if (c->replica_base < 0 || (c->replica_base & 3) || c->replica_base + c->num_replicas > total)
    return FPO_EINVAL;
  switch (c->quorum_kind) {
    case 0: /* threshold f+1 of the group: need 1 <= f+1 <= total */
      if (c->f < 0 || c->f + 1 > total) return FPO_EINVAL;
      break;
    case 1:
    case 3:
      break;
    case 2:
      if (c->grid_rows < 1 || c->grid_cols < 1 || c->grid_rows * c->grid_cols != total)
        return FPO_EINVAL;
      break;
    default:
      return FPO_EINVAL;
  }
  return FPO_OK;
}

int fpo_group_of_slot(const fpo_config* c, int slot) {
  /* multipaxos/ProxyLeader.scala:190  slot % numAcceptorGroups ;
   * mencius/ProxyLeader.scala:169-176,231-234  leader group = slot % numLeaderGroups,
   * acceptor group = (slot / numLeaderGroups) % numAcceptorGroups */
  int lg = slot % c->num_leader_groups;
  int ag = (slot / c->num_leader_groups) % c->num_groups;
  return lg * c->num_groups + ag;
}

static size_t cells(const fpo_sys* s) { return (size_t)s->cfg.num_slots * (size_t)s->cfg.num_replicas; }

void fpo_sys_reset(fpo_sys* s) {
  size_t nsc = (size_t)s->ngroups * (size_t)s->cfg.num_replicas;
  for (size_t i = 0; i < nsc; ++i) s->promised[i] = -1, s->max_voted_slot[i] = -1;
  size_t nc = cells(s);
  for (size_t i = 0; i < nc; ++i) s->vote_round[i] = -1, s->vote_value[i] = -1;
  if (s->ballot)
    for (size_t i = 0; i < nc; ++i) s->ballot[i] = -1;
  for (size_t i = 0; i < s->tab_cap; ++i) free(s->tab[i].rv);
  memset(s->tab, 0, sizeof(tally_t) * s->tab_cap);
  s->tab_n = 0;
  s->seq = 0;
  s->err_index = s->err_slot = s->err_round = -1;
  if (s->log) fpo_log_free(s->log);
  s->log = NULL;
}

fpo_sys* fpo_sys_new(const fpo_config* cfg) {
  if (fpo_config_check(cfg) != FPO_OK) return NULL;
  fpo_sys* s = (fpo_sys*)calloc(1, sizeof(fpo_sys));
  s->cfg = *cfg;
  if (!s->cfg.replicas_total) s->cfg.replicas_total = cfg->num_replicas;
  s->ngroups = cfg->num_leader_groups * cfg->num_groups;
  size_t nsc = (size_t)s->ngroups * (size_t)cfg->num_replicas;
  s->promised = (int*)malloc(sizeof(int) * nsc);
  s->max_voted_slot = (int*)malloc(sizeof(int) * nsc);
  size_t nc = cells(s);
  s->vote_round = (int*)malloc(sizeof(int) * nc);
  s->vote_value = (int*)malloc(sizeof(int) * nc);
  s->ballot = cfg->ballot_mode == 1 ? (int*)malloc(sizeof(int) * nc) : NULL;
  s->tab_cap = 1024;
  s->tab = (tally_t*)calloc(s->tab_cap, sizeof(tally_t));
  fpo_sys_reset(s);
  return s;
}

void fpo_sys_free(fpo_sys* s) {
  if (!s) return;
  free(s->promised);
  free(s->max_voted_slot);
  free(s->vote_round);
  free(s->vote_value);
  free(s->ballot);
  for (size_t i = 0; i < s->tab_cap; ++i) free(s->tab[i].rv);
  free(s->tab);
  if (s->log) fpo_log_free(s->log);
  free(s);
}

/* ---- acceptor ---------------------------------------------------------------------------------- */

int fpo_acceptor_handle_phase2a(fpo_sys* s, int group, int replica, int slot, int round, int value,
                                int* reply_round) {
  const int R = s->cfg.num_replicas;
  size_t cell = (size_t)slot * (size_t)R + (size_t)replica;
  if (s->cfg.ballot_mode == 0) {
    int* my_round = &s->promised[(size_t)group * R + replica];
    /* Acceptor.scala:192-200  if (phase2a.round < round) { leader.send(Nack(round = round)); return } */
    if (round < *my_round) {
      *reply_round = *my_round;
      return 0;
    }
    /* :204-209  round = phase2a.round; states(slot) = State(round, value); maxVotedSlot = max(..) */
    *my_round = round;
  } else {
    /* per-instance ballot: epaxos/Replica.scala:1443-1448 (accept.ballot < ballot -> nack) and
     * :1488-1497 (entry := AcceptedEntry(ballot = accept.ballot, voteBallot = accept.ballot, ..)) */
    if (round < s->ballot[cell]) {
      *reply_round = s->ballot[cell];
      return 0;
    }
    s->ballot[cell] = round;
  }
  s->vote_round[cell] = round;
  s->vote_value[cell] = value;
  int* mvs = &s->max_voted_slot[(size_t)group * R + replica];
  if (slot > *mvs) *mvs = slot;
  /* :211-219  Phase2b(groupIndex, acceptorIndex, slot = phase2a.slot, round = round) */
  *reply_round = round;
  return 1;
}

int fpo_acceptor_handle_phase1a(fpo_sys* s, int group, int replica, int round, int chosen_watermark,
                                int* reply_round) {
  const int R = s->cfg.num_replicas;
  if (s->cfg.ballot_mode == 0) {
    int* my_round = &s->promised[(size_t)group * R + replica];
    /* Acceptor.scala:155-162  if (phase1a.round < round) { Nack(round); return } */
    if (round < *my_round) {
      *reply_round = *my_round;
      return 0;
    }
    /* :166  round = phase1a.round */
    *my_round = round;
    *reply_round = round;
    return 1;
  }
  /* PER_SLOT generalisation: the promise is recorded in every cell of the group at or above the
   * watermark (the slots a Phase1b reports on, :171-180); a cell already at a higher ballot keeps
   * it.  The acceptor answers Phase1b iff no such cell is ahead of the leader. */
  int ahead = -1;
  for (int slot = chosen_watermark < 0 ? 0 : chosen_watermark; slot < s->cfg.num_slots; ++slot) {
    if (fpo_group_of_slot(&s->cfg, slot) != group) continue;
    size_t cell = (size_t)slot * R + replica;
    if (s->ballot[cell] > round) {
      if (s->ballot[cell] > ahead) ahead = s->ballot[cell];
    } else {
      s->ballot[cell] = round;
    }
  }
  if (ahead >= 0) {
    *reply_round = ahead;
    return 0;
  }
  *reply_round = round;
  return 1;
}

/* ---- quorum predicate on a 256-bit acceptor set ----------------------------------------------- */

static int popcount256(const uint64_t v[4]) {
  return __builtin_popcountll(v[0]) + __builtin_popcountll(v[1]) + __builtin_popcountll(v[2]) +
         __builtin_popcountll(v[3]);
}

static void set_bit(uint64_t* v, int j) { v[j >> 6] |= 1ull << (j & 63); }

static int test_bit(const uint64_t v[4], int j) { return (int)((v[j >> 6] >> (j & 63)) & 1u); }

static int has_foreign(const fpo_config* c, const uint64_t v[4]) {
  int total = c->replicas_total ? c->replicas_total : c->num_replicas;
  for (int j = total; j < 256; ++j)
    if (test_bit(v, j)) return 1;
  return 0;
}

static void members_only(const fpo_config* c, const uint64_t v[4], uint64_t out[4]) {
  int total = c->replicas_total ? c->replicas_total : c->num_replicas;
  for (int w = 0; w < 4; ++w) {
    int lo = w * 64;
    uint64_t m = total >= lo + 64 ? ~0ull : (total <= lo ? 0ull : ((1ull << (total - lo)) - 1));
    out[w] = v[w] & m;
  }
}

int fpo_sys_is_write_quorum(const fpo_config* c, const uint64_t nodes[4], int strict) {
  int total = c->replicas_total ? c->replicas_total : c->num_replicas;
  if (strict && has_foreign(c, nodes)) return -1; /* require(nodes.subsetOf(members)) */
  uint64_t x[4];
  members_only(c, nodes, x);
  switch (c->quorum_kind) {
    case 0: /* ProxyLeader.scala:238  phase2bs.size < config.f + 1 -> wait */
      return popcount256(x) >= c->f + 1;
    case 1: /* SimpleMajority.scala:30,46,49 */
      return popcount256(x) >= total / 2 + 1;
    case 2: /* Grid.scala:43-50 with node (row, col) = bit row * cols + col */
      for (int r = 0; r < c->grid_rows; ++r) {
        int hit = 0;
        for (int col = 0; col < c->grid_cols; ++col) hit |= test_bit(x, r * c->grid_cols + col);
        if (!hit) return 0;
      }
      return 1;
    case 3: /* UnanimousWrites.scala:50 / :57 */
      return popcount256(x) == total;
  }
  return -1;
}

int fpo_sys_is_read_quorum(const fpo_config* c, const uint64_t nodes[4], int strict) {
  int total = c->replicas_total ? c->replicas_total : c->num_replicas;
  if (strict && has_foreign(c, nodes)) return -1;
  uint64_t x[4];
  members_only(c, nodes, x);
  switch (c->quorum_kind) {
    case 0: /* read quorum of a (f+1)-threshold write system over n: n - f  (intersection) */
      return popcount256(x) >= total - c->f;
    case 1:
      return popcount256(x) >= total / 2 + 1;
    case 2: /* Grid.scala:36-41 some row fully contained */
      for (int r = 0; r < c->grid_rows; ++r) {
        int all = 1;
        for (int col = 0; col < c->grid_cols; ++col) all &= test_bit(x, r * c->grid_cols + col);
        if (all) return 1;
      }
      return 0;
    case 3: /* UnanimousWrites.scala:41 / :54 */
      return popcount256(x) >= 1;
  }
  return -1;
}

/* ---- proxy leader -------------------------------------------------------------------------------- */

static uint64_t tkey(int slot, int round) { return ((uint64_t)(uint32_t)slot << 32) | (uint32_t)round; }

static size_t thash(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return (size_t)k;
}

static tally_t* tab_find2(fpo_sys* s, uint64_t key, int end) {
  size_t mask = s->tab_cap - 1;
  for (size_t i = thash(key) & mask;; i = (i + 1) & mask) {
    if (s->tab[i].state == 0) return NULL;
    if (s->tab[i].key == key && s->tab[i].end == end) return &s->tab[i];
  }
}

static tally_t* tab_find(fpo_sys* s, uint64_t key) { return tab_find2(s, key, (int)(key >> 32) + 1); }

static void tab_grow(fpo_sys* s) {
  tally_t* old = s->tab;
  size_t old_cap = s->tab_cap;
  s->tab_cap *= 2;
  s->tab = (tally_t*)calloc(s->tab_cap, sizeof(tally_t));
  size_t mask = s->tab_cap - 1;
  for (size_t i = 0; i < old_cap; ++i) {
    if (old[i].state == 0) continue;
    size_t j = thash(old[i].key) & mask;
    while (s->tab[j].state != 0) j = (j + 1) & mask;
    s->tab[j] = old[i];
  }
  free(old);
}

static tally_t* tab_insert(fpo_sys* s, uint64_t key) {
  if ((s->tab_n + 1) * 2 > s->tab_cap) tab_grow(s);
  size_t mask = s->tab_cap - 1;
  size_t i = thash(key) & mask;
  while (s->tab[i].state != 0) i = (i + 1) & mask;
  memset(&s->tab[i], 0, sizeof(tally_t));
  s->tab[i].key = key;
  s->tab[i].end = (int)(key >> 32) + 1;
  s->tab[i].seq = s->seq++;
  s->tab_n++;
  return &s->tab[i];
}

/* GC extension of fpx.h (fpx_proxy_forget; not in the reference): drop every single-slot tally of the
 * slots [first_slot, first_slot + count) and every noop-range tally whose range lies inside that window */
int fpo_proxy_forget(fpo_sys* s, int32_t first_slot, int32_t count) {
  if (first_slot < 0 || count < 0 || (int64_t)first_slot + count > s->cfg.num_slots) return FPO_EINVAL;
  tally_t* old = s->tab;
  size_t old_cap = s->tab_cap;
  s->tab = (tally_t*)calloc(s->tab_cap, sizeof(tally_t));
  s->tab_n = 0;
  size_t mask = s->tab_cap - 1;
  for (size_t i = 0; i < old_cap; ++i) {
    if (old[i].state == 0) continue;
    int slot = (int)(old[i].key >> 32);
    if (!old[i].is_range && slot >= first_slot && slot < first_slot + count) continue;
    if (old[i].is_range && slot >= first_slot && (int64_t)old[i].end <= (int64_t)first_slot + count) {
      free(old[i].rv);
      continue;
    }
    size_t j = thash(old[i].key) & mask;
    while (s->tab[j].state != 0) j = (j + 1) & mask;
    s->tab[j] = old[i];
    s->tab_n++;
  }
  free(old);
  return FPO_OK;
}

/* GC extension of fpx.h (fpx_recycle_slots; not in the reference): the acceptors' `states` lose their entries of
 * the slots [first_slot, first_slot + count) (Acceptor.scala:98 -- as if `states -= slot`), the proxy leader its
 * tallies; rounds / ballots stay */
int fpo_recycle_slots(fpo_sys* s, int32_t first_slot, int32_t count) {
  if (first_slot < 0 || count < 0 || (int64_t)first_slot + count > s->cfg.num_slots) return FPO_EINVAL;
  const int R = s->cfg.num_replicas;
  for (int sl = first_slot; sl < first_slot + count; ++sl)
    for (int r = 0; r < R; ++r) s->vote_round[(size_t)sl * R + r] = -1, s->vote_value[(size_t)sl * R + r] = -1;
  return fpo_proxy_forget(s, first_slot, count);
}

int fpo_proxy_handle_phase2a(fpo_sys* s, int slot, int round, int value) {
  /* ProxyLeader.scala:176-184  states.get(slotround) match { case Some(_) => ignore */
  if (tab_find(s, tkey(slot, round))) return 0;
  /* :213  states(slotround) = Pending(phase2a = phase2a, phase2bs = mutable.Map()) */
  tally_t* t = tab_insert(s, tkey(slot, round));
  t->state = 1;
  t->value = value;
  return 1;
}

int fpo_proxy_handle_phase2b(fpo_sys* s, int acceptor_bit, int slot, int round, int* chosen_value) {
  tally_t* t = tab_find(s, tkey(slot, round));
  /* ProxyLeader.scala:220-225  case None => logger.fatal(...) */
  if (!t) return -1;
  /* :227-232  case Some(Done) => ignored */
  if (t->state == 2) return 2;
  /* mencius/ProxyLeader.scala:327-333  case Some(_: PendingPhase2aNoopRange) => ignored */
  if (t->is_range) return 2;
  /* :235-237  phase2bs((groupIndex, acceptorIndex)) = phase2b   (map key => duplicates collapse) */
  t->v[acceptor_bit >> 6] |= 1ull << (acceptor_bit & 63);
  /* :238-243  non-flexible: size < f+1 -> return ; flexible: !grid.isWriteQuorum(keys) -> return */
  if (fpo_sys_is_write_quorum(&s->cfg, t->v, 0) != 1) return 0;
  /* :246-253  Chosen(slot, pending.phase2a.commandBatchOrNoop) to every replica */
  *chosen_value = t->value;
  /* :256  states(slotround) = Done */
  t->state = 2;
  return 1;
}

/* ---- Mencius noop ranges (K4) --------------------------------------------------------------------- */

/* mencius/Acceptor.scala:237-291 handlePhase2aNoopRange for acceptor `replica` of group `group`
 * (= leaderGroup * num_groups + acceptorGroup).  Returns 1 = Phase2bNoopRange, 0 = Nack(*reply_round). */
int fpo_acceptor_handle_phase2a_noop_range(fpo_sys* s, int group, int replica, int slot_start, int slot_end,
                                           int round, int* reply_round) {
  const int R = s->cfg.num_replicas, L = s->cfg.num_leader_groups, A = s->cfg.num_groups;
  int* my_round = &s->promised[(size_t)group * R + replica];
  /* :245-256 */
  if (round < *my_round) {
    *reply_round = *my_round;
    return 0;
  }
  /* :260 */
  *my_round = round;
  /* :262-267  find the first slot owned by this acceptor group */
  const int my_ag = group % A;
  int start = slot_start;
  while (start < slot_end && (start / L) % A != my_ag) start += L;
  /* :269-277  for (slot <- startSlot until slotEndExclusive by numLeaderGroups * numAcceptorGroups) */
  for (int slot = start; slot < slot_end; slot += L * A) {
    size_t cell = (size_t)slot * R + replica;
    s->vote_round[cell] = round;
    s->vote_value[cell] = -1; /* Noop */
    /* NOT in mencius/Acceptor.scala, which has no maxVotedSlot at all: the library keeps the multipaxos acceptor's
     * scalar (multipaxos/Acceptor.scala:104, :216) in every mode so that one readback / digest serves all contexts.
     * No Mencius message carries it; it is compared GPU vs. oracle as library state only. */
    int* mvs = &s->max_voted_slot[(size_t)group * R + replica];
    if (slot > *mvs) *mvs = slot;
  }
  *reply_round = round;
  return 1;
}

/* mencius/ProxyLeader.scala:255-303: 1 = new PendingPhase2aNoopRange, 0 = ignored (key already known) */
int fpo_proxy_handle_phase2a_noop_range(fpo_sys* s, int slot_start, int slot_end, int round) {
  if (tab_find2(s, tkey(slot_start, round), slot_end)) return 0;
  tally_t* t = tab_insert(s, tkey(slot_start, round));
  t->end = slot_end;
  t->is_range = 1;
  t->state = 1;
  t->value = -1;
  t->rv = (uint64_t*)calloc((size_t)s->cfg.num_groups * 4, sizeof(uint64_t)); /* :295-301 Buffer.fill(groups)(Map()) */
  return 1;
}

/* mencius/ProxyLeader.scala:355-411: vote of acceptor (acceptor_group, acceptor_index).
 * 1 = ChosenNoopRange emitted, 0 = waiting, 2 = ignored (Done or a single-slot Phase2a is pending
 * under the same key), -1 = logger.fatal */
int fpo_proxy_handle_phase2b_noop_range(fpo_sys* s, int acceptor_group, int acceptor_index, int slot_start,
                                        int slot_end, int round) {
  const int R = s->cfg.replicas_total, A = s->cfg.num_groups;
  tally_t* t = tab_find2(s, tkey(slot_start, round), slot_end);
  if (!t) return -1;                 /* :361-368 */
  if (t->state == 2) return 2;       /* :370-376 */
  if (!t->is_range) return 2;        /* :378-385 PendingPhase2a => ignored */
  /* :389-390  phase2bs(acceptorGroupIndex)(acceptorIndex) = phase2b */
  t->rv[(size_t)acceptor_group * 4 + (acceptor_index >> 6)] |= 1ull << (acceptor_index & 63);
  /* :391  if (phase2bs.exists(_.size < config.quorumSize)) return */
  for (int ag = 0; ag < A; ++ag) {
    int c = 0;
    for (int r = 0; r < R; ++r) c += test_bit(t->rv + (size_t)ag * 4, r);
    if (c < s->cfg.f + 1) return 0;
  }
  t->state = 2; /* :410 ; ChosenNoopRange(start, end) :395-407 */
  return 1;
}

/* batch-of-one entry points with the semantics of fpx_acceptor_phase2a_noop_range & co (fpx.h) */
int fpo_acceptor_phase2a_noop_range(fpo_sys* s, int32_t slot_start, int32_t slot_end, int32_t round,
                                    const uint64_t* target_masks, uint64_t* vote_bits, uint64_t* nack_bits,
                                    int32_t* nack_round) {
  const int R = s->cfg.num_replicas, A = s->cfg.num_groups, L = s->cfg.num_leader_groups;
  if (s->cfg.ballot_mode != 0 || slot_start < 0 || slot_end < slot_start || slot_end > s->cfg.num_slots || round < 0)
    return FPO_EINVAL;
  const int lg = slot_start % L; /* slotSystem.leader(slotStartInclusive), ProxyLeader.scala:275 */
  int nr = -1;
  for (int ag = 0; ag < A; ++ag) {
    uint64_t vb[4] = {0, 0, 0, 0}, nb[4] = {0, 0, 0, 0};
    for (int r = 0; r < R; ++r) {
      int bit = s->cfg.replica_base + r;
      if (target_masks && !test_bit(target_masks + (size_t)ag * 4, bit)) continue;
      int reply;
      if (fpo_acceptor_handle_phase2a_noop_range(s, lg * A + ag, r, slot_start, slot_end, round, &reply)) {
        set_bit(vb, bit);
      } else {
        set_bit(nb, bit);
        if (reply > nr) nr = reply;
      }
    }
    if (vote_bits) memcpy(vote_bits + (size_t)ag * 4, vb, sizeof vb);
    if (nack_bits) memcpy(nack_bits + (size_t)ag * 4, nb, sizeof nb);
  }
  if (nack_round) *nack_round = nr;
  return FPO_OK;
}

int fpo_proxy_open_noop_range(fpo_sys* s, int32_t slot_start, int32_t slot_end, int32_t round, uint8_t* is_new) {
  if (slot_start < 0 || slot_end < slot_start || slot_end > s->cfg.num_slots || round < 0) return FPO_EINVAL;
  int fresh = fpo_proxy_handle_phase2a_noop_range(s, slot_start, slot_end, round);
  if (is_new) *is_new = (uint8_t)fresh;
  return FPO_OK;
}

int fpo_proxy_phase2b_noop_range(fpo_sys* s, int32_t slot_start, int32_t slot_end, int32_t round,
                                 const uint64_t* vote_bits, uint8_t* newly_chosen) {
  const int R = s->cfg.replicas_total, A = s->cfg.num_groups;
  int chosen = 0, status = FPO_OK;
  for (int ag = 0; ag < A && status == FPO_OK; ++ag)
    for (int r = 0; r < R; ++r) {
      if (!test_bit(vote_bits + (size_t)ag * 4, r)) continue;
      int rc = fpo_proxy_handle_phase2b_noop_range(s, ag, r, slot_start, slot_end, round);
      if (rc == -1) {
        status = FPO_EFATAL_UNKNOWN_SLOTROUND;
        s->err_index = 0, s->err_slot = slot_start, s->err_round = round;
        break;
      }
      if (rc == 1) chosen = 1;
    }
  if (newly_chosen) *newly_chosen = (uint8_t)chosen;
  return status;
}

/* batches of n ranges, delivered in array order (fpx_*_noop_ranges of fpx.h); bitmaps n x num_groups x 4 */
int fpo_acceptor_phase2a_noop_ranges(fpo_sys* s, int32_t n, const int32_t* start, const int32_t* end,
                                     const int32_t* round, const uint64_t* target_masks, uint64_t* vote_bits,
                                     uint64_t* nack_bits, int32_t* nack_round) {
  const size_t per = (size_t)s->cfg.num_groups * 4;
  for (int i = 0; i < n; ++i)
    if (start[i] < 0 || end[i] < start[i] || end[i] > s->cfg.num_slots || round[i] < 0) {
      s->err_index = i, s->err_slot = start[i], s->err_round = round[i];
      return FPO_EINVAL;
    }
  for (int i = 0; i < n; ++i) {
    int rc = fpo_acceptor_phase2a_noop_range(s, start[i], end[i], round[i], target_masks ? target_masks + i * per : NULL,
                                             vote_bits ? vote_bits + i * per : NULL,
                                             nack_bits ? nack_bits + i * per : NULL, nack_round ? nack_round + i : NULL);
    if (rc) return rc;
  }
  return FPO_OK;
}

int fpo_proxy_open_noop_ranges(fpo_sys* s, int32_t n, const int32_t* start, const int32_t* end, const int32_t* round,
                               uint8_t* is_new) {
  for (int i = 0; i < n; ++i) {
    int rc = fpo_proxy_open_noop_range(s, start[i], end[i], round[i], is_new ? is_new + i : NULL);
    if (rc) return rc;
  }
  return FPO_OK;
}

int fpo_proxy_phase2b_noop_ranges(fpo_sys* s, int32_t n, const int32_t* start, const int32_t* end,
                                  const int32_t* round, const uint64_t* vote_bits, uint8_t* newly_chosen) {
  const size_t per = (size_t)s->cfg.num_groups * 4;
  int status = FPO_OK;
  for (int i = 0; i < n; ++i) {
    int rc = fpo_proxy_phase2b_noop_range(s, start[i], end[i], round[i], vote_bits + i * per,
                                          newly_chosen ? newly_chosen + i : NULL);
    if (rc && status == FPO_OK) {
      status = rc;
      s->err_index = i;
    }
  }
  return status;
}

/* the fused step for ranges: per range, in order, ProxyLeader.handlePhase2aNoopRange (a known key: the message is
 * ignored, not forwarded), the acceptors, then their Phase2bNoopRange's back to the proxy leader */
int fpo_noop_ranges_fused(fpo_sys* s, int32_t n, const int32_t* start, const int32_t* end, const int32_t* round,
                          const uint64_t* target_masks, uint64_t* vote_bits, uint64_t* nack_bits, int32_t* nack_round,
                          uint8_t* is_new, uint8_t* chosen) {
  const size_t per = (size_t)s->cfg.num_groups * 4;
  if (s->cfg.ballot_mode != 0) return FPO_EINVAL;
  for (int i = 0; i < n; ++i)
    if (start[i] < 0 || end[i] < start[i] || end[i] > s->cfg.num_slots || round[i] < 0) {
      s->err_index = i, s->err_slot = start[i], s->err_round = round[i];
      return FPO_EINVAL;
    }
  uint64_t* vb = (uint64_t*)malloc(sizeof(uint64_t) * per);
  for (int i = 0; i < n; ++i) {
    uint8_t fresh = 0, ch = 0;
    memset(vb, 0, sizeof(uint64_t) * per);
    if (nack_bits) memset(nack_bits + i * per, 0, sizeof(uint64_t) * per);
    if (nack_round) nack_round[i] = -1;
    fpo_proxy_open_noop_range(s, start[i], end[i], round[i], &fresh);
    if (fresh) {
      fpo_acceptor_phase2a_noop_range(s, start[i], end[i], round[i], target_masks ? target_masks + i * per : NULL, vb,
                                      nack_bits ? nack_bits + i * per : NULL, nack_round ? nack_round + i : NULL);
      fpo_proxy_phase2b_noop_range(s, start[i], end[i], round[i], vb, &ch);
    }
    if (vote_bits) memcpy(vote_bits + i * per, vb, sizeof(uint64_t) * per);
    if (is_new) is_new[i] = fresh;
    if (chosen) chosen[i] = ch;
  }
  free(vb);
  return FPO_OK;
}

/* state 0 = unknown key, 1 = Pending, 2 = Done; votes (num_groups x 4 words) of a Pending range */
int fpo_read_range_tally(fpo_sys* s, int32_t start, int32_t end, int32_t round, int32_t* state, uint64_t* vote_bits) {
  const size_t per = (size_t)s->cfg.num_groups * 4;
  tally_t* t = tab_find2(s, tkey(start, round), end);
  if (vote_bits) memset(vote_bits, 0, sizeof(uint64_t) * per);
  if (!t || !t->is_range) {
    if (state) *state = 0;
    return FPO_OK;
  }
  if (state) *state = t->state;
  if (vote_bits && t->state == 1) memcpy(vote_bits, t->rv, sizeof(uint64_t) * per);
  return FPO_OK;
}

/* ---- batch entry points (same semantics as fpx.h host entry points) --------------------------- */

static int targeted(const fpo_sys* s, const uint64_t* target_mask, int i, int r) {
  if (!target_mask) return 1;
  return test_bit(target_mask + (size_t)i * 4, s->cfg.replica_base + r);
}

static int check_msg(fpo_sys* s, int i, int slot, int round) {
  if (slot < 0 || slot >= s->cfg.num_slots || round < 0) {
    s->err_index = i, s->err_slot = slot, s->err_round = round;
    return 0;
  }
  return 1;
}


int fpo_acceptor_phase2a(fpo_sys* s, int32_t n, const int32_t* slot, const int32_t* round,
                         const int32_t* value_id, const uint64_t* target_mask, uint64_t* vote_bits,
                         uint64_t* nack_bits, int32_t* nack_round) {
  if (n < 0) return FPO_EINVAL;
  for (int i = 0; i < n; ++i)
    if (!check_msg(s, i, slot[i], round[i])) return FPO_EINVAL;
  const int R = s->cfg.num_replicas, base = s->cfg.replica_base;
  for (int i = 0; i < n; ++i) {
    uint64_t vb[4] = {0, 0, 0, 0}, nb[4] = {0, 0, 0, 0};
    int nr = -1;
    int g = fpo_group_of_slot(&s->cfg, slot[i]);
    for (int r = 0; r < R; ++r) {
      if (!targeted(s, target_mask, i, r)) continue;
      int reply;
      if (fpo_acceptor_handle_phase2a(s, g, r, slot[i], round[i], value_id[i], &reply)) {
        set_bit(vb, base + r);
      } else {
        set_bit(nb, base + r);
        if (reply > nr) nr = reply;
      }
    }
    if (vote_bits) memcpy(vote_bits + (size_t)i * 4, vb, sizeof vb);
    if (nack_bits) memcpy(nack_bits + (size_t)i * 4, nb, sizeof nb);
    if (nack_round) nack_round[i] = nr;
  }
  return FPO_OK;
}

int fpo_acceptor_phase1a(fpo_sys* s, int32_t group, int32_t round, int32_t chosen_watermark,
                         const uint64_t* target_mask, uint64_t* promised_bits, uint64_t* nack_bits) {
  if (group < 0 || group >= s->ngroups || round < 0) return FPO_EINVAL;
  uint64_t pb[4] = {0, 0, 0, 0}, nb[4] = {0, 0, 0, 0};
  if (s->cfg.ballot_mode == 1 && s->cfg.num_slots >= 4096) {
    /* PER_SLOT mode on a big window: the same per-acceptor handler (fpo_acceptor_handle_phase1a's cell
     * rule), with the loops interchanged -- slots outside, acceptors inside.  Acceptors share no state,
     * so the order in which their cells are visited cannot matter; walking an acceptor's column first
     * strides 4 R bytes per access (100 s for 28 Phase1a's on a 2^20 x 256 window, 8 s this way).
     * test_oracle_phase1a_loop_order_is_immaterial holds the two orders equal. */
    const int R = s->cfg.num_replicas;
    uint8_t tgt[256], ahead[256];
    for (int r = 0; r < R; ++r) tgt[r] = (uint8_t)(targeted(s, target_mask, 0, r) ? 1 : 0), ahead[r] = 0;
    for (int slot = chosen_watermark < 0 ? 0 : chosen_watermark; slot < s->cfg.num_slots; ++slot) {
      if (fpo_group_of_slot(&s->cfg, slot) != group) continue;
      int* row = s->ballot + (size_t)slot * R;
      for (int r = 0; r < R; ++r) {
        if (!tgt[r]) continue;
        if (row[r] > round) ahead[r] = 1; else row[r] = round;
      }
    }
    for (int r = 0; r < R; ++r)
      if (tgt[r]) set_bit(ahead[r] ? nb : pb, s->cfg.replica_base + r);
    if (promised_bits) memcpy(promised_bits, pb, sizeof pb);
    if (nack_bits) memcpy(nack_bits, nb, sizeof nb);
    return FPO_OK;
  }
  for (int r = 0; r < s->cfg.num_replicas; ++r) {
    if (!targeted(s, target_mask, 0, r)) continue;
    int reply;
    if (fpo_acceptor_handle_phase1a(s, group, r, round, chosen_watermark, &reply))
      set_bit(pb, s->cfg.replica_base + r);
    else
      set_bit(nb, s->cfg.replica_base + r);
  }
  if (promised_bits) memcpy(promised_bits, pb, sizeof pb);
  if (nack_bits) memcpy(nack_bits, nb, sizeof nb);
  return FPO_OK;
}

static int count_live(fpo_sys* s, int slot) {
  /* number of distinct rounds opened for this slot (capacity model of the dense device table) */
  int c = 0;
  for (size_t i = 0; i < s->tab_cap; ++i)
    if (s->tab[i].state != 0 && (int)(s->tab[i].key >> 32) == slot) ++c;
  return c;
}

int fpo_proxy_open(fpo_sys* s, int32_t n, const int32_t* slot, const int32_t* round,
                   const int32_t* value_id, uint8_t* is_new) {
  if (n < 0) return FPO_EINVAL;
  for (int i = 0; i < n; ++i)
    if (!check_msg(s, i, slot[i], round[i])) return FPO_EINVAL;
  int status = FPO_OK;
  for (int i = 0; i < n; ++i) {
    int fresh = fpo_proxy_handle_phase2a(s, slot[i], round[i], value_id[i]);
    if (is_new) is_new[i] = (uint8_t)fresh;
    (void)status;
  }
  return status;
}

int fpo_proxy_phase2b(fpo_sys* s, int32_t n, const int32_t* slot, const int32_t* round,
                      const uint64_t* vote_bits, uint8_t* newly_chosen, int32_t* chosen_round,
                      int32_t* chosen_value) {
  if (n < 0) return FPO_EINVAL;
  for (int i = 0; i < n; ++i)
    if (!check_msg(s, i, slot[i], round[i])) return FPO_EINVAL;
  int status = FPO_OK;
  const int total = s->cfg.replicas_total;
  for (int i = 0; i < n; ++i) {
    int chosen = 0, cv = -1;
    for (int j = 0; j < total; ++j) {
      if (!test_bit(vote_bits + (size_t)i * 4, j)) continue;
      int v;
      int rc = fpo_proxy_handle_phase2b(s, j, slot[i], round[i], &v);
      if (rc == -1) {
        if (status == FPO_OK) {
          status = FPO_EFATAL_UNKNOWN_SLOTROUND;
          s->err_index = i, s->err_slot = slot[i], s->err_round = round[i];
        }
        break;
      }
      if (rc == 1) chosen = 1, cv = v;
    }
    if (newly_chosen) newly_chosen[i] = (uint8_t)chosen;
    if (chosen_round) chosen_round[i] = chosen ? round[i] : -1;
    if (chosen_value) chosen_value[i] = chosen ? cv : -1;
  }
  return status;
}

int fpo_phase2_fused(fpo_sys* s, int32_t n, const int32_t* slot, const int32_t* round,
                     const int32_t* value_id, const uint64_t* target_mask, uint8_t* chosen,
                     int32_t* chosen_round, int32_t* chosen_value, int32_t* nack_round) {
  if (n < 0) return FPO_EINVAL;
  for (int i = 0; i < n; ++i)
    if (!check_msg(s, i, slot[i], round[i])) return FPO_EINVAL;
  const int R = s->cfg.num_replicas, base = s->cfg.replica_base;
  for (int i = 0; i < n; ++i) {
    int ch = 0, cv = -1, nr = -1;
    /* proxy leader: ProxyLeader.scala:175-215 */
    if (fpo_proxy_handle_phase2a(s, slot[i], round[i], value_id[i])) {
      int g = fpo_group_of_slot(&s->cfg, slot[i]);
      for (int r = 0; r < R; ++r) {
        if (!targeted(s, target_mask, i, r)) continue;
        int reply;
        if (fpo_acceptor_handle_phase2a(s, g, r, slot[i], round[i], value_id[i], &reply)) {
          int v;
          if (fpo_proxy_handle_phase2b(s, base + r, slot[i], reply, &v) == 1) ch = 1, cv = v;
        } else if (reply > nr) {
          nr = reply;
        }
      }
    }
    if (chosen) chosen[i] = (uint8_t)ch;
    if (chosen_round) chosen_round[i] = ch ? round[i] : -1;
    if (chosen_value) chosen_value[i] = ch ? cv : -1;
    if (nack_round) nack_round[i] = nr;
  }
  return FPO_OK;
}

/* A strict FIFO pump standing in for FakeTransport (FakeTransport.scala:89-95 send appends to the
 * message buffer; :142-159 deliverMessage removes one message and calls actor.receive).  The
 * reference's test harness delivers in random order; a FIFO drain is the deterministic schedule
 * a throughput baseline needs. */
typedef struct {
  int kind; /* 0 Phase2a -> proxy leader, 1 Phase2a -> acceptor, 2 Phase2b -> proxy leader */
  int idx, replica, slot, round, value;
} pump_msg;

int fpo_phase2_fifo_pump(fpo_sys* s, int32_t n, const int32_t* slot, const int32_t* round,
                         const int32_t* value_id, const uint64_t* target_mask, uint8_t* chosen,
                         int32_t* chosen_round, int32_t* chosen_value, int32_t* nack_round) {
  if (n < 0) return FPO_EINVAL;
  for (int i = 0; i < n; ++i)
    if (!check_msg(s, i, slot[i], round[i])) return FPO_EINVAL;
  const int R = s->cfg.num_replicas, base = s->cfg.replica_base;
  size_t cap = (size_t)n * (size_t)(2 * R + 1) + 1, head = 0, tail = 0;
  pump_msg* q = (pump_msg*)malloc(sizeof(pump_msg) * cap);
  for (int i = 0; i < n; ++i) {
    if (chosen) chosen[i] = 0;
    if (chosen_round) chosen_round[i] = -1;
    if (chosen_value) chosen_value[i] = -1;
    if (nack_round) nack_round[i] = -1;
    pump_msg m = {0, i, 0, slot[i], round[i], value_id[i]};
    q[tail++] = m; /* Leader -> ProxyLeader */
  }
  while (head < tail) {
    pump_msg m = q[head++];
    if (m.kind == 0) {
      if (!fpo_proxy_handle_phase2a(s, m.slot, m.round, m.value)) continue;
      for (int r = 0; r < R; ++r) {
        if (!targeted(s, target_mask, m.idx, r)) continue;
        pump_msg a = {1, m.idx, r, m.slot, m.round, m.value};
        q[tail++] = a;
      }
    } else if (m.kind == 1) {
      int reply;
      int g = fpo_group_of_slot(&s->cfg, m.slot);
      if (fpo_acceptor_handle_phase2a(s, g, m.replica, m.slot, m.round, m.value, &reply)) {
        pump_msg b = {2, m.idx, m.replica, m.slot, reply, 0};
        q[tail++] = b;
      } else if (nack_round && reply > nack_round[m.idx]) {
        nack_round[m.idx] = reply;
      }
    } else {
      int v;
      if (fpo_proxy_handle_phase2b(s, base + m.replica, m.slot, m.round, &v) == 1) {
        if (chosen) chosen[m.idx] = 1;
        if (chosen_round) chosen_round[m.idx] = m.round;
        if (chosen_value) chosen_value[m.idx] = v;
      }
    }
  }
  free(q);
  return FPO_OK;
}

/* ---- f1: the replica's log --------------------------------------------------------------------- */

static fpo_log* sys_log(fpo_sys* s) {
  if (!s->log) s->log = fpo_log_new(5000); /* BufferMap default growSize, BufferMap.scala:8 */
  return s->log;
}

int fpo_replica_chosen(fpo_sys* s, int32_t n, const int32_t* slot, const int32_t* value_id,
                       const uint8_t* mask, int32_t* executed_watermark, int32_t* num_chosen) {
  if (n < 0) return FPO_EINVAL;
  for (int i = 0; i < n; ++i)
    if ((!mask || mask[i]) && (slot[i] < 0 || slot[i] >= s->cfg.num_slots)) {
      s->err_index = i, s->err_slot = slot[i], s->err_round = -1;
      return FPO_EINVAL;
    }
  fpo_log* l = sys_log(s);
  /* Replica.scala:572-590 for every Chosen in delivery order */
  for (int i = 0; i < n; ++i)
    if (!mask || mask[i]) fpo_log_chosen(l, slot[i], value_id[i]);
  if (executed_watermark) *executed_watermark = l->executed_watermark;
  if (num_chosen) *num_chosen = l->num_chosen;
  return FPO_OK;
}

int fpo_replica_chosen_noop_range(fpo_sys* s, int32_t slot_start, int32_t slot_end,
                                  int32_t* executed_watermark, int32_t* num_chosen) {
  if (slot_start < 0 || slot_end > s->cfg.num_slots) return FPO_EINVAL;
  fpo_log* l = sys_log(s);
  int returned_early = 0;
  /* mencius/Replica.scala:471-484: for (slot <- start until end by config.numLeaderGroups) */
  for (int slot = slot_start; slot < slot_end; slot += s->cfg.num_leader_groups) {
    if (fpo_log_get(l, slot, NULL)) {
      returned_early = 1; /* :475-479 `return`: leaves the handler, not just this iteration */
      break;
    }
    fpo_log_put(l, slot, -1); /* :481, -1 = Noop */
    l->num_chosen += 1;             /* :482 */
  }
  /* :485 executeLog() -- only reached when the loop ran to its end */
  if (!returned_early)
    while (fpo_log_get(l, l->executed_watermark, NULL)) l->executed_watermark += 1;
  if (executed_watermark) *executed_watermark = l->executed_watermark;
  if (num_chosen) *num_chosen = l->num_chosen;
  return FPO_OK;
}

int fpo_replica_read_log(fpo_sys* s, int32_t first, int32_t count, int32_t* values, uint8_t* present) {
  fpo_log* l = sys_log(s);
  for (int i = 0; i < count; ++i) {
    int v = -1;
    int some = fpo_log_get(l, first + i, &v);
    if (values) values[i] = some ? v : -1;
    if (present) present[i] = (uint8_t)some;
  }
  return FPO_OK;
}

/* ---- f2: Phase-1 recovery ------------------------------------------------------------------------ */

int fpo_leader_phase1b_scan(fpo_sys* s, int32_t chosen_watermark, const uint64_t* quorum_masks,
                            int32_t cap, int32_t* max_slot, int32_t* safe_round, int32_t* safe_value) {
  const int R = s->cfg.num_replicas, base = s->cfg.replica_base;
  if (chosen_watermark < 0 || cap < 0) return FPO_EINVAL;
  /* Leader.scala:543-549: maxSlot = max over the Phase1b's of maxPhase1bSlot (largest slot in
   * `info`, which an acceptor fills from chosenWatermark on: Acceptor.scala:171-180), -1 if empty */
  int mx = -1;
  for (int g = 0; g < s->ngroups; ++g)
    for (int r = 0; r < R; ++r) {
      if (!test_bit(quorum_masks + (size_t)g * 4, base + r)) continue;
      int mv = s->max_voted_slot[(size_t)g * R + r];
      if (mv >= chosen_watermark && mv > mx) mx = mv;
    }
  if (max_slot) *max_slot = mx;
  /* :553-565  for (slot <- chosenWatermark to maxSlot) propose safeValue(group.values, slot) */
  for (int slot = chosen_watermark, k = 0; slot <= mx && k < cap; ++slot, ++k) {
    int g = fpo_group_of_slot(&s->cfg, slot);
    /* Leader.scala:319-328 safeValue: slotInfos = the quorum's votes in this slot; empty -> Noop;
     * else maxBy(voteRound).voteValue */
    int best_round = -1, best_value = -1;
    for (int r = 0; r < R; ++r) {
      if (!test_bit(quorum_masks + (size_t)g * 4, base + r)) continue;
      size_t cell = (size_t)slot * R + r;
      if (s->vote_round[cell] > best_round) best_round = s->vote_round[cell], best_value = s->vote_value[cell];
    }
    if (safe_round) safe_round[k] = best_round;
    if (safe_value) safe_value[k] = best_round >= 0 ? best_value : -1;
  }
  return FPO_OK;
}

int fpo_error_detail(fpo_sys* s, int32_t* index, int32_t* slot, int32_t* round) {
  if (index) *index = s->err_index;
  if (slot) *slot = s->err_slot;
  if (round) *round = s->err_round;
  return FPO_OK;
}

int fpo_read_acceptor(fpo_sys* s, int32_t group, int32_t replica, int32_t* promised,
                      int32_t* max_voted_slot, int32_t* vote_round, int32_t* vote_value,
                      int32_t* ballot) {
  const int R = s->cfg.num_replicas, S = s->cfg.num_slots;
  if (group < 0 || group >= s->ngroups || replica < 0 || replica >= R) return FPO_EINVAL;
  if (promised) *promised = s->cfg.ballot_mode == 0 ? s->promised[(size_t)group * R + replica] : -1;
  if (max_voted_slot) *max_voted_slot = s->max_voted_slot[(size_t)group * R + replica];
  for (int sl = 0; sl < S; ++sl) {
    int mine = fpo_group_of_slot(&s->cfg, sl) == group;
    size_t cell = (size_t)sl * R + replica;
    if (vote_round) vote_round[sl] = mine ? s->vote_round[cell] : -1;
    if (vote_value) vote_value[sl] = mine ? s->vote_value[cell] : -1;
    if (ballot) ballot[sl] = (mine && s->ballot) ? s->ballot[cell] : -1;
  }
  return FPO_OK;
}

/* multipaxos/Acceptor.scala:166-178
 *   info = states.iteratorFrom(phase1a.chosenWatermark).map({ case (slot, state) =>
 *            Phase1bSlotInfo(slot = slot, voteRound = state.voteRound, voteValue = state.voteValue) }).toSeq
 * `states` is a SortedMap keyed by slot that holds exactly the slots the acceptor has voted in. */
int fpo_acceptor_phase1b_info(fpo_sys* s, int32_t group, int32_t replica, int32_t chosen_watermark, int32_t cap,
                              int32_t* count, int32_t* slot, int32_t* vote_round, int32_t* vote_value) {
  const int R = s->cfg.num_replicas, S = s->cfg.num_slots;
  if (!count || cap < 0 || group < 0 || group >= s->ngroups || replica < 0 || replica >= R) return FPO_EINVAL;
  int k = 0;
  for (int sl = chosen_watermark < 0 ? 0 : chosen_watermark; sl < S; ++sl) {
    if (fpo_group_of_slot(&s->cfg, sl) != group) continue;
    const size_t cell = (size_t)sl * R + replica;
    if (s->vote_round[cell] < 0) continue; /* not in `states` */
    if (k < cap) slot[k] = sl, vote_round[k] = s->vote_round[cell], vote_value[k] = s->vote_value[cell];
    ++k;
  }
  *count = k;
  return FPO_OK;
}

int fpo_read_state(fpo_sys* s, int32_t* vote_round, int32_t* vote_value, int32_t* ballot) {
  size_t nc = cells(s);
  if (vote_round) memcpy(vote_round, s->vote_round, sizeof(int) * nc);
  if (vote_value) memcpy(vote_value, s->vote_value, sizeof(int) * nc);
  if (ballot) {
    if (s->ballot)
      memcpy(ballot, s->ballot, sizeof(int) * nc);
    else
      for (size_t i = 0; i < nc; ++i) ballot[i] = -1;
  }
  return FPO_OK;
}

int fpo_read_scalars(fpo_sys* s, int32_t* promised, int32_t* max_voted_slot) {
  size_t nsc = (size_t)s->ngroups * (size_t)s->cfg.num_replicas;
  if (promised) memcpy(promised, s->promised, sizeof(int) * nsc);
  if (max_voted_slot) memcpy(max_voted_slot, s->max_voted_slot, sizeof(int) * nsc);
  return FPO_OK;
}

/* Whole-state digests, the CPU twin of fpx_state_digest (include/fpx.h): the same order-independent sums
 * over the oracle's own arrays, so that full-size parity compares 8 words instead of gigabytes. */
static uint64_t dg_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static uint64_t dg_term(uint64_t idx, int32_t v) {
  return dg_mix64(idx * 0x9E3779B97F4A7C15ull + (uint64_t)(uint32_t)v + 1ull);
}
int fpo_state_digest(fpo_sys* s, uint64_t out[8]) {
  const size_t nc = cells(s), nsc = (size_t)s->ngroups * (size_t)s->cfg.num_replicas;
  memset(out, 0, sizeof(uint64_t) * 8);
  for (size_t i = 0; i < nc; ++i) {
    out[0] += dg_term(i, s->vote_round[i]);
    out[1] += dg_term(i, s->vote_value[i]);
    if (s->ballot) out[2] += dg_term(i, s->ballot[i]);
  }
  for (size_t i = 0; i < nsc; ++i) {
    out[3] += dg_term(i, s->promised[i]);
    out[4] += dg_term(i, s->max_voted_slot[i]);
  }
  for (size_t i = 0; i < s->tab_cap; ++i) {
    const tally_t* t = &s->tab[i];
    if (t->state == 0 || t->is_range) continue;
    const uint64_t done = t->state == 2 ? 1ull : 0ull;
    uint64_t h = dg_mix64(t->key * 0x9E3779B97F4A7C15ull + done); /* key = slot << 32 | round */
    if (!done) {
      h = dg_mix64(h ^ (uint64_t)(uint32_t)t->value);
      for (int w = 0; w < 4; ++w) h = dg_mix64(h ^ t->v[w]);
    }
    out[5] += h;
  }
  {
    int wm = 0, nch = 0;
    if (s->log) {
      const fpo_log* l = s->log;
      for (int k = 0; k <= l->largest_key; ++k) {
        int v;
        if (fpo_log_get(l, k, &v)) out[6] += dg_term((uint64_t)k, v);
      }
      wm = l->executed_watermark, nch = l->num_chosen;
    }
    out[6] += dg_mix64((uint64_t)(uint32_t)wm * 0x9E3779B97F4A7C15ull + 7ull) +
              dg_mix64((uint64_t)(uint32_t)nch * 0x9E3779B97F4A7C15ull + 11ull);
  }
  for (size_t i = 0; i < s->tab_cap; ++i) { /* the noop-range tallies */
    const tally_t* t = &s->tab[i];
    if (t->state == 0 || !t->is_range) continue;
    const uint64_t done = t->state == 2 ? 1ull : 0ull;
    const uint64_t start = t->key >> 32, round = (uint32_t)t->key;
    uint64_t h = dg_mix64(((start << 32) | (uint32_t)t->end) * 0x9E3779B97F4A7C15ull + (round << 1) + done);
    if (!done)
      for (int w = 0; w < s->cfg.num_groups * 4; ++w) h = dg_mix64(h ^ t->rv[w]);
    out[7] += h;
  }
  return FPO_OK;
}

int fpo_read_tally(fpo_sys* s, int32_t slot, int32_t* num_entries, int32_t* rounds, int32_t* states,
                   int32_t* values, uint64_t* vote_bits) {
  /* entries of this slot in insertion order; values/vote_bits are reported for Pending entries
   * only (a Done entry has dropped them, ProxyLeader.scala:256) */
  int cnt = 0;
  int order[64];
  size_t where[64];
  for (size_t i = 0; i < s->tab_cap && cnt < 64; ++i) {
    if (s->tab[i].state == 0 || (int)(s->tab[i].key >> 32) != slot || s->tab[i].is_range) continue;
    order[cnt] = s->tab[i].seq;
    where[cnt] = i;
    ++cnt;
  }
  for (int a = 0; a < cnt; ++a)
    for (int b = a + 1; b < cnt; ++b)
      if (order[b] < order[a]) {
        int t = order[a];
        order[a] = order[b];
        order[b] = t;
        size_t w = where[a];
        where[a] = where[b];
        where[b] = w;
      }
  for (int a = 0; a < cnt; ++a) {
    const tally_t* t = &s->tab[where[a]];
    if (rounds) rounds[a] = (int)(uint32_t)t->key;
    if (states) states[a] = t->state == 2 ? 1 : 0;
    if (values) values[a] = t->state == 2 ? -1 : t->value;
    if (vote_bits)
      for (int w = 0; w < 4; ++w) vote_bits[(size_t)a * 4 + w] = t->state == 2 ? 0 : t->v[w];
  }
  if (num_entries) *num_entries = cnt;
  (void)count_live;
  return FPO_OK;
}
