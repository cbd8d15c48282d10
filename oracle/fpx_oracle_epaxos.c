/*
 * fpx_oracle_epaxos.c -- TEST INFRASTRUCTURE ONLY (see fpx_oracle.h).
 *
 * CPU restatement of the EPaxos pre-accept fast path, message-at-a-time:
 *   util/TopOne.scala:6-24                                   per-leader "largest id + 1" vector
 *   statemachine/KeyValueStore.scala:225-302                 top-one conflict index of the key-value store
 *   epaxos/Replica.scala:569-600   computeSequenceNumberAndDependencies (topKDependencies == 1)
 *   epaxos/Replica.scala:633-729   transitionToPreAcceptPhase (the leader's own pre-accept)
 *   epaxos/Replica.scala:1159-1289 handlePreAccept (fresh instance: cmdLog.get == None)
 *   epaxos/Replica.scala:1291-1419 handlePreAcceptOk (fast path test) ; :796-813 preAcceptingSlowPath
 *   Util.scala:7-21                histogram / popularItems
 *   epaxos/Config.scala:7-9        n = 2f+1, fastQuorumSize = n-1, slowQuorumSize = f+1
 *
 * Scope of the restatement (and of the GPU kernels it checks): one tick of FRESH instances with
 * single-key GetRequest / SetRequest commands, thrifty fast quorums (the leader sends PreAccept to
 * exactly fastQuorumSize - 1 = n - 2 other replicas, Replica.scala:705-706), sequence numbers are the
 * constant 0 the reference uses (:575-578, 598), dependencies are per-leader watermark vectors
 * (InstancePrefixSet.fromTopOne, epaxos/InstancePrefixSet.scala:19-29).  Commit messages of the
 * tick reach every replica after the tick (commit -> updateConflictIndex, Replica.scala:815-828).
 * The reference has no known-answer test for this path (T/epaxos/EPaxos.scala is a randomized
 * invariant check): parity unpinned; anchored on the citations and tests/test_epaxos.py traces.
 * TopOne itself is pinned by T/util/TopOneTest.scala (tests/test_epaxos.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int n, num_keys;
  int* gets; /* [n][num_keys][n]  replica r's conflict index: gets(key) TopOne */
  int* sets; /* [n][num_keys][n] */
} fpo_epx;

fpo_epx* fpo_epx_new(int n, int num_keys) {
  if (n < 3 || n > 8 || !(n & 1) || num_keys < 1) return NULL;
  fpo_epx* e = (fpo_epx*)calloc(1, sizeof(fpo_epx));
  e->n = n;
  e->num_keys = num_keys;
  /* TopOne.scala:10 topOnes = Buffer.fill(numLeaders)(0) */
  e->gets = (int*)calloc((size_t)n * num_keys * n, sizeof(int));
  e->sets = (int*)calloc((size_t)n * num_keys * n, sizeof(int));
  return e;
}

void fpo_epx_free(fpo_epx* e) {
  if (!e) return;
  free(e->gets);
  free(e->sets);
  free(e);
}

/* TopOne.put  util/TopOne.scala:12-15 */
void fpo_top_one_put(int* top_ones, int leader_index, int id) {
  if (id + 1 > top_ones[leader_index]) top_ones[leader_index] = id + 1;
}

/* TopOne.mergeEquals  util/TopOne.scala:19-23 */
void fpo_top_one_merge(int* top_ones, const int* other, int num_leaders) {
  for (int i = 0; i < num_leaders; ++i)
    if (other[i] > top_ones[i]) top_ones[i] = other[i];
}

static int* idx(int* base, const fpo_epx* e, int r, int key) {
  return base + ((size_t)r * e->num_keys + key) * e->n;
}

/* KeyValueStore.scala:259-302 getTopOneConflicts for a single-key command (snapshots are all 0) */
static void get_top_one_conflicts(fpo_epx* e, int r, int key, int is_set, int* merged) {
  memset(merged, 0, sizeof(int) * (size_t)e->n);
  fpo_top_one_merge(merged, idx(e->sets, e, r, key), e->n);             /* :269-272 / :285-288 */
  if (is_set) fpo_top_one_merge(merged, idx(e->gets, e, r, key), e->n); /* :289-292 */
}

/* KeyValueStore.scala:232-253 put */
static void conflict_index_put(fpo_epx* e, int r, int key, int is_set, int leader, int number) {
  fpo_top_one_put(idx(is_set ? e->sets : e->gets, e, r, key), leader, number);
}

/* the two conflict-index operations by themselves, so that the reference's known-answer test
 * (statemachine/TopKConflictIndexTest.scala:281-329, k = 1) can be run against them */
void fpo_epx_index_put(fpo_epx* e, int replica, int key, int is_set, int leader, int number) {
  conflict_index_put(e, replica, key, is_set, leader, number);
}
void fpo_epx_index_conflicts(fpo_epx* e, int replica, int key, int is_set, int32_t* out) {
  get_top_one_conflicts(e, replica, key, is_set, out);
}

static int popcount8(unsigned x) { return __builtin_popcount(x & 0xffu); }

/*
 * One tick.  rank[r * m + i] = position of message i in replica r's processing order.
 * resp_mask[i]: the n-2 other replicas the leader sends PreAccept to.
 * Outputs: fast[i] (1 = fast path commit, 0 = slow path -> Accept phase), deps[i * n + l] = the
 * committed dependencies (fast) or the union the Accept phase proposes (slow), leader_deps[i * n + l].
 * Returns 0, or 1 (EINVAL) on malformed input.
 */
int fpo_epx_preaccept(fpo_epx* e, int32_t m, const int32_t* leader, const int32_t* number, const int32_t* key,
                      const uint8_t* is_set, const uint8_t* resp_mask, const uint8_t* seen_mask,
                      const int32_t* rank, uint8_t* fast, int32_t* deps, int32_t* leader_deps) {
  const int n = e->n;
  if (m < 0) return 1;
  for (int i = 0; i < m; ++i) {
    if (leader[i] < 0 || leader[i] >= n || number[i] < 0 || key[i] < 0 || key[i] >= e->num_keys) return 1;
    if ((resp_mask[i] >> leader[i]) & 1u) return 1;                     /* "other" replicas only */
    if (resp_mask[i] >> n) return 1;
    if (popcount8(resp_mask[i]) != n - 2) return 1;                     /* fastQuorumSize - 1, :705 */
    /* the replicas the PreAccept is sent to: thriftyOtherReplicas(fastQuorumSize - 1), :556-562, 705 --
     * exactly the n-2 with a thrifty system, every other replica with the default NotThrifty (:83) */
    const unsigned seen = seen_mask ? seen_mask[i] : resp_mask[i];
    if ((resp_mask[i] & ~seen) || ((seen >> leader[i]) & 1u) || (seen >> n)) return 1;
  }
  /* local conflicts seen by replica r for message i, in r's own processing order */
  int* conf = (int*)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1) * n * n);
  int* order = (int*)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
  /* every replica's rank must be a permutation of 0..m-1 (a delivery order); checked for all replicas before
   * anything is applied */
  for (int r = 0; r < n; ++r) {
    for (int p = 0; p < m; ++p) order[p] = -1;
    for (int i = 0; i < m; ++i) {
      int p = rank[(size_t)r * m + i];
      if (p < 0 || p >= m || order[p] != -1) {
        free(conf), free(order);
        return 1;
      }
      order[p] = i;
    }
  }
  for (int r = 0; r < n; ++r) {
    for (int i = 0; i < m; ++i) order[rank[(size_t)r * m + i]] = i;
    for (int p = 0; p < m; ++p) {
      const int i = order[p];
      const int participates = r == leader[i] || (((seen_mask ? seen_mask[i] : resp_mask[i]) >> r) & 1u);
      if (!participates) continue;
      /* Replica.scala:580-583: getTopOneConflicts then subtractOne(instance); the instance is fresh,
       * so it is not in the index yet and subtractOne changes nothing.  Then updateConflictIndex
       * (:696 for the leader, :1274 for the others). */
      get_top_one_conflicts(e, r, key[i], is_set[i], conf + ((size_t)i * n + r) * n);
      conflict_index_put(e, r, key[i], is_set[i], leader[i], number[i]);
    }
  }
  for (int i = 0; i < m; ++i) {
    const int L = leader[i];
    const int* D = conf + ((size_t)i * n + L) * n; /* the leader's own (seq = 0, deps), :641-642 */
    int first[8], uni[8];
    int have_first = 0, all_equal = 1;
    memcpy(uni, D, sizeof(int) * (size_t)n);
    for (int r = 0; r < n; ++r) {
      if (!((resp_mask[i] >> r) & 1u)) continue;
      /* handlePreAccept :1252-1257: dependencies = local conflicts ++ preAccept.dependencies */
      int resp[8];
      for (int l = 0; l < n; ++l) {
        const int c = conf[((size_t)i * n + r) * n + l];
        resp[l] = c > D[l] ? c : D[l];
        if (resp[l] > uni[l]) uni[l] = resp[l]; /* preAcceptingSlowPath :804-807 union of all responses */
      }
      if (!have_first) {
        memcpy(first, resp, sizeof(int) * (size_t)n);
        have_first = 1;
      } else if (memcmp(first, resp, sizeof(int) * (size_t)n) != 0) {
        all_equal = 0;
      }
    }
    /* handlePreAcceptOk :1376-1410: with the leader's own response plus the first n-2 others to arrive
     * (resp_mask) the fast quorum (n-1) is reached and the leader decides at once;
     * popularItems(others' (seq, deps), n-2) is non-empty iff all n-2 agree.  A later answer of a further
     * replica in seen_mask finds the instance committed / accepting and is ignored (:1308-1334). */
    const int is_fast = all_equal;
    if (fast) fast[i] = (uint8_t)is_fast;
    for (int l = 0; l < n; ++l) {
      if (deps) deps[(size_t)i * n + l] = is_fast ? first[l] : uni[l];
      if (leader_deps) leader_deps[(size_t)i * n + l] = D[l];
    }
  }
  /* commit (fast path) / Accept+commit (slow path) reach every replica after the tick:
   * commit -> updateConflictIndex(instance, command)  Replica.scala:815-828 */
  for (int r = 0; r < n; ++r)
    for (int i = 0; i < m; ++i) conflict_index_put(e, r, key[i], is_set[i], leader[i], number[i]);
  free(conf);
  free(order);
  return 0;
}

/* replica r's conflict index entries for `key`: gets[n], sets[n] */
void fpo_epx_read_index(fpo_epx* e, int r, int key, int32_t* gets, int32_t* sets) {
  memcpy(gets, idx(e->gets, e, r, key), sizeof(int) * (size_t)e->n);
  memcpy(sets, idx(e->sets, e, r, key), sizeof(int) * (size_t)e->n);
}

/* Util.popularItems (Util.scala:19-21) over int vectors of length `width`: writes the indices of one
 * representative of every item that appears >= n times; returns how many */
int fpo_popular_items(const int32_t* xs, int count, int width, int n, int32_t* out) {
  int k = 0;
  for (int a = 0; a < count; ++a) {
    int seen_before = 0, c = 0;
    for (int b = 0; b < count; ++b) {
      if (memcmp(xs + (size_t)a * width, xs + (size_t)b * width, sizeof(int32_t) * (size_t)width) == 0) {
        if (b < a) seen_before = 1;
        ++c;
      }
    }
    if (!seen_before && c >= n) out[k++] = a;
  }
  return k;
}
