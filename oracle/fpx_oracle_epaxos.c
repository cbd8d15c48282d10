/*
 * fpx_oracle_epaxos.c -- TEST INFRASTRUCTURE ONLY (see fpx_oracle.h).
 *
 * CPU restatement of the EPaxos pre-accept fast path, message-at-a-time:
 *   util/TopOne.scala:6-24                                   per-leader "largest id + 1" vector
 *   statemachine/KeyValueStore.scala:225-302                 top-one conflict index of the key-value store
 *   epaxos/Replica.scala:569-600   computeSequenceNumberAndDependencies (topKDependencies == 1)
 *   epaxos/Replica.scala:633-729   transitionToPreAcceptPhase (the leader's own pre-accept)
 *   epaxos/Replica.scala:1159-1289 handlePreAccept (the tick-at-once form: fresh instances, cmdLog.get == None;
 *                                  fpo_epx_handle_preaccept: every branch)
 *   epaxos/Replica.scala:1291-1419 handlePreAcceptOk (fast path test) ; :796-813 preAcceptingSlowPath
 *   Util.scala:7-21                histogram / popularItems
 *   epaxos/Config.scala:7-9        n = 2f+1, fastQuorumSize = n-1, slowQuorumSize = f+1
 *
 * Scope of the restatement (and of the GPU kernels it checks): one tick of FRESH instances with
 * single-key GetRequest / SetRequest commands, thrifty fast quorums (the leader sends PreAccept to
 * exactly fastQuorumSize - 1 = n - 2 other replicas, Replica.scala:705-706), sequence numbers are the
 * constant 0 the reference uses (:575-578, 598), dependencies are per-leader watermark vectors
 * (InstancePrefixSet.fromTopOne, epaxos/InstancePrefixSet.scala:19-29) from which the instance itself is
 * removed (dependencies.subtractOne(instance), Replica.scala:582; compact/IntPrefixSet.scala:388-398).
 * subtractOne only ever touches the column of the instance's OWN leader, so that column is carried as a
 * real IntPrefixSet (watermark + explicit values, restated below with its add-all / equality / compaction
 * rules) and the other columns as bare watermarks (their `values` stay empty through every operation on
 * this path: fromWatermarks -> addAll of value-less sets is a max of watermarks, IntPrefixSet.scala:296-301).
 * Commit messages of the tick reach every replica after the tick (commit -> updateConflictIndex,
 * Replica.scala:815-828).
 * The reference has no known-answer test for this path (T/epaxos/EPaxos.scala is a randomized
 * invariant check): parity unpinned; anchored on the citations and tests/test_epaxos.py traces.
 * TopOne itself is pinned by T/util/TopOneTest.scala (tests/test_epaxos.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* cmdLog entry kinds (epaxos/Replica.scala:303-330) */
enum { CL_NONE = 0, CL_NO_COMMAND = 1, CL_PRE_ACCEPTED = 2, CL_ACCEPTED = 3, CL_COMMITTED = 4 };

typedef struct {
  int n, num_keys;
  int* gets; /* [n][num_keys][n]  replica r's conflict index: gets(key) TopOne */
  int* sets; /* [n][num_keys][n] */
  /* cmdLog: mutable.Map[Instance, CmdLogEntry] per replica (Replica.scala:440), for the instances
   * (leader, number < num_instances); ballots are (ordering, replicaIndex) compared lexicographically
   * (BallotHelpers.scala:11-21), kept as ordering * 8 + replicaIndex; nullBallot (-1, -1) (Replica.scala:256) = -1 */
  int num_instances;
  unsigned char* cl_status; /* [n][n * num_instances] */
  int* cl_ballot;
  int* cl_vote;
  int* cl_triple;            /* the CommandTriple, by the caller's id */
  int* cl_deps;              /* [cells][n] the triple's dependencies as watermarks (column 0 = -1: the entry was written
                              * by an Accept, which names its triple by id only) */
  int* cl_dend;              /* [cells] end of the explicit values number + 1 .. end - 1 of the own-leader column, 0 = none */
  int* largest_ballot;       /* [n]  Replica.scala:458  var largestBallot = Ballot(0, index) */
} fpo_epx;

static int enc_ballot(int ordering, int replica) { return (ordering < 0 || replica < 0) ? -1 : ordering * 8 + replica; }

fpo_epx* fpo_epx_new2(int n, int num_keys, int num_instances) {
  if (n < 3 || n > 8 || !(n & 1) || num_keys < 1 || num_instances < 0) return NULL;
  fpo_epx* e = (fpo_epx*)calloc(1, sizeof(fpo_epx));
  e->n = n;
  e->num_keys = num_keys;
  /* TopOne.scala:10 topOnes = Buffer.fill(numLeaders)(0) */
  e->gets = (int*)calloc((size_t)n * num_keys * n, sizeof(int));
  e->sets = (int*)calloc((size_t)n * num_keys * n, sizeof(int));
  e->num_instances = num_instances;
  if (num_instances > 0) {
    const size_t cells = (size_t)n * n * num_instances;
    e->cl_status = (unsigned char*)calloc(cells, 1);
    e->cl_ballot = (int*)malloc(sizeof(int) * cells);
    e->cl_vote = (int*)malloc(sizeof(int) * cells);
    e->cl_triple = (int*)malloc(sizeof(int) * cells);
    e->cl_deps = (int*)calloc(cells * (size_t)n, sizeof(int));
    e->cl_dend = (int*)calloc(cells, sizeof(int));
    for (size_t i = 0; i < cells; ++i) e->cl_ballot[i] = e->cl_vote[i] = e->cl_triple[i] = -1;
  }
  e->largest_ballot = (int*)malloc(sizeof(int) * (size_t)n);
  for (int r = 0; r < n; ++r) e->largest_ballot[r] = enc_ballot(0, r);
  return e;
}

fpo_epx* fpo_epx_new(int n, int num_keys) { return fpo_epx_new2(n, num_keys, 0); }

void fpo_epx_free(fpo_epx* e) {
  if (!e) return;
  free(e->gets);
  free(e->sets);
  free(e->cl_status);
  free(e->cl_ballot);
  free(e->cl_vote);
  free(e->cl_triple);
  free(e->cl_deps);
  free(e->cl_dend);
  free(e->largest_ballot);
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

/* ---- compact.IntPrefixSet, the part of it this path reaches  (compact/IntPrefixSet.scala) ------------
 * watermark: every x < watermark is in the set; values: the other members, sorted ascending, all
 * >= watermark (kept compacted exactly like the Scala: the constructor and every mutator call compact()). */
typedef struct {
  int watermark;
  int nvalues;
  int* values; /* NULL when nvalues == 0 */
} ips_t;

static void ips_free(ips_t* s) {
  free(s->values);
  s->values = NULL;
  s->nvalues = 0;
}

/* IntPrefixSet.compact (IntPrefixSet.scala, private): while values contains watermark, pop it and bump */
static void ips_compact(ips_t* s) {
  int k = 0;
  /* drop members below the watermark, then absorb the run that starts at the watermark */
  while (k < s->nvalues && s->values[k] < s->watermark) ++k;
  while (k < s->nvalues && s->values[k] == s->watermark) ++k, ++s->watermark;
  if (k > 0) {
    memmove(s->values, s->values + k, sizeof(int) * (size_t)(s->nvalues - k));
    s->nvalues -= k;
  }
  if (s->nvalues == 0) ips_free(s);
}

/* IntPrefixSet.fromWatermark  :13-14 */
static ips_t ips_from_watermark(int w) {
  ips_t s = {w, 0, NULL};
  return s;
}

static ips_t ips_clone(const ips_t* a) {
  ips_t s = {a->watermark, a->nvalues, NULL};
  if (a->nvalues) {
    s.values = (int*)malloc(sizeof(int) * (size_t)a->nvalues);
    memcpy(s.values, a->values, sizeof(int) * (size_t)a->nvalues);
  }
  return s;
}

/* IntPrefixSet.subtractOne  :388-398
 *   if (x >= watermark) values -= x  else { for (i <- x + 1 until watermark) values += i; watermark = x } */
static void ips_subtract_one(ips_t* s, int x) {
  if (x >= s->watermark) {
    for (int k = 0; k < s->nvalues; ++k) {
      if (s->values[k] == x) {
        memmove(s->values + k, s->values + k + 1, sizeof(int) * (size_t)(s->nvalues - k - 1));
        if (--s->nvalues == 0) ips_free(s);
        break;
      }
    }
  } else {
    const int add = s->watermark - (x + 1);
    int* v = (int*)malloc(sizeof(int) * (size_t)(add + s->nvalues + 1));
    for (int i = 0; i < add; ++i) v[i] = x + 1 + i; /* all below the old watermark, hence below every old value */
    if (s->nvalues) memcpy(v + add, s->values, sizeof(int) * (size_t)s->nvalues);
    free(s->values);
    s->values = v;
    s->nvalues += add;
    s->watermark = x;
    if (s->nvalues == 0) ips_free(s);
  }
}

/* IntPrefixSet.addAll  :296-330: the union, compacted (the four emptiness cases of the Scala all compute
 * watermark = max, values = (values ++ other.values).filter(_ >= watermark), compact()) */
static void ips_add_all(ips_t* s, const ips_t* o) {
  const int w = s->watermark > o->watermark ? s->watermark : o->watermark;
  int* v = (int*)malloc(sizeof(int) * (size_t)(s->nvalues + o->nvalues + 1));
  int a = 0, b = 0, k = 0;
  while (a < s->nvalues || b < o->nvalues) { /* sorted merge without duplicates */
    int x;
    if (b >= o->nvalues || (a < s->nvalues && s->values[a] <= o->values[b])) {
      x = s->values[a++];
      if (b < o->nvalues && o->values[b] == x) ++b;
    } else {
      x = o->values[b++];
    }
    if (x >= w) v[k++] = x;
  }
  free(s->values);
  s->values = v;
  s->nvalues = k;
  s->watermark = w;
  if (k == 0) ips_free(s);
  ips_compact(s);
}

/* IntPrefixSet.equals  :214-221: (watermark, values) tuples */
static int ips_equals(const ips_t* a, const ips_t* b) {
  if (a->watermark != b->watermark || a->nvalues != b->nvalues) return 0;
  return a->nvalues == 0 || memcmp(a->values, b->values, sizeof(int) * (size_t)a->nvalues) == 0;
}

/* for the tests: the operations above on their own (T/compact/IntPrefixSetTest.scala pins them) */
int fpo_ips_subtract_one(int watermark, const int* values, int nvalues, int x, int* out_values, int* out_watermark) {
  ips_t s = {watermark, nvalues, NULL};
  if (nvalues) {
    s.values = (int*)malloc(sizeof(int) * (size_t)nvalues);
    memcpy(s.values, values, sizeof(int) * (size_t)nvalues);
  }
  ips_compact(&s);
  ips_subtract_one(&s, x);
  const int n = s.nvalues;
  if (n) memcpy(out_values, s.values, sizeof(int) * (size_t)n);
  *out_watermark = s.watermark;
  ips_free(&s);
  return n;
}
int fpo_ips_add_all(int wa, const int* va, int na, int wb, const int* vb, int nb, int* out_values, int* out_watermark) {
  ips_t a = {wa, na, NULL}, b = {wb, nb, NULL};
  if (na) a.values = (int*)malloc(sizeof(int) * (size_t)na), memcpy(a.values, va, sizeof(int) * (size_t)na);
  if (nb) b.values = (int*)malloc(sizeof(int) * (size_t)nb), memcpy(b.values, vb, sizeof(int) * (size_t)nb);
  ips_compact(&a);
  ips_compact(&b);
  ips_add_all(&a, &b);
  const int n = a.nvalues;
  if (n) memcpy(out_values, a.values, sizeof(int) * (size_t)n);
  *out_watermark = a.watermark;
  ips_free(&a);
  ips_free(&b);
  return n;
}

int fpo_epx_preaccept2(fpo_epx* e, int32_t m, const int32_t* leader, const int32_t* number, const int32_t* key,
                       const uint8_t* is_set, const uint8_t* resp_mask, const uint8_t* seen_mask,
                       const int32_t* rank, const int32_t* triple_id, uint8_t* fast, int32_t* deps,
                       int32_t* leader_deps, int32_t* own_values_end);

/* the dependencies of a command-log triple: watermarks of the n columns, the own-leader column from its
 * IntPrefixSet (its explicit values are a run that ends at cl_dend - 1; the callers check that) */
static void store_deps(fpo_epx* e, size_t cell, int own_leader, const int* watermarks, const ips_t* own) {
  for (int l = 0; l < e->n; ++l) e->cl_deps[cell * (size_t)e->n + l] = l == own_leader ? own->watermark : watermarks[l];
  e->cl_dend[cell] = own->nvalues ? own->values[own->nvalues - 1] + 1 : 0;
}

/*
 * One tick.  rank[r * m + i] = position of message i in replica r's processing order.
 * resp_mask[i]: the n-2 other replicas the leader sends PreAccept to.
 * Outputs: fast[i] (1 = fast path commit, 0 = slow path -> Accept phase); deps = the committed
 * dependencies (fast) or the union the Accept phase proposes (slow) and leader_deps = the PreAccept's, both
 * as InstancePrefixSets: deps[i * n + l] = the IntPrefixSet watermark of column l; only the column of the
 * instance's own leader can carry explicit values (subtractOne), and they always are the run
 * number[i] + 1 .. own_values_end - 1: own_values_end[i * 2 + 0] for deps, [i * 2 + 1] for leader_deps,
 * 0 = no values.  Returns 0, 1 (EINVAL) on malformed input, 99 if an own column's values are NOT such a run
 * (the encoding could not represent the reference's set: never on this path, asserted by the tests).
 */
int fpo_epx_preaccept(fpo_epx* e, int32_t m, const int32_t* leader, const int32_t* number, const int32_t* key,
                      const uint8_t* is_set, const uint8_t* resp_mask, const uint8_t* seen_mask,
                      const int32_t* rank, uint8_t* fast, int32_t* deps, int32_t* leader_deps,
                      int32_t* own_values_end) {
  return fpo_epx_preaccept2(e, m, leader, number, key, is_set, resp_mask, seen_mask, rank, NULL, fast, deps, leader_deps,
                            own_values_end);
}

/* the same with the command log kept (num_instances > 0): triple_id[i] (NULL = -1) names the CommandTriple of
 * message i; every participating replica must not have seen the instance (cmdLog.get == None, the only branch of
 * handlePreAccept this tick-at-once form covers -- anything else is EINVAL 1 and nothing is applied) and records
 * PreAcceptedEntry(ballot = voteBallot = Ballot(0, leader), triple) (:688-696, :1259-1271); a fast-path commit
 * makes it CommittedEntry at every replica (commit :815-823 + Commit to the others, handleCommit), a slow-path
 * decision leaves the entries for the Accept phase (fpo_epx_accept) */
int fpo_epx_preaccept2(fpo_epx* e, int32_t m, const int32_t* leader, const int32_t* number, const int32_t* key,
                       const uint8_t* is_set, const uint8_t* resp_mask, const uint8_t* seen_mask,
                       const int32_t* rank, const int32_t* triple_id, uint8_t* fast, int32_t* deps,
                       int32_t* leader_deps, int32_t* own_values_end) {
  const int n = e->n;
  if (m < 0) return 1;
  if (e->num_instances > 0) {
    for (int i = 0; i < m; ++i) {
      if (leader[i] < 0 || leader[i] >= n || number[i] < 0 || number[i] >= e->num_instances) return 1;
      const unsigned part = (seen_mask ? seen_mask[i] : resp_mask[i]) | (1u << leader[i]);
      for (int r = 0; r < n; ++r)
        if (((part >> r) & 1u) && e->cl_status[((size_t)r * n + leader[i]) * e->num_instances + number[i]] != CL_NONE)
          return 1;
    }
  }
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
  /* local conflicts seen by replica r for message i, in r's own processing order: the watermarks of the
   * n columns, and the own-leader column again as the IntPrefixSet that subtractOne leaves */
  const size_t mm = (size_t)(m > 0 ? m : 1);
  int* conf = (int*)malloc(sizeof(int) * mm * n * n);
  ips_t* own = (ips_t*)calloc(mm * n, sizeof(ips_t));
  int* order = (int*)malloc(sizeof(int) * mm);
  /* every replica's rank must be a permutation of 0..m-1 (a delivery order); checked for all replicas before
   * anything is applied */
  for (int r = 0; r < n; ++r) {
    for (int p = 0; p < m; ++p) order[p] = -1;
    for (int i = 0; i < m; ++i) {
      int p = rank[(size_t)r * m + i];
      if (p < 0 || p >= m || order[p] != -1) {
        free(conf), free(own), free(order);
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
      /* Replica.scala:580-583: InstancePrefixSet.fromTopOne(getTopOneConflicts), then
       * dependencies.subtractOne(instance) -- IntPrefixSet.subtractOne on the column of the instance's own
       * leader (epaxos/InstancePrefixSet.scala subtractOne -> intPrefixSets(replicaIndex)).  The instance is
       * fresh, so it is not in the index itself; but a HIGHER-numbered instance of the same leader on the same
       * key may be (processed earlier here): then the watermark drops to the instance number and the ids
       * above it become explicit values.  Then updateConflictIndex (:696 for the leader, :1274 for the others). */
      int* c = conf + ((size_t)i * n + r) * n;
      get_top_one_conflicts(e, r, key[i], is_set[i], c);
      ips_t* o = &own[(size_t)i * n + r];
      *o = ips_from_watermark(c[leader[i]]);
      ips_subtract_one(o, number[i]);
      conflict_index_put(e, r, key[i], is_set[i], leader[i], number[i]);
    }
  }
  int rc = 0;
  for (int i = 0; i < m; ++i) {
    const int L = leader[i];
    const int* D = conf + ((size_t)i * n + L) * n; /* the leader's own (seq = 0, deps), :641-642 */
    const ips_t* Down = &own[(size_t)i * n + L];
    int first[8], uni[8];
    ips_t first_own = ips_from_watermark(0), uni_own = ips_clone(Down);
    int have_first = 0, all_equal = 1;
    memcpy(uni, D, sizeof(int) * (size_t)n);
    for (int r = 0; r < n; ++r) {
      if (!((resp_mask[i] >> r) & 1u)) continue;
      /* handlePreAccept :1252-1257: dependencies = local conflicts, addAll(preAccept.dependencies) */
      int resp[8];
      ips_t resp_own = ips_clone(&own[(size_t)i * n + r]);
      ips_add_all(&resp_own, Down);
      ips_add_all(&uni_own, &resp_own); /* preAcceptingSlowPath :804-807 union of all responses */
      for (int l = 0; l < n; ++l) {
        const int c = conf[((size_t)i * n + r) * n + l];
        resp[l] = c > D[l] ? c : D[l];
        if (resp[l] > uni[l]) uni[l] = resp[l];
      }
      if (!have_first) {
        memcpy(first, resp, sizeof(int) * (size_t)n);
        first_own = ips_clone(&resp_own);
        have_first = 1;
      } else {
        /* Util.popularItems compares (sequenceNumber, InstancePrefixSet) values: per column
         * (watermark, values) tuples (InstancePrefixSet.equals -> IntPrefixSet.equals) */
        for (int l = 0; l < n; ++l) {
          if (l == L) {
            if (!ips_equals(&first_own, &resp_own)) all_equal = 0;
          } else if (first[l] != resp[l]) {
            all_equal = 0;
          }
        }
      }
      ips_free(&resp_own);
    }
    /* handlePreAcceptOk :1376-1410: with the leader's own response plus the first n-2 others to arrive
     * (resp_mask) the fast quorum (n-1) is reached and the leader decides at once;
     * popularItems(others' (seq, deps), n-2) is non-empty iff all n-2 agree.  A later answer of a further
     * replica in seen_mask finds the instance committed / accepting and is ignored (:1308-1334). */
    const int is_fast = all_equal;
    if (fast) fast[i] = (uint8_t)is_fast;
    const ips_t* out_own = is_fast ? &first_own : &uni_own;
    if (e->num_instances > 0) {
      const unsigned part = (seen_mask ? seen_mask[i] : resp_mask[i]) | (1u << L);
      for (int r = 0; r < n; ++r) {
        const size_t c = ((size_t)r * n + L) * e->num_instances + number[i];
        if (is_fast) {
          /* the committed triple: (seq, deps) of the fast-path agreement (:1399-1408 -> commit) */
          e->cl_status[c] = CL_COMMITTED, e->cl_ballot[c] = e->cl_vote[c] = -1;
          e->cl_triple[c] = triple_id ? triple_id[i] : -1;
          int wm[8];
          memcpy(wm, first, sizeof(int) * (size_t)n);
          store_deps(e, c, L, wm, out_own);
        } else if ((part >> r) & 1u) {
          /* every replica that processed the PreAccept keeps ITS answer as the triple (:1259-1271); the leader
           * the dependencies it proposed (:688-696) */
          e->cl_status[c] = CL_PRE_ACCEPTED, e->cl_ballot[c] = e->cl_vote[c] = enc_ballot(0, L);
          e->cl_triple[c] = triple_id ? triple_id[i] : -1;
          int wm[8];
          ips_t mine = ips_clone(&own[(size_t)i * n + r]);
          if (r != L) ips_add_all(&mine, Down);
          for (int l = 0; l < n; ++l) {
            const int cl = conf[((size_t)i * n + r) * n + l];
            wm[l] = (r != L && D[l] > cl) ? D[l] : cl;
          }
          store_deps(e, c, L, wm, &mine);
          ips_free(&mine);
        }
      }
    }
    for (int l = 0; l < n; ++l) {
      if (deps) deps[(size_t)i * n + l] = l == L ? out_own->watermark : (is_fast ? first[l] : uni[l]);
      if (leader_deps) leader_deps[(size_t)i * n + l] = l == L ? Down->watermark : D[l];
    }
    /* the own column's explicit values, as the end of the run number + 1 .. end - 1 */
    const ips_t* sets2[2] = {out_own, Down};
    for (int k = 0; k < 2; ++k) {
      const ips_t* q = sets2[k];
      int end = 0;
      if (q->nvalues) {
        end = q->values[q->nvalues - 1] + 1;
        if (q->watermark != number[i] || q->values[0] != number[i] + 1 || end - q->values[0] != q->nvalues) rc = 99;
      }
      if (own_values_end) own_values_end[(size_t)i * 2 + k] = end;
    }
    ips_free(&first_own);
    ips_free(&uni_own);
  }
  /* commit (fast path) / Accept+commit (slow path) reach every replica after the tick:
   * commit -> updateConflictIndex(instance, command)  Replica.scala:815-828 */
  for (int r = 0; r < n; ++r)
    for (int i = 0; i < m; ++i) conflict_index_put(e, r, key[i], is_set[i], leader[i], number[i]);
  for (size_t k = 0; k < mm * n; ++k) ips_free(&own[k]);
  free(conf);
  free(own);
  free(order);
  return rc;
}

/* ---- the per-instance Paxos of EPaxos: Prepare (phase 1) and Accept (phase 2) on the command log ---------------
 * Messages are delivered in array order, each to the replicas of its target_mask (bit r); the instances of one
 * call must be pairwise distinct (EINVAL 1 otherwise: the GPU evaluates a batch at once).  Per message the replies:
 * ok_bits / nack_bits / commit_bits (the replica answered with the Commit it already has), nack_ballot = the largest
 * `largestBallot` a Nack carried (encoded ordering * 8 + replicaIndex; -1 = none). */
/* an Accept names its triple by the caller's id alone: the stored dependencies are marked unknown */
static void deps_by_id(fpo_epx* e, size_t cell) {
  for (int l = 0; l < e->n; ++l) e->cl_deps[cell * (size_t)e->n + l] = -1;
  e->cl_dend[cell] = 0;
}

static int instances_ok(const fpo_epx* e, int m, const int32_t* leader, const int32_t* number, const int32_t* b_ord,
                        const int32_t* b_rep, const uint8_t* target) {
  if (e->num_instances <= 0 || m < 0) return 0;
  unsigned char* seen = (unsigned char*)calloc((size_t)e->n * e->num_instances, 1);
  int ok = 1;
  for (int i = 0; i < m && ok; ++i) {
    if (leader[i] < 0 || leader[i] >= e->n || number[i] < 0 || number[i] >= e->num_instances || b_ord[i] < 0 ||
        b_ord[i] >= (1 << 27) || b_rep[i] < 0 || b_rep[i] >= e->n || (target[i] >> e->n))
      ok = 0;
    else if (seen[(size_t)leader[i] * e->num_instances + number[i]]++)
      ok = 0;
  }
  free(seen);
  return ok;
}

/* Replica.handlePrepare  epaxos/Replica.scala:1632-1757, at every replica of target_mask.  Per (message, replica):
 * reply_status = the PrepareOk's CommandStatus as the entry kind (0 NotSeen, 2 PreAccepted, 3 Accepted; -1 when the
 * replica did not answer PrepareOk), reply_vote = its voteBallot (encoded), reply_triple = its triple id. */
int fpo_epx_prepare(fpo_epx* e, int32_t m, const int32_t* leader, const int32_t* number, const int32_t* b_ord,
                    const int32_t* b_rep, const uint8_t* target, uint8_t* ok_bits, uint8_t* nack_bits,
                    uint8_t* commit_bits, int32_t* nack_ballot, int32_t* reply_status, int32_t* reply_vote,
                    int32_t* reply_triple) {
  const int n = e->n;
  if (!instances_ok(e, m, leader, number, b_ord, b_rep, target)) return 1;
  for (int i = 0; i < m; ++i) {
    const int ballot = enc_ballot(b_ord[i], b_rep[i]);
    unsigned ok = 0, nack = 0, com = 0;
    int nb = -1;
    for (int r = 0; r < n; ++r) {
      int rs = -1, rv = -1, rt = -1;
      if ((target[i] >> r) & 1u) {
        /* :1637 largestBallot = max(largestBallot, prepare.ballot) -- before anything else */
        if (ballot > e->largest_ballot[r]) e->largest_ballot[r] = ballot;
        const size_t c = ((size_t)r * n + leader[i]) * e->num_instances + number[i];
        const int st = e->cl_status[c];
        if (st == CL_COMMITTED) {
          com |= 1u << r; /* :1744-1755 */
        } else if (st != CL_NONE && ballot < e->cl_ballot[c]) {
          nack |= 1u << r; /* :1681-1684, :1706-1709, :1726-1729  Nack(instance, largestBallot) */
          if (e->largest_ballot[r] > nb) nb = e->largest_ballot[r];
        } else {
          ok |= 1u << r;
          if (st == CL_NONE || st == CL_NO_COMMAND) { /* :1654-1669, :1686-1701 */
            rs = 0, rv = -1, rt = -1;
            e->cl_status[c] = CL_NO_COMMAND, e->cl_vote[c] = -1, e->cl_triple[c] = -1;
          } else { /* :1711-1724, :1731-1743: the entry keeps its vote, only `ballot` moves */
            rs = st, rv = e->cl_vote[c], rt = e->cl_triple[c];
          }
          e->cl_ballot[c] = ballot;
        }
      }
      if (reply_status) reply_status[(size_t)i * n + r] = rs;
      if (reply_vote) reply_vote[(size_t)i * n + r] = rv;
      if (reply_triple) reply_triple[(size_t)i * n + r] = rt;
    }
    if (ok_bits) ok_bits[i] = (uint8_t)ok;
    if (nack_bits) nack_bits[i] = (uint8_t)nack;
    if (commit_bits) commit_bits[i] = (uint8_t)com;
    if (nack_ballot) nack_ballot[i] = nb;
  }
  return 0;
}

/* Replica.handlePrepareOk  epaxos/Replica.scala:1759-1884 -- the recovering replica's decision once slowQuorumSize
 * PrepareOks are in.  Instance i is recovered by replica b_rep[i] in ballot (b_ord[i], b_rep[i]); resp_mask[i] = the
 * replicas whose PrepareOk it holds (fpo_epx_prepare's ok_bits), reply_* = their contents (fpo_epx_prepare's outputs).
 *   action 0  fewer than f + 1 responses: wait (:1799-1801)
 *   action 1  transitionToAcceptPhase(instance, ballot, triple of `source`)   (:1810-1824, :1846-1851)
 *   action 2  transitionToPreAcceptPhase(instance, ballot, the command of `source`'s triple, avoidFastPath) (:1856-1862)
 *   action 3  transitionToPreAcceptPhase(instance, ballot, Noop, avoidFastPath)  (:1863-1868)
 * as_intended = 0: AS THE REFERENCE EVALUATES IT.  Two of its tests can never hold:
 *   (a) :1810  prepareOks.find(_.status == Some(CommandStatus.Accepted)) -- `status` is a REQUIRED proto field
 *       (EPaxos.proto:203), so a CommandStatus is compared with an Option and the comparison is always false: an
 *       accepted value is never picked up here;
 *   (b) :1831  .filter(p => p.ballot == Ballot(0, p.instance.replicaIndex)) -- p.ballot is the ballot of the Prepare
 *       being answered (handlePrepare copies prepare.ballot), i.e. the recovery ballot, whose ordering is >= 1
 *       (transitionToPreparePhase: largestBallot.ordering + 1): the filter keeps nothing, popularItems finds nothing.
 *   What is left: among the responses with the highest voteBallot, ANY PreAccepted one restarts the pre-accept phase
 *   with its command, otherwise a Noop is proposed -- also when those responses were Accepted.
 * as_intended = 1: as the comments of :1806-1809, :1826-1829 say: an Accepted response at the highest voteBallot wins;
 *   else f identical PreAccepted triples voted in the instance's default ballot Ballot(0, leader), not from the
 *   recovering replica, win (popularItems(.., f), the triples compared as the command log holds them: id,
 *   dependencies, explicit values); else as above.
 * `responses` is a hash map: which of several eligible responses `find` returns is unspecified; they all carry the
 * same command (one voteBallot = one proposer), `source` is the lowest replica index. */
int fpo_epx_handle_prepare_oks(fpo_epx* e, int32_t m, const int32_t* leader, const int32_t* number, const int32_t* b_ord,
                               const int32_t* b_rep, const uint8_t* resp_mask, const int32_t* reply_status,
                               const int32_t* reply_vote, const int32_t* reply_triple, int32_t as_intended,
                               int32_t* action, int32_t* source, int32_t* triple) {
  const int n = e->n, f = (n - 1) / 2;
  if (e->num_instances <= 0 || m < 0) return 1;
  for (int i = 0; i < m; ++i) {
    if (leader[i] < 0 || leader[i] >= n || number[i] < 0 || number[i] >= e->num_instances || b_rep[i] < 0 ||
        b_rep[i] >= n || b_ord[i] < 0 || (resp_mask[i] >> n))
      return 1;
    for (int r = 0; r < n; ++r)
      if (((resp_mask[i] >> r) & 1u) && reply_status[(size_t)i * n + r] < 0) return 1; /* no PrepareOk from r */
  }
  for (int i = 0; i < m; ++i) {
    const int32_t* rs = reply_status + (size_t)i * n;
    const int32_t* rv = reply_vote + (size_t)i * n;
    const int32_t* rt = reply_triple + (size_t)i * n;
    const unsigned mask = resp_mask[i];
    int act = 0, src = -1, tr = -1;
    if (popcount8((uint8_t)mask) >= f + 1) { /* :1799 responses.size < slowQuorumSize -> wait */
      int maxvb = -2;                          /* :1805-1807 only the responses of the highest voteBallot count */
      for (int r = 0; r < n; ++r)
        if (((mask >> r) & 1u) && rv[r] > maxvb) maxvb = rv[r];
      if (as_intended) {
        for (int r = 0; r < n && act == 0; ++r) /* :1810-1824 some response was accepted: go with it */
          if (((mask >> r) & 1u) && rv[r] == maxvb && rs[r] == CL_ACCEPTED) act = 1, src = r, tr = rt[r];
      }
      if (act == 0) {
        /* :1830-1851 f matching PreAccepted responses in the default ballot, the recovering replica's own excluded */
        const int in_default = as_intended ? maxvb == enc_ballot(0, leader[i]) /* the vote was cast in it */
                                           : enc_ballot(b_ord[i], b_rep[i]) == enc_ballot(0, leader[i]); /* p.ballot */
        for (int r = 0; r < n && act == 0 && in_default; ++r) {
          if (!((mask >> r) & 1u) || rv[r] != maxvb || rs[r] != CL_PRE_ACCEPTED || r == b_rep[i]) continue;
          int same = 0;
          const size_t cr = ((size_t)r * n + leader[i]) * e->num_instances + number[i];
          for (int q = 0; q < n; ++q) {
            if (!((mask >> q) & 1u) || rv[q] != maxvb || rs[q] != CL_PRE_ACCEPTED || q == b_rep[i]) continue;
            const size_t cq = ((size_t)q * n + leader[i]) * e->num_instances + number[i];
            int eq = rt[q] == rt[r] && e->cl_dend[cq] == e->cl_dend[cr];
            for (int l = 0; l < n && eq; ++l) eq = e->cl_deps[cq * n + l] == e->cl_deps[cr * n + l];
            same += eq;
          }
          if (same >= f) act = 1, src = r, tr = rt[r]; /* Util.popularItems(.., config.f) */
        }
      }
      if (act == 0) { /* :1856-1868 start over: with a command if any response pre-accepted one, else with a Noop */
        act = 3;
        for (int r = 0; r < n && act == 3; ++r)
          if (((mask >> r) & 1u) && rv[r] == maxvb && rs[r] == CL_PRE_ACCEPTED) act = 2, src = r, tr = rt[r];
      }
    }
    if (action) action[i] = act;
    if (source) source[i] = src;
    if (triple) triple[i] = tr;
  }
  return 0;
}

/* The Accept phase of message i, proposed by replica b_rep[i] in ballot (b_ord[i], b_rep[i]):
 *   transitionToAcceptPhase at the proposer (:732-792): its own entry becomes AcceptedEntry(ballot, ballot, triple)
 *     -- a CommittedEntry there is logger.fatal (:740-744), an entry with a larger ballot a failed logger.checkLe
 *     (:749-757): both return 9 = FPX_EFATAL_PROTOCOL (nothing of message i is applied; the other messages are) -- and its own
 *     AcceptOk is the first response (:780-789);
 *   handleAccept at every replica of target_mask (:1421-1511), the proposer excluded;
 *   handleAcceptOk at the proposer (:1513-1565): slowQuorumSize = f + 1 responses (Config.scala:9) -> commit;
 *   commit (:815-860) with informOthers: CommittedEntry at every replica once the tick is over.
 * committed[i] = 1 if the instance got committed by this message. */
int fpo_epx_accept(fpo_epx* e, int32_t m, const int32_t* leader, const int32_t* number, const int32_t* b_ord,
                   const int32_t* b_rep, const int32_t* triple_id, const int32_t* key, const uint8_t* is_set,
                   const uint8_t* target, uint8_t* ok_bits, uint8_t* nack_bits, uint8_t* commit_bits,
                   int32_t* nack_ballot, uint8_t* committed) {
  const int n = e->n, f = (n - 1) / 2;
  if (!instances_ok(e, m, leader, number, b_ord, b_rep, target)) return 1;
  if (m > 0 && (!key || !is_set)) return 1;
  for (int i = 0; i < m; ++i) {
    if ((target[i] >> b_rep[i]) & 1u) return 1; /* thriftyOtherReplicas: never the proposer itself (:774) */
    if (key[i] < -1 || key[i] >= e->num_keys) return 1;
  }
  int status = 0;
  for (int i = 0; i < m; ++i) {
    const int ballot = enc_ballot(b_ord[i], b_rep[i]), P = b_rep[i];
    unsigned ok = 0, nack = 0, com = 0;
    int nb = -1;
    if (ok_bits) ok_bits[i] = 0;
    if (nack_bits) nack_bits[i] = 0;
    if (commit_bits) commit_bits[i] = 0;
    if (nack_ballot) nack_ballot[i] = -1;
    if (committed) committed[i] = 0;
    const size_t cp = ((size_t)P * n + leader[i]) * e->num_instances + number[i];
    if (e->cl_status[cp] == CL_COMMITTED || (e->cl_status[cp] != CL_NONE && e->cl_ballot[cp] > ballot) ||
        (e->cl_status[cp] >= CL_PRE_ACCEPTED && e->cl_vote[cp] > ballot)) {
      if (!status) status = 9; /* FPX_EFATAL_PROTOCOL */
      continue;
    }
    e->cl_status[cp] = CL_ACCEPTED, e->cl_ballot[cp] = e->cl_vote[cp] = ballot, e->cl_triple[cp] = triple_id[i];
    deps_by_id(e, cp);
    /* updateConflictIndex(instance, triple.commandOrNoop) :763 -- a Noop has no bytes to put (:602-614) */
    if (key[i] >= 0) conflict_index_put(e, P, key[i], is_set[i], leader[i], number[i]);
    ok |= 1u << P;
    for (int r = 0; r < n; ++r) {
      if (!((target[i] >> r) & 1u)) continue;
      const size_t c = ((size_t)r * n + leader[i]) * e->num_instances + number[i];
      const int st = e->cl_status[c];
      if (st == CL_COMMITTED) {
        com |= 1u << r; /* :1463-1474 */
        continue;
      }
      if (st != CL_NONE && ballot < e->cl_ballot[c]) { /* :1432-1449 Nack(instance, largestBallot) */
        nack |= 1u << r;
        if (e->largest_ballot[r] > nb) nb = e->largest_ballot[r];
        continue;
      }
      if (st == CL_ACCEPTED && ballot == e->cl_vote[c]) { /* :1451-1461 already answered: re-send the AcceptOk */
        ok |= 1u << r;
        continue;
      }
      if (ballot > e->largest_ballot[r]) e->largest_ballot[r] = ballot; /* :1487 */
      e->cl_status[c] = CL_ACCEPTED, e->cl_ballot[c] = e->cl_vote[c] = ballot, e->cl_triple[c] = triple_id[i]; /* :1493-1502 */
      deps_by_id(e, c);
      if (key[i] >= 0) conflict_index_put(e, r, key[i], is_set[i], leader[i], number[i]); /* :1503 */
      ok |= 1u << r;
    }
    if (ok_bits) ok_bits[i] = (uint8_t)ok;
    if (nack_bits) nack_bits[i] = (uint8_t)nack;
    if (commit_bits) commit_bits[i] = (uint8_t)com;
    if (nack_ballot) nack_ballot[i] = nb;
    if (popcount8(ok) >= f + 1) { /* :1557-1563 */
      if (committed) committed[i] = 1;
      for (int r = 0; r < n; ++r) {
        const size_t c = ((size_t)r * n + leader[i]) * e->num_instances + number[i];
        e->cl_status[c] = CL_COMMITTED, e->cl_ballot[c] = e->cl_vote[c] = -1, e->cl_triple[c] = triple_id[i];
        deps_by_id(e, c);
        if (key[i] >= 0) conflict_index_put(e, r, key[i], is_set[i], leader[i], number[i]); /* commit :828 */
      }
    }
  }
  return status;
}

/* Replica.handleCommit (epaxos/Replica.scala:1567-1575) -> commit(.., informOthers = false) (:815-830) at every replica of
 * target_mask, messages in array order: whatever the replica's command log held for the instance is replaced by
 * CommittedEntry(triple) (:826-827, no ballot is looked at: a Commit is final), and the conflict index learns the
 * command (:828).  deps = n watermarks per message + deps_values_end (the explicit values number + 1 .. end - 1 of the own
 * column), or NULL: the triple is known by its id alone, as after an Accept.  key -1 = Noop.  Timers and leaderStates
 * (:822, :831) are the caller's; so is the dependency graph (:859-875). */
int fpo_epx_handle_commit(fpo_epx* e, int32_t m, const int32_t* leader, const int32_t* number, const int32_t* triple_id,
                              const int32_t* key, const uint8_t* is_set, const int32_t* deps, const int32_t* deps_values_end,
                              const uint8_t* target_mask) {
  if (!e || m < 0 || e->num_instances <= 0) return 1;
  if (m == 0) return 0;
  if (!leader || !number || !triple_id || !key || !is_set || !target_mask) return 1;
  const int n = e->n;
  for (int i = 0; i < m; ++i) {
    if (leader[i] < 0 || leader[i] >= n || number[i] < 0 || number[i] >= e->num_instances || key[i] < -1 || key[i] >= e->num_keys ||
        (target_mask[i] >> n) != 0)
      return 1;
    if (deps) {
      for (int l = 0; l < n; ++l)
        if (deps[(size_t)i * n + l] < 0) return 1;
      const int end = deps_values_end ? deps_values_end[i] : 0;
      if (end != 0 && (end <= number[i] + 1 || deps[(size_t)i * n + leader[i]] > number[i])) return 1;
    }
  }
  for (int i = 0; i < m; ++i)
    for (int r = 0; r < n; ++r) {
      if (!((target_mask[i] >> r) & 1u)) continue;
      const size_t c = ((size_t)r * n + leader[i]) * e->num_instances + number[i];
      e->cl_status[c] = CL_COMMITTED, e->cl_ballot[c] = e->cl_vote[c] = -1, e->cl_triple[c] = triple_id[i];
      if (deps) {
        for (int l = 0; l < n; ++l) e->cl_deps[c * (size_t)n + l] = deps[(size_t)i * n + l];
        e->cl_dend[c] = deps_values_end ? deps_values_end[i] : 0;
      } else {
        deps_by_id(e, c);
      }
      if (key[i] >= 0) conflict_index_put(e, r, key[i], is_set[i], leader[i], number[i]);
    }
  return 0;
}

/* Replica.handlePreAccept in full (epaxos/Replica.scala:1159-1289) at every replica of target_mask, messages delivered
 * in array order; the instances of one call are pairwise distinct.  Message i = PreAccept(instance (leader, number),
 * ballot (b_ord, b_rep), commandOrNoop = single-key get / set on key[i] or Noop (key[i] == -1), sequenceNumber 0,
 * dependencies deps_in[i] = n watermarks + deps_in_end[i] (the explicit values number + 1 .. end - 1 of the instance's
 * own-leader column, 0 = none; NULL = none anywhere)).  Per (message, replica):
 *   cmdLog.get(instance)                                                      :1169
 *     None                                     -> process
 *     NoCommandEntry(b):  ballot < b           -> Nack(instance, largestBallot)       :1174-1185
 *     PreAcceptedEntry(b, vb, triple): ballot < b -> Nack ; ballot == vb -> the PreAcceptOk again, from the stored
 *                                                 triple                                :1187-1211
 *     AcceptedEntry(b, vb, _): ballot < b -> Nack ; ballot == vb -> ignored             :1213-1225
 *     CommittedEntry(triple)                   -> Commit(triple) back                   :1227-1238
 *   process: largestBallot = max(largestBallot, ballot) :1251; dependencies = computeSequenceNumberAndDependencies
 *     (conflicts of the command in this replica's index minus the instance itself; empty for a Noop, :569-600)
 *     addAll preAccept.dependencies :1257-1262; cmdLog.put(PreAcceptedEntry(ballot, ballot, triple)) :1265-1276;
 *     updateConflictIndex (a Noop leaves the index alone, :602-614) :1279; PreAcceptOk(dependencies) :1282-1293.
 * (The leaderStates / timer bookkeeping of :1243-1254 is leader-side state outside this path.)
 * Outputs per message: ok_bits (processed), resend_bits (PreAcceptOk sent again), nack_bits, commit_bits; the other
 * replicas of target_mask ignored it.  nack_ballot = the largest `largestBallot` a Nack carried.  reply_deps[i][r] /
 * reply_end[i][r] / reply_triple[i][r]: the dependencies (and triple id) of the PreAcceptOk or Commit replica r sent,
 * zeros / 0 / -1 where it sent neither.  Returns 0, 1 (EINVAL), 99 if a set is not representable (never: asserted). */
int fpo_epx_handle_preaccept(fpo_epx* e, int32_t m, const int32_t* leader, const int32_t* number, const int32_t* b_ord,
                             const int32_t* b_rep, const int32_t* key, const uint8_t* is_set, const int32_t* triple_id,
                             const int32_t* deps_in, const int32_t* deps_in_end, const uint8_t* target,
                             uint8_t* ok_bits, uint8_t* resend_bits, uint8_t* nack_bits, uint8_t* commit_bits,
                             int32_t* nack_ballot, int32_t* reply_deps, int32_t* reply_end, int32_t* reply_triple) {
  const int n = e->n;
  if (!instances_ok(e, m, leader, number, b_ord, b_rep, target)) return 1;
  for (int i = 0; i < m; ++i) {
    if (key[i] < -1 || key[i] >= e->num_keys) return 1;
    const int x = number[i], end = deps_in_end ? deps_in_end[i] : 0;
    for (int l = 0; l < n; ++l)
      if (deps_in[(size_t)i * n + l] < 0) return 1;
    /* a PreAccept never depends on its own instance (:582), and the only explicit values this path produces are
     * the run number + 1 .. end - 1 above a watermark equal to the instance number */
    const int w = deps_in[(size_t)i * n + leader[i]];
    if (end == 0 ? w > x : (w != x || end < x + 2)) return 1;
  }
  int rc = 0;
  for (int i = 0; i < m; ++i) {
    const int L = leader[i], x = number[i], ballot = enc_ballot(b_ord[i], b_rep[i]);
    unsigned ok = 0, resend = 0, nack = 0, com = 0;
    int nb = -1;
    for (int r = 0; r < n; ++r) {
      int* rd = reply_deps ? reply_deps + ((size_t)i * n + r) * n : NULL;
      if (rd) memset(rd, 0, sizeof(int) * (size_t)n);
      if (reply_end) reply_end[(size_t)i * n + r] = 0;
      if (reply_triple) reply_triple[(size_t)i * n + r] = -1;
      if (!((target[i] >> r) & 1u)) continue;
      const size_t c = ((size_t)r * n + L) * e->num_instances + x;
      const int st = e->cl_status[c];
      int stored = 0;
      if (st == CL_COMMITTED) {
        com |= 1u << r, stored = 1;
      } else if (st != CL_NONE && ballot < e->cl_ballot[c]) {
        nack |= 1u << r;
        if (e->largest_ballot[r] > nb) nb = e->largest_ballot[r];
        continue;
      } else if (st == CL_PRE_ACCEPTED && ballot == e->cl_vote[c]) {
        resend |= 1u << r, stored = 1;
      } else if (st == CL_ACCEPTED && ballot == e->cl_vote[c]) {
        continue;
      }
      if (stored) {
        if (rd) memcpy(rd, e->cl_deps + c * (size_t)n, sizeof(int) * (size_t)n);
        if (reply_end) reply_end[(size_t)i * n + r] = e->cl_dend[c];
        if (reply_triple) reply_triple[(size_t)i * n + r] = e->cl_triple[c];
        continue;
      }
      if (ballot > e->largest_ballot[r]) e->largest_ballot[r] = ballot;
      int wm[8];
      memset(wm, 0, sizeof(wm));
      ips_t own = ips_from_watermark(0);
      if (key[i] >= 0) {
        get_top_one_conflicts(e, r, key[i], is_set[i], wm);
        own = ips_from_watermark(wm[L]);
        ips_subtract_one(&own, x);
      }
      ips_t in = ips_from_watermark(deps_in[(size_t)i * n + L]);
      const int end = deps_in_end ? deps_in_end[i] : 0;
      if (end) {
        in.nvalues = end - (x + 1);
        in.values = (int*)malloc(sizeof(int) * (size_t)in.nvalues);
        for (int k = 0; k < in.nvalues; ++k) in.values[k] = x + 1 + k;
      }
      ips_add_all(&own, &in);
      ips_free(&in);
      for (int l = 0; l < n; ++l)
        if (deps_in[(size_t)i * n + l] > wm[l]) wm[l] = deps_in[(size_t)i * n + l];
      if (own.nvalues && (own.watermark != x || own.values[0] != x + 1 ||
                          own.values[own.nvalues - 1] + 1 - own.values[0] != own.nvalues))
        rc = 99;
      e->cl_status[c] = CL_PRE_ACCEPTED, e->cl_ballot[c] = e->cl_vote[c] = ballot;
      e->cl_triple[c] = triple_id ? triple_id[i] : -1;
      store_deps(e, c, L, wm, &own);
      if (key[i] >= 0) conflict_index_put(e, r, key[i], is_set[i], L, x);
      ok |= 1u << r;
      if (rd) memcpy(rd, e->cl_deps + c * (size_t)n, sizeof(int) * (size_t)n);
      if (reply_end) reply_end[(size_t)i * n + r] = e->cl_dend[c];
      if (reply_triple) reply_triple[(size_t)i * n + r] = e->cl_triple[c];
      ips_free(&own);
    }
    if (ok_bits) ok_bits[i] = (uint8_t)ok;
    if (resend_bits) resend_bits[i] = (uint8_t)resend;
    if (nack_bits) nack_bits[i] = (uint8_t)nack;
    if (commit_bits) commit_bits[i] = (uint8_t)com;
    if (nack_ballot) nack_ballot[i] = nb;
  }
  return rc;
}

/* the dependencies stored with one command-log entry: n watermarks (column 0 = -1: known by triple id only) and
 * the end of the own-leader column's explicit values */
int fpo_epx_read_cmdlog_deps(fpo_epx* e, int replica, int leader, int number, int32_t* deps, int32_t* values_end) {
  if (e->num_instances <= 0 || replica < 0 || replica >= e->n || leader < 0 || leader >= e->n || number < 0 ||
      number >= e->num_instances)
    return 1;
  const size_t c = ((size_t)replica * e->n + leader) * e->num_instances + number;
  memcpy(deps, e->cl_deps + c * (size_t)e->n, sizeof(int) * (size_t)e->n);
  *values_end = e->cl_dend[c];
  return 0;
}

/* one command-log entry: kind, ballot, voteBallot (encoded), triple id; and the replica's largestBallot */
int fpo_epx_read_cmdlog(fpo_epx* e, int replica, int leader, int number, int32_t* out5) {
  if (e->num_instances <= 0 || replica < 0 || replica >= e->n || leader < 0 || leader >= e->n || number < 0 ||
      number >= e->num_instances)
    return 1;
  const size_t c = ((size_t)replica * e->n + leader) * e->num_instances + number;
  out5[0] = e->cl_status[c], out5[1] = e->cl_ballot[c], out5[2] = e->cl_vote[c], out5[3] = e->cl_triple[c];
  out5[4] = e->largest_ballot[replica];
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
