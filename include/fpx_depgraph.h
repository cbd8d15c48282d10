/*
 * fpx_depgraph.h -- dependency-graph execution of libfpx: what an EPaxos replica does
 * with every committed (instance, sequenceNumber, dependencies) triple after the GPU pre-accept / accept kernels of
 * include/fpx.h have produced it.  HOST code (irregular pointer chasing; SURVEY.md keeps it on the CPU), plain C ABI,
 * no GPU needed.
 *
 * Reference (paths relative to shared/src/main/scala/frankenpaxos/):
 *
 *   depgraph/DependencyGraph.scala:126-192          the interface: commit, execute / appendExecute /
 *                                                   executeByComponent, updateExecuted, numVertices
 *   depgraph/TarjanDependencyGraph.scala:171-465    FPX_DG_TARJAN: one pass of Tarjan's SCC algorithm interleaved
 *                                                   with the eligibility test; components in reverse topological
 *                                                   order, inside a component sorted by (sequenceNumber, key)
 *   depgraph/ZigzagTarjanDependencyGraph.scala:247-721  FPX_DG_ZIGZAG: the variant epaxos/ReplicaMain.scala:127
 *                                                   deploys -- vertices in one BufferMap per leader, roots taken
 *                                                   round-robin over the leader columns from their executed
 *                                                   watermarks, a component is marked executed the moment it forms
 *   epaxos/Replica.scala:859-917                    the caller: dependencyGraph.commit(instance, sequenceNumber,
 *                                                   dependencies) for every committed triple, then
 *                                                   dependencyGraph.appendExecute(numBlockers, executables, blockers)
 *   compact/IntPrefixSet.scala, epaxos/InstancePrefixSet.scala   the dependency / executed sets
 *
 * A key is (leader, id): epaxos.Instance(replicaIndex, instanceNumber), ordered lexicographically
 * (epaxos/InstanceHelpers.scala:6-12); a plain Int key (DependencyGraphTest) is (0, id) with num_leaders = 1.
 * A dependency set is an InstancePrefixSet: per leader column a watermark (every id below it) plus explicit ids.
 *
 * Order.  Everything the reference fixes is kept: reverse topological order of components, (sequenceNumber, key)
 * inside a component, the column round-robin of the zigzag variant, dependencies visited column by column, the
 * watermark range before the explicit ids.  Where the reference iterates a JVM hash collection -- the roots of
 * TarjanDependencyGraph (`for ((key, vertex) <- vertices)`, :329) and the explicit ids of a set -- its own tests
 * accept every outcome (DependencyGraphTest.scala:188-191, 226-230, 274-281); this library takes ascending key order
 * there.  Quirks kept: FPX_DG_TARJAN ignores a key committed twice (:231-236), FPX_DG_ZIGZAG replaces the vertex
 * unless the key is already executed (`vertices.contains(key)` at :350 compares a Buffer of BufferMaps with a key and
 * is always false) and ignores numBlockers; blockers of the zigzag variant include the next missing id of every column.
 *
 * Not thread-safe per handle (single-threaded like every reference actor, Transport.scala:37-39).
 */
#ifndef FPX_DEPGRAPH_H
#define FPX_DEPGRAPH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { FPX_DG_TARJAN = 0, FPX_DG_ZIGZAG = 1 } fpx_depgraph_kind;

typedef struct {
  int32_t kind;        /* fpx_depgraph_kind */
  int32_t num_leaders; /* columns: EPaxos n; 1 for plain Int keys */
  int32_t gc_every_n;  /* zigzag: garbage collect the vertex columns below the executed watermarks after this many
                          executed commands (ZigzagTarjanDependencyGraphOptions.garbageCollectEveryNCommands, default
                          1000 when <= 0); never observable through this interface */
} fpx_depgraph_config;

typedef struct fpx_depgraph fpx_depgraph;

int32_t fpx_depgraph_create(const fpx_depgraph_config* cfg, fpx_depgraph** out);
int32_t fpx_depgraph_destroy(fpx_depgraph* g);

/* DependencyGraph.commit for n vertices, in array order.  Vertex i: key (leader[i], id[i]), sequence number seq[i]
 * (NULL = all 0: what EPaxos uses with top-k dependencies, Replica.scala:575-578), dependencies =
 *   { (l, x) : x < dep_watermark[i * num_leaders + l] }  U  the explicit ids
 *   (dep_values_leader[j], dep_values_id[j]) for j in dep_values_off[i] .. dep_values_off[i + 1]  (all three NULL = none).
 * FPX_EINVAL (nothing committed): a leader outside [0, num_leaders), a negative id or watermark.
 * FPX_ECAPACITY (nothing committed; commit, commit_epx and update_executed alike): a key more than 2^26 ids ahead of its
 * column's executed watermark -- the columns are dense from the watermark on, one far-away key would cost memory for
 * everything in between. */
int32_t fpx_depgraph_commit(fpx_depgraph* g, int32_t n, const int32_t* leader, const int32_t* id, const int32_t* seq,
                            const int32_t* dep_watermark, const int64_t* dep_values_off,
                            const int32_t* dep_values_leader, const int32_t* dep_values_id);
/* The same for dependencies in the encoding fpx_epx_preaccept / fpx_epx_handle_preaccept / fpx_epx_read_cmdlog_deps
 * produce (include/fpx.h): deps[i * num_leaders + l] watermarks and own_values_end[i * own_stride] (NULL = none): the
 * explicit ids id[i] + 1 .. end - 1 of the instance's OWN leader column, 0 = none.  mask (NULL = all): commit only the
 * vertices with mask[i] != 0 -- e.g. the `fast` output of a pre-accept tick, the `committed` output of fpx_epx_accept. */
int32_t fpx_depgraph_commit_epx(fpx_depgraph* g, int32_t n, const int32_t* leader, const int32_t* id,
                                const int32_t* seq, const int32_t* deps, const int32_t* own_values_end,
                                int32_t own_stride, const uint8_t* mask);
/* DependencyGraph.updateExecuted(keys): keys = { (l, x) : x < watermark[l] } U the n explicit (leader, id) pairs. */
int32_t fpx_depgraph_update_executed(fpx_depgraph* g, const int32_t* watermark, int32_t n, const int32_t* leader,
                                     const int32_t* id);
/* DependencyGraph.executeByComponent(numBlockers) (num_blockers < 0 = None).  The result stays in the handle until the
 * next execute: counts come back here, the contents through fpx_depgraph_read_result.  execute / appendExecute of the
 * reference are the same keys flattened. */
int32_t fpx_depgraph_execute(fpx_depgraph* g, int32_t num_blockers, int64_t* num_executables, int64_t* num_components,
                             int64_t* num_blockers_found);
/* Every pointer may be NULL.  exec_leader / exec_id: the executables in execution order; component_size: how many
 * consecutive executables form each strongly connected component; blocker_leader / blocker_id: the blockers, ascending. */
int32_t fpx_depgraph_read_result(fpx_depgraph* g, int32_t* exec_leader, int32_t* exec_id, int32_t* component_size,
                                 int32_t* blocker_leader, int32_t* blocker_id);
/* number of committed, not yet executed (zigzag: not yet garbage collected) vertices.  The reference's zigzag variant
 * answers the constant 42 (:339); this one counts. */
int64_t fpx_depgraph_num_vertices(fpx_depgraph* g);
/* the executed set's watermark per leader column (IntPrefixSet.getWatermark; num_leaders values) */
int32_t fpx_depgraph_executed_watermark(fpx_depgraph* g, int32_t* watermark);

#ifdef __cplusplus
}
#endif
#endif
