/*
 * fpx_jni.c -- the JNI shim between the reference's JVM actors and the C ABI of include/fpx.h / fpx_wire.h.
 *
 * Class: frankenpaxos.gpu.Native (frankenpaxos_amd/jni/Native.scala; all methods static native returning the
 * int32 status).  Build on a machine with a JDK: `make -C frankenpaxos_amd/jni JAVA_HOME=/path/to/jdk`.  This
 * image has no JDK; tests/jni_stub/ carries a small mock of the JNI function table the shim uses, against which
 * the shim is compiled AND RUN by tests/test_jni_shim.py (argument checking on the CPU, a fused tick on the GPU).
 *
 * Rules the shim keeps (JNI specification):
 *   - no blocking call ever runs inside a Get/ReleasePrimitiveArrayCritical region: libfpx entry points allocate,
 *     copy over PCIe and synchronise the GPU stream, and a critical region stalls the garbage collector for all
 *     of it.  Primitive arrays are COPIED in and out with Get/Set<Type>ArrayRegion (a memcpy each; the payload
 *     crosses PCIe anyway); the zero-copy path is the *Direct natives over direct ByteBuffers (page-locked
 *     memory from hostAlloc), which is also what the reference's Netty transport hands out.
 *   - every array is checked against the batch size before native code touches it: a short array is FPX_EINVAL
 *     (the IllegalArgumentException of a require(...)), never an out-of-bounds read or write.
 */
#include <jni.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/fpx.h"
#include "../../include/fpx_wire.h"

#define CTX(h) ((fpx_ctx*)(intptr_t)(h))

/* ---- checked copies between Java arrays and native buffers ------------------------------------------------ */
static int has(JNIEnv* env, jarray a, jlong need) { return a != NULL && (jlong)(*env)->GetArrayLength(env, a) >= need; }
/* an optional array: absent, or long enough */
static int opt(JNIEnv* env, jarray a, jlong need) { return a == NULL || has(env, a, need); }

static jint* in_ints(JNIEnv* env, jintArray a, jlong n) {
  if (!a || n <= 0) return NULL;
  jint* p = (jint*)malloc((size_t)n * sizeof(jint));
  if (p) (*env)->GetIntArrayRegion(env, a, 0, (jsize)n, p);
  return p;
}
static jlong* in_longs(JNIEnv* env, jlongArray a, jlong n) {
  if (!a || n <= 0) return NULL;
  jlong* p = (jlong*)malloc((size_t)n * sizeof(jlong));
  if (p) (*env)->GetLongArrayRegion(env, a, 0, (jsize)n, p);
  return p;
}
static jbyte* in_bytes(JNIEnv* env, jbyteArray a, jlong n) {
  if (!a || n <= 0) return NULL;
  jbyte* p = (jbyte*)malloc((size_t)n);
  if (p) (*env)->GetByteArrayRegion(env, a, 0, (jsize)n, p);
  return p;
}
static void* out_buf(jarray a, jlong n, size_t elem) { return (a && n > 0) ? calloc((size_t)n, elem) : NULL; }
static void put_ints(JNIEnv* env, jintArray a, jlong n, const jint* p) {
  if (a && p && n > 0) (*env)->SetIntArrayRegion(env, a, 0, (jsize)n, p);
}
static void put_longs(JNIEnv* env, jlongArray a, jlong n, const jlong* p) {
  if (a && p && n > 0) (*env)->SetLongArrayRegion(env, a, 0, (jsize)n, p);
}
static void put_bytes(JNIEnv* env, jbyteArray a, jlong n, const jbyte* p) {
  if (a && p && n > 0) (*env)->SetByteArrayRegion(env, a, 0, (jsize)n, p);
}

/* The sizes libfpx copies with are the HANDLE's (its n, its number of acceptor groups), not what the caller says the
 * handle is: a native that takes numReplicas / numGroups (to size Java arrays on the Scala side) refuses a value that
 * differs from the context's -- FPX_EINVAL, never a short calloc'd reply buffer. */
static int epx_n_is(jlong h, jint numReplicas) {
  int32_t n = 0;
  return fpx_epx_info((fpx_epx*)(intptr_t)h, &n, NULL, NULL) == FPX_OK && n == numReplicas;
}
static int ctx_groups_is(jlong h, jint numGroups) {
  fpx_config c;
  return fpx_get_config(CTX(h), &c) == FPX_OK && c.num_groups == numGroups;
}

/* fpx_leader_phase1b_scan reads one quorum mask per (leader group, acceptor group) */
static int ctx_all_groups_is(jlong h, jint numGroups) {
  fpx_config c;
  return fpx_get_config(CTX(h), &c) == FPX_OK && (jlong)c.num_groups * c.num_leader_groups == numGroups;
}

static int read_config(JNIEnv* env, jintArray jcfg, fpx_config* cfg) {
  jint c[15];
  if (!has(env, jcfg, 15)) return FPX_EINVAL;
  (*env)->GetIntArrayRegion(env, jcfg, 0, 15, c);
  cfg->num_slots = c[0]; cfg->num_replicas = c[1]; cfg->num_groups = c[2]; cfg->num_leader_groups = c[3];
  cfg->f = c[4]; cfg->quorum_kind = c[5]; cfg->grid_rows = c[6]; cfg->grid_cols = c[7]; cfg->num_leaders = c[8];
  cfg->ballot_mode = c[9]; cfg->tally_ways = c[10]; cfg->replica_base = c[11]; cfg->replicas_total = c[12];
  cfg->device = c[13]; cfg->flags = (uint32_t)c[14];
  return FPX_OK;
}

/* long create(int[] cfg: the 15 fpx_config fields in order) -> handle, or -status */
JNIEXPORT jlong JNICALL Java_frankenpaxos_gpu_Native_create(JNIEnv* env, jclass cls, jintArray jcfg) {
  fpx_config cfg;
  int32_t st = read_config(env, jcfg, &cfg);
  if (st) return -(jlong)st;
  fpx_ctx* ctx = NULL;
  st = fpx_create(&cfg, &ctx);
  return st == FPX_OK ? (jlong)(intptr_t)ctx : -(jlong)st;
}

JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_destroy(JNIEnv* env, jclass cls, jlong h) {
  return fpx_destroy(CTX(h));
}

/* ClassicRoundRobin (roundsystem/RoundSystem.scala:60-87); numLeaders < 1 is the require() of :61 */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_roundLeader(JNIEnv* env, jclass cls, jint numLeaders, jint round) {
  return numLeaders < 1 ? -FPX_EINVAL : fpx_round_leader(numLeaders, round);
}

/* Acceptor.handlePhase2a for one tick: multipaxos/Acceptor.scala:184-220 */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_acceptorPhase2a(
    JNIEnv* env, jclass cls, jlong h, jint n, jintArray slot, jintArray round, jintArray value,
    jlongArray targetMask, jlongArray voteBits, jlongArray nackBits, jintArray nackRound) {
  if (n < 0) return FPX_EINVAL;
  if (n == 0) return FPX_OK;
  if (!has(env, slot, n) || !has(env, round, n) || !has(env, value, n) || !opt(env, targetMask, 4 * (jlong)n) ||
      !opt(env, voteBits, 4 * (jlong)n) || !opt(env, nackBits, 4 * (jlong)n) || !opt(env, nackRound, n))
    return FPX_EINVAL;
  jint *s = in_ints(env, slot, n), *r = in_ints(env, round, n), *v = in_ints(env, value, n);
  jlong* t = in_longs(env, targetMask, 4 * (jlong)n);
  jlong *vb = out_buf(voteBits, 4 * (jlong)n, 8), *nb = out_buf(nackBits, 4 * (jlong)n, 8);
  jint* nr = out_buf(nackRound, n, 4);
  int32_t st = fpx_acceptor_phase2a(CTX(h), n, s, r, v, (const uint64_t*)t, (uint64_t*)vb, (uint64_t*)nb, nr);
  put_longs(env, voteBits, 4 * (jlong)n, vb); put_longs(env, nackBits, 4 * (jlong)n, nb); put_ints(env, nackRound, n, nr);
  free(s); free(r); free(v); free(t); free(vb); free(nb); free(nr);
  return st;
}

/* ProxyLeader.handlePhase2a bookkeeping: multipaxos/ProxyLeader.scala:175-215 */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_proxyOpen(JNIEnv* env, jclass cls, jlong h, jint n,
                                                              jintArray slot, jintArray round, jintArray value,
                                                              jbyteArray isNew) {
  if (n < 0) return FPX_EINVAL;
  if (n == 0) return FPX_OK;
  if (!has(env, slot, n) || !has(env, round, n) || !has(env, value, n) || !opt(env, isNew, n)) return FPX_EINVAL;
  jint *s = in_ints(env, slot, n), *r = in_ints(env, round, n), *v = in_ints(env, value, n);
  jbyte* f = out_buf(isNew, n, 1);
  int32_t st = fpx_proxy_open(CTX(h), n, s, r, v, (uint8_t*)f);
  put_bytes(env, isNew, n, f);
  free(s); free(r); free(v); free(f);
  return st;
}

/* ProxyLeader.handlePhase2b: multipaxos/ProxyLeader.scala:217-258 */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_proxyPhase2b(JNIEnv* env, jclass cls, jlong h, jint n,
                                                                 jintArray slot, jintArray round, jlongArray voteBits,
                                                                 jbyteArray newlyChosen, jintArray chosenRound,
                                                                 jintArray chosenValue) {
  if (n < 0) return FPX_EINVAL;
  if (n == 0) return FPX_OK;
  if (!has(env, slot, n) || !has(env, round, n) || !has(env, voteBits, 4 * (jlong)n) || !opt(env, newlyChosen, n) ||
      !opt(env, chosenRound, n) || !opt(env, chosenValue, n))
    return FPX_EINVAL;
  jint *s = in_ints(env, slot, n), *r = in_ints(env, round, n);
  jlong* vb = in_longs(env, voteBits, 4 * (jlong)n);
  jbyte* ch = out_buf(newlyChosen, n, 1);
  jint *cr = out_buf(chosenRound, n, 4), *cv = out_buf(chosenValue, n, 4);
  int32_t st = fpx_proxy_phase2b(CTX(h), n, s, r, (const uint64_t*)vb, (uint8_t*)ch, cr, cv);
  put_bytes(env, newlyChosen, n, ch); put_ints(env, chosenRound, n, cr); put_ints(env, chosenValue, n, cv);
  free(s); free(r); free(vb); free(ch); free(cr); free(cv);
  return st;
}

/* the fused tick (open + Phase2a to the targeted acceptors + tally) */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_phase2Fused(
    JNIEnv* env, jclass cls, jlong h, jint n, jintArray slot, jintArray round, jintArray value,
    jlongArray targetMask, jbyteArray chosen, jintArray chosenRound, jintArray chosenValue, jintArray nackRound) {
  if (n < 0) return FPX_EINVAL;
  if (n == 0) return FPX_OK;
  if (!has(env, slot, n) || !has(env, round, n) || !has(env, value, n) || !opt(env, targetMask, 4 * (jlong)n) ||
      !opt(env, chosen, n) || !opt(env, chosenRound, n) || !opt(env, chosenValue, n) || !opt(env, nackRound, n))
    return FPX_EINVAL;
  jint *s = in_ints(env, slot, n), *r = in_ints(env, round, n), *v = in_ints(env, value, n);
  jlong* t = in_longs(env, targetMask, 4 * (jlong)n);
  jbyte* ch = out_buf(chosen, n, 1);
  jint *cr = out_buf(chosenRound, n, 4), *cv = out_buf(chosenValue, n, 4), *nr = out_buf(nackRound, n, 4);
  int32_t st = fpx_phase2_fused(CTX(h), n, s, r, v, (const uint64_t*)t, (uint8_t*)ch, cr, cv, nr);
  put_bytes(env, chosen, n, ch); put_ints(env, chosenRound, n, cr); put_ints(env, chosenValue, n, cv);
  put_ints(env, nackRound, n, nr);
  free(s); free(r); free(v); free(t); free(ch); free(cr); free(cv); free(nr);
  return st;
}

/* quorums.*.isWriteQuorum / isSuperSetOfWriteQuorum on bitmaps: quorums/QuorumSystem.scala:21,24 */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_quorumEval(JNIEnv* env, jclass cls, jintArray jcfg, jint n,
                                                               jlongArray nodes, jint strict, jbyteArray out) {
  fpx_config cfg;
  int32_t st = read_config(env, jcfg, &cfg);
  if (st) return st;
  if (n < 0 || (n > 0 && (!has(env, nodes, 4 * (jlong)n) || !has(env, out, n)))) return FPX_EINVAL;
  if (n == 0) return FPX_OK;
  jlong* nd = in_longs(env, nodes, 4 * (jlong)n);
  jbyte* o = out_buf(out, n, 1);
  st = fpx_quorum_eval(&cfg, n, (const uint64_t*)nd, strict, (uint8_t*)o);
  put_bytes(env, out, n, o);
  free(nd); free(o);
  return st;
}

/* ---- page-locked batches: direct ByteBuffers over fpx_host_alloc memory ------------------------------
 * A tick's SoA batch lives in direct buffers the JVM fills in place (IntBuffer / LongBuffer views, native
 * byte order), the counterpart of the Netty direct buffers the reference's transport decodes from: no
 * copy, DMA straight out of the buffer. */
JNIEXPORT jobject JNICALL Java_frankenpaxos_gpu_Native_hostAlloc(JNIEnv* env, jclass cls, jlong bytes) {
  void* p = NULL;
  if (fpx_host_alloc(bytes, &p) != FPX_OK) return NULL;
  return (*env)->NewDirectByteBuffer(env, p, bytes);
}

JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_hostFree(JNIEnv* env, jclass cls, jobject buffer) {
  return fpx_host_free(buffer ? (*env)->GetDirectBufferAddress(env, buffer) : NULL);
}

/* address of a direct buffer that must hold `need` bytes; *bad is set when it is present but too small or not direct */
static void* direct(JNIEnv* env, jobject buf, jlong need, int* bad) {
  if (!buf) return NULL;
  void* p = (*env)->GetDirectBufferAddress(env, buf);
  if (!p || (*env)->GetDirectBufferCapacity(env, buf) < need) *bad = 1;
  return p;
}

JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_phase2FusedDirect(
    JNIEnv* env, jclass cls, jlong h, jint n, jobject slot, jobject round, jobject value, jobject targetMask,
    jobject chosen, jobject chosenRound, jobject chosenValue, jobject nackRound) {
  if (n < 0) return FPX_EINVAL;
  int bad = 0;
  const jlong n4 = 4 * (jlong)n, n32 = 32 * (jlong)n;
  const int32_t* s = direct(env, slot, n4, &bad);
  const int32_t* r = direct(env, round, n4, &bad);
  const int32_t* v = direct(env, value, n4, &bad);
  const uint64_t* t = direct(env, targetMask, n32, &bad);
  uint8_t* ch = direct(env, chosen, n, &bad);
  int32_t *cr = direct(env, chosenRound, n4, &bad), *cv = direct(env, chosenValue, n4, &bad);
  int32_t* nr = direct(env, nackRound, n4, &bad);
  if (bad || (n > 0 && (!s || !r || !v))) return FPX_EINVAL;
  return fpx_phase2_fused(CTX(h), n, s, r, v, t, ch, cr, cv, nr);
}

/* ---- the rows around the fused step ---------------------------------------------------------------------- */
/* Acceptor.handlePhase1a: multipaxos/Acceptor.scala:148-182.  bits: promised[4] then nack[4] */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_acceptorPhase1a(JNIEnv* env, jclass cls, jlong h, jint group,
                                                                    jint round, jint chosenWatermark,
                                                                    jlongArray targetMask, jlongArray bits) {
  if (!opt(env, targetMask, 4) || !opt(env, bits, 8)) return FPX_EINVAL;
  jlong t[4], b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (targetMask) (*env)->GetLongArrayRegion(env, targetMask, 0, 4, t);
  int32_t st = fpx_acceptor_phase1a(CTX(h), group, round, chosenWatermark, targetMask ? (const uint64_t*)t : NULL,
                                    (uint64_t*)b, (uint64_t*)b + 4);
  put_longs(env, bits, 8, b);
  return st;
}

/* Leader.handlePhase1b safe values: multipaxos/Leader.scala:306-329, 543-566.  maxSlot[0] */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_leaderPhase1bScan(JNIEnv* env, jclass cls, jlong h,
                                                                      jint chosenWatermark, jint numGroups,
                                                                      jlongArray quorumMasks, jint cap,
                                                                      jintArray maxSlot, jintArray safeRound,
                                                                      jintArray safeValue) {
  if (cap < 0 || numGroups < 1 || !ctx_all_groups_is(h, numGroups) || !has(env, quorumMasks, 4 * (jlong)numGroups) || !opt(env, maxSlot, 1) ||
      !opt(env, safeRound, cap) || !opt(env, safeValue, cap))
    return FPX_EINVAL;
  jlong* q = in_longs(env, quorumMasks, 4 * (jlong)numGroups);
  jint mx = -1;
  jint *sr = out_buf(safeRound, cap, 4), *sv = out_buf(safeValue, cap, 4);
  int32_t st = fpx_leader_phase1b_scan(CTX(h), chosenWatermark, (const uint64_t*)q, cap, &mx, sr, sv);
  put_ints(env, maxSlot, 1, &mx); put_ints(env, safeRound, cap, sr); put_ints(env, safeValue, cap, sv);
  free(q); free(sr); free(sv);
  return st;
}

/* Replica.handleChosen + executeLog: multipaxos/Replica.scala:572-590, 394-447.  state = {executedWatermark, numChosen} */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_replicaChosen(JNIEnv* env, jclass cls, jlong h, jint n,
                                                                  jintArray slot, jintArray value, jbyteArray mask,
                                                                  jintArray state) {
  if (n < 0 || (n > 0 && (!has(env, slot, n) || !has(env, value, n))) || !opt(env, mask, n) || !opt(env, state, 2))
    return FPX_EINVAL;
  jint *s = in_ints(env, slot, n), *v = in_ints(env, value, n), o[2] = {0, 0};
  jbyte* m = in_bytes(env, mask, n);
  int32_t st = fpx_replica_chosen(CTX(h), n, s, v, (const uint8_t*)m, &o[0], &o[1]);
  put_ints(env, state, 2, o);
  free(s); free(v); free(m);
  return st;
}

/* mencius.Replica.handleChosenNoopRange: mencius/Replica.scala:464-485.  state = {executedWatermark, numChosen} */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_replicaChosenNoopRange(JNIEnv* env, jclass cls, jlong h,
                                                                           jint slotStart, jint slotEnd,
                                                                           jintArray state) {
  if (!opt(env, state, 2)) return FPX_EINVAL;
  jint o[2] = {0, 0};
  int32_t st = fpx_replica_chosen_noop_range(CTX(h), slotStart, slotEnd, &o[0], &o[1]);
  put_ints(env, state, 2, o);
  return st;
}

/* Mencius noop ranges, n per call (mencius/Acceptor.scala:237-291, mencius/ProxyLeader.scala:255-303, 355-411): the
 * fused step = open + acceptors + tally.  Bitmaps are n x numGroups x 4 longs. */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_noopRangesFused(
    JNIEnv* env, jclass cls, jlong h, jint n, jint numGroups, jintArray slotStart, jintArray slotEnd, jintArray round,
    jlongArray targetMasks, jlongArray voteBits, jlongArray nackBits, jintArray nackRound, jbyteArray isNew,
    jbyteArray chosen) {
  if (n < 0 || numGroups < 1 || !ctx_groups_is(h, numGroups)) return FPX_EINVAL;
  if (n == 0) return FPX_OK;
  const jlong w = 4 * (jlong)n * numGroups;
  if (!has(env, slotStart, n) || !has(env, slotEnd, n) || !has(env, round, n) || !opt(env, targetMasks, w) ||
      !opt(env, voteBits, w) || !opt(env, nackBits, w) || !opt(env, nackRound, n) || !opt(env, isNew, n) ||
      !opt(env, chosen, n))
    return FPX_EINVAL;
  jint *s = in_ints(env, slotStart, n), *e = in_ints(env, slotEnd, n), *r = in_ints(env, round, n);
  jlong* t = in_longs(env, targetMasks, w);
  jlong *vb = out_buf(voteBits, w, 8), *nb = out_buf(nackBits, w, 8);
  jint* nr = out_buf(nackRound, n, 4);
  jbyte *nw = out_buf(isNew, n, 1), *ch = out_buf(chosen, n, 1);
  int32_t st = fpx_noop_ranges_fused(CTX(h), n, s, e, r, (const uint64_t*)t, (uint64_t*)vb, (uint64_t*)nb, nr,
                                     (uint8_t*)nw, (uint8_t*)ch);
  put_longs(env, voteBits, w, vb); put_longs(env, nackBits, w, nb); put_ints(env, nackRound, n, nr);
  put_bytes(env, isNew, n, nw); put_bytes(env, chosen, n, ch);
  free(s); free(e); free(r); free(t); free(vb); free(nb); free(nr); free(nw); free(ch);
  return st;
}

/* the unfused pieces for one range: bits = vote[numGroups x 4] then nack[numGroups x 4]; nackRound[0] */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_acceptorPhase2aNoopRange(
    JNIEnv* env, jclass cls, jlong h, jint slotStart, jint slotEnd, jint round, jint numGroups, jlongArray targetMasks,
    jlongArray bits, jintArray nackRound) {
  if (numGroups < 1 || !ctx_groups_is(h, numGroups)) return FPX_EINVAL;
  const jlong w = 4 * (jlong)numGroups;
  if (!opt(env, targetMasks, w) || !opt(env, bits, 2 * w) || !opt(env, nackRound, 1)) return FPX_EINVAL;
  jlong* t = in_longs(env, targetMasks, w);
  jlong* b = out_buf(bits, 2 * w, 8);
  jint nr = -1;
  int32_t st = fpx_acceptor_phase2a_noop_range(CTX(h), slotStart, slotEnd, round, (const uint64_t*)t, (uint64_t*)b,
                                               b ? (uint64_t*)b + w : NULL, &nr);
  put_longs(env, bits, 2 * w, b); put_ints(env, nackRound, 1, &nr);
  free(t); free(b);
  return st;
}

JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_proxyOpenNoopRange(JNIEnv* env, jclass cls, jlong h,
                                                                       jint slotStart, jint slotEnd, jint round,
                                                                       jbyteArray isNew) {
  if (!opt(env, isNew, 1)) return FPX_EINVAL;
  jbyte f = 0;
  int32_t st = fpx_proxy_open_noop_range(CTX(h), slotStart, slotEnd, round, (uint8_t*)&f);
  put_bytes(env, isNew, 1, &f);
  return st;
}

JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_proxyPhase2bNoopRange(JNIEnv* env, jclass cls, jlong h,
                                                                          jint slotStart, jint slotEnd, jint round,
                                                                          jint numGroups, jlongArray voteBits,
                                                                          jbyteArray newlyChosen) {
  if (numGroups < 1 || !ctx_groups_is(h, numGroups) || !has(env, voteBits, 4 * (jlong)numGroups) ||
      !opt(env, newlyChosen, 1))
    return FPX_EINVAL;
  jlong* vb = in_longs(env, voteBits, 4 * (jlong)numGroups);
  jbyte c = 0;
  int32_t st = fpx_proxy_phase2b_noop_range(CTX(h), slotStart, slotEnd, round, (const uint64_t*)vb, (uint8_t*)&c);
  put_bytes(env, newlyChosen, 1, &c);
  free(vb);
  return st;
}

/* ---- EPaxos pre-accept fast path: epaxos/Replica.scala:569-600, 633-729, 1159-1419 ------------------------- */
JNIEXPORT jlong JNICALL Java_frankenpaxos_gpu_Native_epxCreate(JNIEnv* env, jclass cls, jint numReplicas,
                                                               jint numKeys, jint device) {
  fpx_epx_config cfg = {numReplicas, numKeys, device, 0};
  fpx_epx* e = NULL;
  int32_t st = fpx_epx_create(&cfg, &e);
  return st == FPX_OK ? (jlong)(intptr_t)e : -(jlong)st;
}

JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_epxDestroy(JNIEnv* env, jclass cls, jlong h) {
  return fpx_epx_destroy((fpx_epx*)(intptr_t)h);
}

/* numReplicas = n of the context (checked against the handle): sizes rank (n x m) and the deps (m x n) */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_epxPreaccept(
    JNIEnv* env, jclass cls, jlong h, jint m, jint numReplicas, jintArray leader, jintArray number, jintArray key,
    jbyteArray isSet, jbyteArray respMask, jbyteArray seenMask, jintArray rank, jbyteArray fast, jintArray deps,
    jintArray leaderDeps, jintArray ownValuesEnd) {
  if (m < 0 || numReplicas < 3 || !epx_n_is(h, numReplicas)) return FPX_EINVAL;
  if (m == 0) return FPX_OK;
  const jlong mn = (jlong)m * numReplicas;
  if (!has(env, leader, m) || !has(env, number, m) || !has(env, key, m) || !has(env, isSet, m) ||
      !has(env, respMask, m) || !opt(env, seenMask, m) || !has(env, rank, mn) || !opt(env, fast, m) ||
      !opt(env, deps, mn) || !opt(env, leaderDeps, mn) || !opt(env, ownValuesEnd, 2 * (jlong)m))
    return FPX_EINVAL;
  jint *l = in_ints(env, leader, m), *nu = in_ints(env, number, m), *k = in_ints(env, key, m), *rk = in_ints(env, rank, mn);
  jbyte *is = in_bytes(env, isSet, m), *rm = in_bytes(env, respMask, m), *sm = in_bytes(env, seenMask, m);
  jbyte* f = out_buf(fast, m, 1);
  jint *d = out_buf(deps, mn, 4), *ld = out_buf(leaderDeps, mn, 4), *ov = out_buf(ownValuesEnd, 2 * (jlong)m, 4);
  int32_t st = fpx_epx_preaccept((fpx_epx*)(intptr_t)h, m, l, nu, k, (const uint8_t*)is, (const uint8_t*)rm,
                                 (const uint8_t*)sm, rk, NULL, (uint8_t*)f, d, ld, ov);
  put_bytes(env, fast, m, f); put_ints(env, deps, mn, d); put_ints(env, leaderDeps, mn, ld);
  put_ints(env, ownValuesEnd, 2 * (jlong)m, ov);
  free(l); free(nu); free(k); free(rk); free(is); free(rm); free(sm); free(f); free(d); free(ld); free(ov);
  return st;
}

/* ---- EPaxos on the command log: Prepare / Accept / handlePreAccept in full (epaxos/Replica.scala:732-860,
 * 1159-1289, 1421-1565, 1632-1757); a context with a command log of numInstances instances per leader */
JNIEXPORT jlong JNICALL Java_frankenpaxos_gpu_Native_epxCreateWithLog(JNIEnv* env, jclass cls, jint numReplicas,
                                                                      jint numKeys, jint device, jint numInstances) {
  fpx_epx_config cfg = {numReplicas, numKeys, device, 0, numInstances};
  fpx_epx* e = NULL;
  int32_t st = fpx_epx_create(&cfg, &e);
  return st == FPX_OK ? (jlong)(intptr_t)e : -(jlong)st;
}

/* replies = okBits | nackBits | commitBits (3 x m bytes); nackBallot m ints; prepareOk = status | voteBallot |
 * triple (3 x m x n ints) */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_epxPrepare(JNIEnv* env, jclass cls, jlong h, jint m,
                                                               jint numReplicas, jintArray leader, jintArray number,
                                                               jintArray ballotOrdering, jintArray ballotReplica,
                                                               jbyteArray targetMask, jbyteArray replies,
                                                               jintArray nackBallot, jintArray prepareOk) {
  if (m < 0 || numReplicas < 3 || !epx_n_is(h, numReplicas)) return FPX_EINVAL;
  if (m == 0) return FPX_OK;
  const jlong mn = (jlong)m * numReplicas;
  if (!has(env, leader, m) || !has(env, number, m) || !has(env, ballotOrdering, m) || !has(env, ballotReplica, m) ||
      !has(env, targetMask, m) || !opt(env, replies, 3 * (jlong)m) || !opt(env, nackBallot, m) || !opt(env, prepareOk, 3 * mn))
    return FPX_EINVAL;
  jint *l = in_ints(env, leader, m), *nu = in_ints(env, number, m), *bo = in_ints(env, ballotOrdering, m),
       *br = in_ints(env, ballotReplica, m);
  jbyte* tg = in_bytes(env, targetMask, m);
  jbyte* rp = out_buf(replies, 3 * (jlong)m, 1);
  jint *nb = out_buf(nackBallot, m, 4), *po = out_buf(prepareOk, 3 * mn, 4);
  uint8_t* r8 = (uint8_t*)rp;
  int32_t st = fpx_epx_prepare((fpx_epx*)(intptr_t)h, m, l, nu, bo, br, (const uint8_t*)tg, r8, r8 ? r8 + m : NULL,
                               r8 ? r8 + 2 * (size_t)m : NULL, nb, po, po ? po + mn : NULL, po ? po + 2 * mn : NULL);
  put_bytes(env, replies, 3 * (jlong)m, rp); put_ints(env, nackBallot, m, nb); put_ints(env, prepareOk, 3 * mn, po);
  free(l); free(nu); free(bo); free(br); free(tg); free(rp); free(nb); free(po);
  return st;
}

/* key / isSet: the triples' commands, key -1 = Noop (updateConflictIndex wherever the triple is stored,
 * Replica.scala:763, 1503, 828); replies = okBits | nackBits | commitBits | committed (4 x m bytes) */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_epxAccept(JNIEnv* env, jclass cls, jlong h, jint m, jintArray leader,
                                                              jintArray number, jintArray ballotOrdering,
                                                              jintArray ballotReplica, jintArray tripleId,
                                                              jintArray key, jbyteArray isSet,
                                                              jbyteArray targetMask, jbyteArray replies,
                                                              jintArray nackBallot) {
  if (m < 0) return FPX_EINVAL;
  if (m == 0) return FPX_OK;
  if (!has(env, leader, m) || !has(env, number, m) || !has(env, ballotOrdering, m) || !has(env, ballotReplica, m) ||
      !has(env, tripleId, m) || !has(env, key, m) || !has(env, isSet, m) || !has(env, targetMask, m) ||
      !opt(env, replies, 4 * (jlong)m) || !opt(env, nackBallot, m))
    return FPX_EINVAL;
  jint *l = in_ints(env, leader, m), *nu = in_ints(env, number, m), *bo = in_ints(env, ballotOrdering, m),
       *br = in_ints(env, ballotReplica, m), *tr = in_ints(env, tripleId, m), *k = in_ints(env, key, m);
  jbyte *tg = in_bytes(env, targetMask, m), *is = in_bytes(env, isSet, m);
  jbyte* rp = out_buf(replies, 4 * (jlong)m, 1);
  jint* nb = out_buf(nackBallot, m, 4);
  uint8_t* r8 = (uint8_t*)rp;
  int32_t st = (!l || !nu || !bo || !br || !tr || !k || !tg || !is || (replies && !rp) || (nackBallot && !nb))
                   ? FPX_ENOMEM
                   : fpx_epx_accept((fpx_epx*)(intptr_t)h, m, l, nu, bo, br, tr, k, (const uint8_t*)is, (const uint8_t*)tg, r8,
                                    r8 ? r8 + m : NULL, r8 ? r8 + 2 * (size_t)m : NULL, nb, r8 ? r8 + 3 * (size_t)m : NULL);
  put_bytes(env, replies, 4 * (jlong)m, rp); put_ints(env, nackBallot, m, nb);
  free(l); free(nu); free(bo); free(br); free(tr); free(k); free(is); free(tg); free(rp); free(nb);
  return st;
}

/* Replica.handleCommit at the replicas of targetMask (fpx_epx_handle_commit): key -1 = Noop; deps m x n with depsValuesEnd m,
 * or both null (the triple by its id alone) */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_epxHandleCommit(JNIEnv* env, jclass cls, jlong h, jint m, jint numReplicas,
                                                                    jintArray leader, jintArray number, jintArray tripleId,
                                                                    jintArray key, jbyteArray isSet, jintArray deps,
                                                                    jintArray depsValuesEnd, jbyteArray targetMask) {
  if (m < 0 || numReplicas < 3 || !epx_n_is(h, numReplicas)) return FPX_EINVAL;
  if (m == 0) return FPX_OK;
  const jlong mn = (jlong)m * numReplicas;
  if (!has(env, leader, m) || !has(env, number, m) || !has(env, tripleId, m) || !has(env, key, m) || !has(env, isSet, m) ||
      !opt(env, deps, mn) || !opt(env, depsValuesEnd, m) || !has(env, targetMask, m) || (depsValuesEnd && !deps))
    return FPX_EINVAL;
  jint *l = in_ints(env, leader, m), *nu = in_ints(env, number, m), *tr = in_ints(env, tripleId, m), *k = in_ints(env, key, m);
  jint* d = deps ? in_ints(env, deps, mn) : NULL;
  jint* de = depsValuesEnd ? in_ints(env, depsValuesEnd, m) : NULL;
  jbyte *is = in_bytes(env, isSet, m), *tg = in_bytes(env, targetMask, m);
  int32_t st = (!l || !nu || !tr || !k || !is || !tg || (deps && !d) || (depsValuesEnd && !de))
                   ? FPX_ENOMEM
                   : fpx_epx_handle_commit((fpx_epx*)(intptr_t)h, m, l, nu, tr, k, (const uint8_t*)is, d, de, (const uint8_t*)tg);
  free(l); free(nu); free(tr); free(k); free(d); free(de); free(is); free(tg);
  return st;
}

/* Dependency-graph execution of committed instances on the device (fpx_epx_execute; Replica.execute, epaxos/Replica.scala:859-917,
 * depgraph/TarjanDependencyGraph.scala:225-276).  deps m x n, depsValuesEnd m (may be null), committed m bytes (may be null = all),
 * first / count n each (the columns are dense), order / component m each (out), counts = {executed, components, needsHostPath} (out) */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_epxExecute(JNIEnv* env, jclass cls, jlong h, jint m, jint numReplicas,
                                                               jintArray leader, jintArray number, jintArray deps,
                                                               jintArray depsValuesEnd, jbyteArray committed, jintArray first,
                                                               jintArray count, jintArray order, jintArray component,
                                                               jintArray counts) {
  if (m < 0 || numReplicas < 3 || !epx_n_is(h, numReplicas)) return FPX_EINVAL;
  const jlong mn = (jlong)m * numReplicas;
  if (!has(env, leader, m) || !has(env, number, m) || !has(env, deps, mn) || !opt(env, depsValuesEnd, m) || !opt(env, committed, m) ||
      !has(env, first, numReplicas) || !has(env, count, numReplicas) || !has(env, order, m) || !has(env, component, m) ||
      !has(env, counts, 3))
    return FPX_EINVAL;
  jint *l = in_ints(env, leader, m), *nu = in_ints(env, number, m), *d = in_ints(env, deps, mn);
  jint* de = depsValuesEnd ? in_ints(env, depsValuesEnd, m) : NULL;
  jbyte* cm = committed ? in_bytes(env, committed, m) : NULL;
  jint *f = in_ints(env, first, numReplicas), *c = in_ints(env, count, numReplicas);
  jint *o = (jint*)malloc(sizeof(jint) * (size_t)(m > 0 ? m : 1)), *co = (jint*)malloc(sizeof(jint) * (size_t)(m > 0 ? m : 1));
  int64_t ne = 0, nc = 0;
  int32_t nh = 0;
  int32_t st = (!l || !nu || !d || !f || !c || !o || !co || (depsValuesEnd && !de) || (committed && !cm))
                   ? FPX_ENOMEM
                   : fpx_epx_execute((fpx_epx*)(intptr_t)h, m, l, nu, d, de, (const uint8_t*)cm, f, c, o, co, &ne, &nc, &nh);
  if (st == FPX_OK) {
    const jint out[3] = {(jint)ne, (jint)nc, (jint)nh};
    put_ints(env, order, ne, o);
    put_ints(env, component, ne, co);
    (*env)->SetIntArrayRegion(env, counts, 0, 3, out);
  }
  free(l); free(nu); free(d); free(de); free(cm); free(f); free(c); free(o); free(co);
  return st;
}

/* key -1 = Noop; depsIn m x n, depsInValuesEnd m (may be null); replies = okBits | resendBits | nackBits | commitBits
 * (4 x m bytes); replyDeps m x n x n, replyEndTriple = valuesEnd | tripleId (2 x m x n ints) */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_epxHandlePreaccept(
    JNIEnv* env, jclass cls, jlong h, jint m, jint numReplicas, jintArray leader, jintArray number,
    jintArray ballotOrdering, jintArray ballotReplica, jintArray key, jbyteArray isSet, jintArray tripleId,
    jintArray depsIn, jintArray depsInValuesEnd, jbyteArray targetMask, jbyteArray replies, jintArray nackBallot,
    jintArray replyDeps, jintArray replyEndTriple) {
  if (m < 0 || numReplicas < 3 || !epx_n_is(h, numReplicas)) return FPX_EINVAL;
  if (m == 0) return FPX_OK;
  const jlong mn = (jlong)m * numReplicas;
  if (!has(env, leader, m) || !has(env, number, m) || !has(env, ballotOrdering, m) || !has(env, ballotReplica, m) ||
      !has(env, key, m) || !has(env, isSet, m) || !opt(env, tripleId, m) || !has(env, depsIn, mn) ||
      !opt(env, depsInValuesEnd, m) || !has(env, targetMask, m) || !opt(env, replies, 4 * (jlong)m) ||
      !opt(env, nackBallot, m) || !opt(env, replyDeps, mn * numReplicas) || !opt(env, replyEndTriple, 2 * mn))
    return FPX_EINVAL;
  jint *l = in_ints(env, leader, m), *nu = in_ints(env, number, m), *bo = in_ints(env, ballotOrdering, m),
       *br = in_ints(env, ballotReplica, m), *k = in_ints(env, key, m), *tr = in_ints(env, tripleId, m),
       *di = in_ints(env, depsIn, mn), *de = in_ints(env, depsInValuesEnd, m);
  jbyte *is = in_bytes(env, isSet, m), *tg = in_bytes(env, targetMask, m);
  jbyte* rp = out_buf(replies, 4 * (jlong)m, 1);
  jint *nb = out_buf(nackBallot, m, 4), *rd = out_buf(replyDeps, mn * numReplicas, 4), *re = out_buf(replyEndTriple, 2 * mn, 4);
  uint8_t* r8 = (uint8_t*)rp;
  int32_t st = (!l || !nu || !bo || !br || !k || !di || !is || !tg || (tripleId && !tr) || (depsInValuesEnd && !de) ||
                (replies && !rp) || (nackBallot && !nb) || (replyDeps && !rd) || (replyEndTriple && !re))
                   ? FPX_ENOMEM
                   : fpx_epx_handle_preaccept((fpx_epx*)(intptr_t)h, m, l, nu, bo, br, k, (const uint8_t*)is, tr, di, de,
                                              (const uint8_t*)tg, r8, r8 ? r8 + m : NULL, r8 ? r8 + 2 * (size_t)m : NULL,
                                              r8 ? r8 + 3 * (size_t)m : NULL, nb, rd, re, re ? re + mn : NULL);
  put_bytes(env, replies, 4 * (jlong)m, rp); put_ints(env, nackBallot, m, nb);
  put_ints(env, replyDeps, mn * numReplicas, rd); put_ints(env, replyEndTriple, 2 * mn, re);
  free(l); free(nu); free(bo); free(br); free(k); free(tr); free(di); free(de); free(is); free(tg); free(rp); free(nb);
  free(rd); free(re);
  return st;
}

/* entry = kind, ballot, voteBallot, triple id, the replica's largestBallot, then the n stored dependency watermarks
 * and the own column's values end (5 + n + 1 ints) */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_epxReadCmdlog(JNIEnv* env, jclass cls, jlong h, jint numReplicas,
                                                                  jint replica, jint leader, jint number,
                                                                  jintArray entry) {
  if (numReplicas < 3 || numReplicas > 7 || !epx_n_is(h, numReplicas) || !has(env, entry, 6 + (jlong)numReplicas))
    return FPX_EINVAL;
  jint out[5 + 7 + 1];
  int32_t st = fpx_epx_read_cmdlog((fpx_epx*)(intptr_t)h, replica, leader, number, out);
  if (st == FPX_OK) st = fpx_epx_read_cmdlog_deps((fpx_epx*)(intptr_t)h, replica, leader, number, out + 5, out + 5 + numReplicas);
  if (st == FPX_OK) (*env)->SetIntArrayRegion(env, entry, 0, 6 + numReplicas, out);
  return st;
}

/* ---- multi-GPU: the RCCL communicator behind the C ABI (fpx_comm_*) ---------------------------------------- */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_commUniqueId(JNIEnv* env, jclass cls, jbyteArray id) {
  if (!has(env, id, FPX_COMM_ID_BYTES)) return FPX_EINVAL;
  uint8_t b[FPX_COMM_ID_BYTES];
  int32_t st = fpx_comm_unique_id(b);
  if (st == FPX_OK) put_bytes(env, id, FPX_COMM_ID_BYTES, (const jbyte*)b);
  return st;
}

JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_commCreate(JNIEnv* env, jclass cls, jlong h, jbyteArray id,
                                                               jint rank, jint world) {
  if (!has(env, id, FPX_COMM_ID_BYTES)) return FPX_EINVAL;
  uint8_t b[FPX_COMM_ID_BYTES];
  (*env)->GetByteArrayRegion(env, id, 0, FPX_COMM_ID_BYTES, (jbyte*)b);
  return fpx_comm_create(CTX(h), b, rank, world);
}

JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_commDestroy(JNIEnv* env, jclass cls, jlong h) {
  return fpx_comm_destroy(CTX(h));
}

/* ---- wire adapter (include/fpx_wire.h): a tick of ProxyLeaderInbound byte arrays, packed into one direct buffer
 * with n + 1 offsets, decoded straight into the SoA arrays of a batch.  fields = kind, slot, round, isNoop,
 * valueLen, groupIndex, acceptorIndex (7 x n ints); valueOff n longs; returns the status, badIndex[0] on error */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_wireDecodeProxyLeaderInbound(
    JNIEnv* env, jclass cls, jobject buf, jlongArray offsets, jint n, jintArray fields, jlongArray valueOff,
    jintArray badIndex) {
  if (n < 0 || !has(env, offsets, (jlong)n + 1) || !has(env, fields, 7 * (jlong)n) || !opt(env, valueOff, n) ||
      !opt(env, badIndex, 1))
    return FPX_EINVAL;
  if (n == 0) return FPX_OK;
  jlong* off = in_longs(env, offsets, (jlong)n + 1);
  int bad = 0;
  const uint8_t* b = direct(env, buf, 0, &bad);
  /* the decoder checks every offset against the buffer's CAPACITY before it parses a byte (offsets such as
   * [0, 10^9, 5] pass an off[n]-only check) */
  const jlong capacity = b ? (*env)->GetDirectBufferCapacity(env, buf) : -1;
  if (bad || !b || !off || capacity < 0) {
    free(off);
    return FPX_EINVAL;
  }
  jint* f = out_buf(fields, 7 * (jlong)n, 4);
  jlong* vo = (jlong*)calloc((size_t)n, 8);
  jint bi = -1;
  int32_t st = fpx_wire_decode_proxy_leader_inbound(b, (int64_t)capacity, (const int64_t*)off, n, f, f + n, f + 2 * (size_t)n,
                                                    f + 3 * (size_t)n, (int64_t*)vo, f + 4 * (size_t)n,
                                                    f + 5 * (size_t)n, f + 6 * (size_t)n, &bi);
  put_ints(env, fields, 7 * (jlong)n, f); put_longs(env, valueOff, n, vo); put_ints(env, badIndex, 1, &bi);
  free(off); free(f); free(vo);
  return st;
}

/* ---- the acceptors' half of Phase 1, and the log window -------------------------------------------------------- */
/* Phase1b.info of one acceptor (multipaxos/Acceptor.scala:166-178): returns the number of votes at or above the
 * watermark (>= 0; the first min(count, cap) are written), or -status */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_acceptorPhase1bInfo(JNIEnv* env, jclass cls, jlong h, jint group,
                                                                        jint replica, jint chosenWatermark, jint cap,
                                                                        jintArray slot, jintArray voteRound,
                                                                        jintArray voteValue) {
  if (cap < 0 || (cap > 0 && (!has(env, slot, cap) || !has(env, voteRound, cap) || !has(env, voteValue, cap))))
    return -FPX_EINVAL;
  jint *s = out_buf(slot, cap, 4), *r = out_buf(voteRound, cap, 4), *v = out_buf(voteValue, cap, 4);
  int32_t count = 0;
  int32_t st = (cap > 0 && (!s || !r || !v)) ? FPX_ENOMEM
                                              : fpx_acceptor_phase1b_info(CTX(h), group, replica, chosenWatermark, cap, &count, s, r, v);
  const jlong k = count < cap ? count : cap;
  if (st == FPX_OK) { put_ints(env, slot, k, s); put_ints(env, voteRound, k, r); put_ints(env, voteValue, k, v); }
  free(s); free(r); free(v);
  return st == FPX_OK ? count : -st;
}

JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_recycleSlots(JNIEnv* env, jclass cls, jlong h, jint firstSlot, jint count) {
  return fpx_recycle_slots(CTX(h), firstSlot, count);
}

JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_proxyForget(JNIEnv* env, jclass cls, jlong h, jint firstSlot, jint count) {
  return fpx_proxy_forget(CTX(h), firstSlot, count);
}

/* a tick of AcceptorInbound byte arrays (Phase1a / Phase2a): fields = kind, slot, round, isNoop, valueLen,
 * chosenWatermark (6 x n ints); valueOff n longs */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_wireDecodeAcceptorInbound(
    JNIEnv* env, jclass cls, jobject buf, jlongArray offsets, jint n, jintArray fields, jlongArray valueOff,
    jintArray badIndex) {
  if (n < 0 || !has(env, offsets, (jlong)n + 1) || !has(env, fields, 6 * (jlong)n) || !opt(env, valueOff, n) ||
      !opt(env, badIndex, 1))
    return FPX_EINVAL;
  if (n == 0) return FPX_OK;
  jlong* off = in_longs(env, offsets, (jlong)n + 1);
  int bad = 0;
  const uint8_t* b = direct(env, buf, 0, &bad);
  const jlong capacity = b ? (*env)->GetDirectBufferCapacity(env, buf) : -1;
  if (bad || !b || !off || capacity < 0) {
    free(off);
    return FPX_EINVAL;
  }
  jint* f = out_buf(fields, 6 * (jlong)n, 4);
  jlong* vo = (jlong*)calloc((size_t)n, 8);
  jint bi = -1;
  int32_t st = (!f || !vo) ? FPX_ENOMEM
                           : fpx_wire_decode_acceptor_inbound(b, (int64_t)capacity, (const int64_t*)off, n, f, f + n,
                                                              f + 2 * (size_t)n, f + 3 * (size_t)n, (int64_t*)vo,
                                                              f + 4 * (size_t)n, f + 5 * (size_t)n, &bi);
  put_ints(env, fields, 6 * (jlong)n, f); put_longs(env, valueOff, n, vo); put_ints(env, badIndex, 1, &bi);
  free(off); free(f); free(vo);
  return st;
}

/* LeaderInbound{Phase1b} into a direct buffer: the serialised CommandBatchOrNoop of entry j is
 * values[valueOff[j] .. + valueLen[j]) (a direct buffer too), Noop where isNoop[j] != 0.  Returns the length, or the
 * negated length needed when `out` is too small, or Long.MinValue on a bad argument */
JNIEXPORT jlong JNICALL Java_frankenpaxos_gpu_Native_wireEncodeLeaderPhase1b(
    JNIEnv* env, jclass cls, jobject out, jint groupIndex, jint acceptorIndex, jint round, jint nInfo, jintArray slot,
    jintArray voteRound, jobject values, jlongArray valueOff, jintArray valueLen, jbyteArray isNoop) {
  const jlong BAD = INT64_MIN;
  if (nInfo < 0 || (nInfo > 0 && (!has(env, slot, nInfo) || !has(env, voteRound, nInfo) || !has(env, valueOff, nInfo) ||
                                  !has(env, valueLen, nInfo))) || !opt(env, isNoop, nInfo))
    return BAD;
  int bad = 0;
  uint8_t* o = direct(env, out, 0, &bad);
  const uint8_t* vals = direct(env, values, 0, &bad);
  if (bad || !o) return BAD;
  const jlong ocap = (*env)->GetDirectBufferCapacity(env, out);
  const jlong vcap = vals ? (*env)->GetDirectBufferCapacity(env, values) : 0;
  jint *s = in_ints(env, slot, nInfo), *r = in_ints(env, voteRound, nInfo), *vl = in_ints(env, valueLen, nInfo);
  jlong* vo = in_longs(env, valueOff, nInfo);
  jbyte* nz = in_bytes(env, isNoop, nInfo);
  jlong len = BAD;
  int ok = nInfo == 0 || (s && r && vl && vo);
  for (jint j = 0; ok && j < nInfo; ++j)   /* every value must lie inside the values buffer (or be a Noop) */
    if (!(nz && nz[j]) && (vl[j] < 0 || vo[j] < 0 || vo[j] + vl[j] > vcap)) ok = 0;
  if (ok)
    len = fpx_wire_encode_leader_phase1b(o, ocap, groupIndex, acceptorIndex, round, nInfo, s, r, vals, (const int64_t*)vo, vl,
                                         (const uint8_t*)nz);
  free(s); free(r); free(vl); free(vo); free(nz);
  return len;
}

JNIEXPORT jlong JNICALL Java_frankenpaxos_gpu_Native_wireEncodeLeaderNack(JNIEnv* env, jclass cls, jobject out, jint round) {
  int bad = 0;
  uint8_t* o = direct(env, out, 0, &bad);
  if (bad || !o) return INT64_MIN;
  return fpx_wire_encode_leader_nack(o, (*env)->GetDirectBufferCapacity(env, out), round);
}

/* Acceptor.round (multipaxos/Acceptor.scala:95) of one acceptor: >= -1, or -(status + 1) <= -2 on error.  What a
 * Nack to a stale Phase1a carries (:155-162). */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_acceptorRound(JNIEnv* env, jclass cls, jlong h, jint group, jint replica) {
  int32_t round = -1;
  int32_t st = fpx_read_acceptor(CTX(h), group, replica, &round, NULL, NULL, NULL, NULL);
  return st == FPX_OK ? round : -(st + 1);
}

/* Acceptor.maxVotedSlot over the rows [firstRow, firstRow + count) of one acceptor (fpx_acceptor_max_voted_in; the read
 * path, multipaxos/Acceptor.scala:222-254): >= -1, or -(status + 1) <= -2 on error. */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_acceptorMaxVotedIn(JNIEnv* env, jclass cls, jlong h, jint group, jint replica,
                                                                       jint firstRow, jint count) {
  int32_t row = -1;
  int32_t st = fpx_acceptor_max_voted_in(CTX(h), group, replica, firstRow, count, &row);
  return st == FPX_OK ? row : -(st + 1);
}

/* ---- calls in flight on page-locked batches (fpx_phase2_fused_submit / _wait): every buffer a direct ByteBuffer over
 * hostAlloc memory.  submit returns the ticket (>= 0) or -status */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_phase2FusedSubmitDirect(
    JNIEnv* env, jclass cls, jlong h, jint n, jobject slot, jobject round, jobject value, jobject targetMask,
    jobject chosen, jobject chosenRound, jobject chosenValue, jobject nackRound) {
  if (n <= 0) return -FPX_EINVAL;
  int bad = 0;
  const jlong n4 = 4 * (jlong)n, n32 = 32 * (jlong)n;
  const int32_t* s = direct(env, slot, n4, &bad);
  const int32_t* r = direct(env, round, n4, &bad);
  const int32_t* v = direct(env, value, n4, &bad);
  const uint64_t* t = direct(env, targetMask, n32, &bad);
  uint8_t* ch = direct(env, chosen, n, &bad);
  int32_t *cr = direct(env, chosenRound, n4, &bad), *cv = direct(env, chosenValue, n4, &bad);
  int32_t* nr = direct(env, nackRound, n4, &bad);
  if (bad || !s || !r || !v) return -FPX_EINVAL;
  int32_t ticket = -1;
  const int32_t st = fpx_phase2_fused_submit(CTX(h), n, s, r, v, t, ch, cr, cv, nr, &ticket);
  return st == FPX_OK ? ticket : -st;
}

JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_phase2FusedWait(JNIEnv* env, jclass cls, jlong h, jint ticket) {
  return fpx_phase2_fused_wait(CTX(h), ticket);
}

/* K8 Replica.handlePrepareOk (epaxos/Replica.scala:1759-1884): prepareOk = epxPrepare's output (status | voteBallot |
 * triple, 3 x m x numReplicas), respMask = its okBits; decision = action | source | triple (3 x m ints) */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_epxHandlePrepareOks(JNIEnv* env, jclass cls, jlong h, jint m,
                                                                        jint numReplicas, jintArray leader,
                                                                        jintArray number, jintArray ballotOrdering,
                                                                        jintArray ballotReplica, jbyteArray respMask,
                                                                        jintArray prepareOk, jint asIntended,
                                                                        jintArray decision) {
  if (m < 0 || numReplicas < 3 || !epx_n_is(h, numReplicas)) return FPX_EINVAL;
  if (m == 0) return FPX_OK;
  const jlong mn = (jlong)m * numReplicas;
  if (!has(env, leader, m) || !has(env, number, m) || !has(env, ballotOrdering, m) || !has(env, ballotReplica, m) ||
      !has(env, respMask, m) || !has(env, prepareOk, 3 * mn) || !has(env, decision, 3 * (jlong)m))
    return FPX_EINVAL;
  jint *l = in_ints(env, leader, m), *nu = in_ints(env, number, m), *bo = in_ints(env, ballotOrdering, m),
       *br = in_ints(env, ballotReplica, m), *po = in_ints(env, prepareOk, 3 * mn);
  jbyte* mk = in_bytes(env, respMask, m);
  jint* d = out_buf(decision, 3 * (jlong)m, 4);
  int32_t st = (!l || !nu || !bo || !br || !po || !mk || !d)
                   ? FPX_ENOMEM
                   : fpx_epx_handle_prepare_oks((fpx_epx*)(intptr_t)h, m, l, nu, bo, br, (const uint8_t*)mk, po, po + mn,
                                                po + 2 * mn, asIntended, d, d + m, d + 2 * (size_t)m);
  put_ints(env, decision, 3 * (jlong)m, d);
  free(l); free(nu); free(bo); free(br); free(po); free(mk); free(d);
  return st;
}

/* ---- Mencius on the wire (fpx_wire.h, mencius/Mencius.proto) ----------------------------------------------------
 * a tick of mencius ProxyLeaderInbound byte arrays: fields = kind, slot (slotStartInclusive of a range), slotEnd
 * (slotEndExclusive, -1 for a single slot), round, isNoop, valueLen, groupIndex, acceptorIndex (8 x n ints) */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_wireMenciusDecodeProxyLeaderInbound(
    JNIEnv* env, jclass cls, jobject buf, jlongArray offsets, jint n, jintArray fields, jlongArray valueOff,
    jintArray badIndex) {
  if (n < 0 || !has(env, offsets, (jlong)n + 1) || !has(env, fields, 8 * (jlong)n) || !opt(env, valueOff, n) ||
      !opt(env, badIndex, 1))
    return FPX_EINVAL;
  if (n == 0) return FPX_OK;
  jlong* off = in_longs(env, offsets, (jlong)n + 1);
  int bad = 0;
  const uint8_t* b = direct(env, buf, 0, &bad);
  const jlong capacity = b ? (*env)->GetDirectBufferCapacity(env, buf) : -1;
  if (bad || !b || !off || capacity < 0) {
    free(off);
    return FPX_EINVAL;
  }
  jint* f = out_buf(fields, 8 * (jlong)n, 4);
  jlong* vo = (jlong*)calloc((size_t)n, 8);
  jint bi = -1;
  const size_t N = (size_t)n;
  int32_t st = (!f || !vo) ? FPX_ENOMEM
                           : fpx_wire_mencius_decode_proxy_leader_inbound(b, (int64_t)capacity, (const int64_t*)off, n, f, f + N,
                                                                          f + 2 * N, f + 3 * N, f + 4 * N, (int64_t*)vo,
                                                                          f + 5 * N, f + 6 * N, f + 7 * N, &bi);
  put_ints(env, fields, 8 * (jlong)n, f); put_longs(env, valueOff, n, vo); put_ints(env, badIndex, 1, &bi);
  free(off); free(f); free(vo);
  return st;
}

/* a tick of mencius AcceptorInbound byte arrays (Phase1a / Phase2a / Phase2aNoopRange): fields = kind, slot, slotEnd,
 * round, isNoop, valueLen, chosenWatermark (7 x n ints) */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_wireMenciusDecodeAcceptorInbound(
    JNIEnv* env, jclass cls, jobject buf, jlongArray offsets, jint n, jintArray fields, jlongArray valueOff,
    jintArray badIndex) {
  if (n < 0 || !has(env, offsets, (jlong)n + 1) || !has(env, fields, 7 * (jlong)n) || !opt(env, valueOff, n) ||
      !opt(env, badIndex, 1))
    return FPX_EINVAL;
  if (n == 0) return FPX_OK;
  jlong* off = in_longs(env, offsets, (jlong)n + 1);
  int bad = 0;
  const uint8_t* b = direct(env, buf, 0, &bad);
  const jlong capacity = b ? (*env)->GetDirectBufferCapacity(env, buf) : -1;
  if (bad || !b || !off || capacity < 0) {
    free(off);
    return FPX_EINVAL;
  }
  jint* f = out_buf(fields, 7 * (jlong)n, 4);
  jlong* vo = (jlong*)calloc((size_t)n, 8);
  jint bi = -1;
  const size_t N = (size_t)n;
  int32_t st = (!f || !vo) ? FPX_ENOMEM
                           : fpx_wire_mencius_decode_acceptor_inbound(b, (int64_t)capacity, (const int64_t*)off, n, f, f + N,
                                                                      f + 2 * N, f + 3 * N, f + 4 * N, (int64_t*)vo, f + 5 * N,
                                                                      f + 6 * N, &bi);
  put_ints(env, fields, 7 * (jlong)n, f); put_longs(env, valueOff, n, vo); put_ints(env, badIndex, 1, &bi);
  free(off); free(f); free(vo);
  return st;
}

/* mencius LeaderInbound{Nack} (field 7, not MultiPaxos' 6) */
JNIEXPORT jlong JNICALL Java_frankenpaxos_gpu_Native_wireMenciusEncodeLeaderNack(JNIEnv* env, jclass cls, jobject out,
                                                                                 jint round) {
  int bad = 0;
  uint8_t* o = direct(env, out, 0, &bad);
  if (bad || !o) return INT64_MIN;
  return fpx_wire_mencius_encode_leader_nack(o, (*env)->GetDirectBufferCapacity(env, out), round);
}

/* ---- EPaxos on the wire (epaxos/EPaxos.proto: ReplicaInbound) ---------------------------------------------------
 * a tick of ReplicaInbound byte arrays: fields = kind, instanceLeader, instanceNumber, ballotOrdering, ballotReplica,
 * replicaIndex, sequenceNumber, voteBallotOrdering, voteBallotReplica, status, isNoop, cmdLen, depsNumReplicas
 * (13 x n ints); cmdOff n longs; depsWatermark n x maxReplicas ints; the explicit ids of message i are
 * values[valuesOff[i] .. valuesOff[i + 1]) (leader) and values[valuesCap + the same] (id): valuesOff n + 1 longs,
 * values 2 x valuesCap ints.  FPX_ECAPACITY: valuesOff[n] = the capacity needed, call again. */
JNIEXPORT jint JNICALL Java_frankenpaxos_gpu_Native_wireEpaxosDecodeReplicaInbound(
    JNIEnv* env, jclass cls, jobject buf, jlongArray offsets, jint n, jint maxReplicas, jintArray fields,
    jlongArray cmdOff, jintArray depsWatermark, jlongArray valuesOff, jint valuesCap, jintArray values,
    jintArray badIndex) {
  if (n < 0 || maxReplicas < 1 || maxReplicas > 8 || valuesCap < 0 || !has(env, offsets, (jlong)n + 1) ||
      !has(env, fields, 13 * (jlong)n) || !opt(env, cmdOff, n) || !opt(env, depsWatermark, (jlong)n * maxReplicas) ||
      !has(env, valuesOff, (jlong)n + 1) || (valuesCap > 0 && !has(env, values, 2 * (jlong)valuesCap)) ||
      !opt(env, badIndex, 1))
    return FPX_EINVAL;
  if (n == 0) return FPX_OK;
  jlong* off = in_longs(env, offsets, (jlong)n + 1);
  int bad = 0;
  const uint8_t* b = direct(env, buf, 0, &bad);
  const jlong capacity = b ? (*env)->GetDirectBufferCapacity(env, buf) : -1;
  if (bad || !b || !off || capacity < 0) {
    free(off);
    return FPX_EINVAL;
  }
  const size_t N = (size_t)n;
  jint* f = out_buf(fields, 13 * (jlong)n, 4);
  jlong* co = (jlong*)calloc(N, 8);
  jint* dw = (jint*)calloc(N * (size_t)maxReplicas, 4);
  jlong* vo = (jlong*)calloc(N + 1, 8);
  jint* vals = (jint*)calloc(2 * (size_t)valuesCap + 1, 4);
  jint bi = -1;
  int32_t st = (!f || !co || !dw || !vo || !vals)
                   ? FPX_ENOMEM
                   : fpx_wire_epaxos_decode_replica_inbound(b, (int64_t)capacity, (const int64_t*)off, n, maxReplicas, f, f + N,
                                                            f + 2 * N, f + 3 * N, f + 4 * N, f + 5 * N, f + 6 * N, f + 7 * N,
                                                            f + 8 * N, f + 9 * N, f + 10 * N, (int64_t*)co, f + 11 * N,
                                                            f + 12 * N, dw, (int64_t*)vo, (int64_t)valuesCap, vals,
                                                            vals + valuesCap, &bi);
  put_ints(env, fields, 13 * (jlong)n, f); put_longs(env, cmdOff, n, co);
  put_ints(env, depsWatermark, (jlong)n * maxReplicas, dw); put_longs(env, valuesOff, (jlong)n + 1, vo);
  put_ints(env, values, 2 * (jlong)valuesCap, vals); put_ints(env, badIndex, 1, &bi);
  free(off); free(f); free(co); free(dw); free(vo); free(vals);
  return st;
}

/* ONE ReplicaInbound into a direct buffer.  head = kind, instanceLeader, instanceNumber, ballotOrdering, ballotReplica,
 * replicaIndex, sequenceNumber (-1: absent), voteBallotOrdering, voteBallotReplica, status, isNoop (-1: no
 * CommandOrNoop) (11 ints); command = the serialised CommandOrNoop (direct, commandLen bytes at commandOff);
 * numReplicas < 0: no dependencies, else depsWatermark[numReplicas] and the explicit ids values = leader[numValues] |
 * id[numValues].  Returns the length, the negated length needed, or Long.MinValue on a bad argument. */
JNIEXPORT jlong JNICALL Java_frankenpaxos_gpu_Native_wireEpaxosEncodeReplicaInbound(
    JNIEnv* env, jclass cls, jobject out, jintArray head, jobject command, jlong commandOff, jint commandLen,
    jint numReplicas, jintArray depsWatermark, jint numValues, jintArray values) {
  const jlong BAD = INT64_MIN;
  if (!has(env, head, 11) || numValues < 0 || commandLen < 0 || commandOff < 0 ||
      (numReplicas > 0 && !has(env, depsWatermark, numReplicas)) || (numValues > 0 && !has(env, values, 2 * (jlong)numValues)))
    return BAD;
  int bad = 0;
  uint8_t* o = direct(env, out, 0, &bad);
  const uint8_t* cmd = direct(env, command, 0, &bad);
  if (bad || !o) return BAD;
  if (commandLen > 0 && (!cmd || commandOff + commandLen > (*env)->GetDirectBufferCapacity(env, command))) return BAD;
  jint h[11];
  (*env)->GetIntArrayRegion(env, head, 0, 11, h);
  jint* dw = in_ints(env, depsWatermark, numReplicas);
  jint* vals = in_ints(env, values, 2 * (jlong)numValues);
  if ((numReplicas > 0 && !dw) || (numValues > 0 && !vals)) {
    free(dw); free(vals);
    return BAD;
  }
  fpx_wire_epx_msg m;
  memset(&m, 0, sizeof(m));
  m.kind = h[0], m.instance_leader = h[1], m.instance_number = h[2], m.ballot_ordering = h[3], m.ballot_replica = h[4];
  m.replica_index = h[5], m.sequence_number = h[6], m.has_sequence_number = h[6] >= 0;
  m.vote_ballot_ordering = h[7], m.vote_ballot_replica = h[8], m.status = h[9], m.is_noop = h[10];
  m.command = cmd ? cmd + commandOff : NULL, m.command_len = commandLen;
  m.num_replicas = numReplicas, m.deps_watermark = dw, m.num_values = numValues;
  m.values_leader = vals, m.values_id = vals ? vals + numValues : NULL;
  const jlong len = fpx_wire_epaxos_encode_replica_inbound(o, (*env)->GetDirectBufferCapacity(env, out), &m);
  free(dw); free(vals);
  return len;
}
