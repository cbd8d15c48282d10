/* tests/jni_stub/mock_jvm.c -- test infrastructure: the handful of JNIEnv functions of tests/jni_stub/jni.h over
 * plain heap blocks, plus helpers the python test drives through ctypes to play the JVM's part: make arrays, read
 * them back, call a native.  An out-of-range region access aborts loudly (a real JVM throws
 * ArrayIndexOutOfBoundsException): the shim must never cause one. */
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct _jobject {
  int kind;     /* 4 = int[], 8 = long[], 1 = byte[], 0 = direct buffer */
  jlong len;    /* elements, or capacity in bytes */
  void* data;
  int owns;
};

static void check(jobject a, int kind, jsize start, jsize len, const char* what) {
  if (!a || a->kind != kind || start < 0 || len < 0 || (jlong)start + len > a->len) {
    fprintf(stderr, "mock JVM: ArrayIndexOutOfBoundsException in %s (kind %d, start %d, len %d, array length %lld)\n",
            what, a ? a->kind : -1, (int)start, (int)len, a ? (long long)a->len : -1LL);
    abort();
  }
}
static jsize m_len(JNIEnv* e, jarray a) { (void)e; return (jsize)a->len; }
#define REGION(NAME, T, KIND)                                                                                      \
  static void m_get_##NAME(JNIEnv* e, jarray a, jsize s, jsize n, T* buf) {                                         \
    (void)e; check(a, KIND, s, n, "Get" #NAME "ArrayRegion"); memcpy(buf, (T*)a->data + s, (size_t)n * sizeof(T)); \
  }                                                                                                                 \
  static void m_set_##NAME(JNIEnv* e, jarray a, jsize s, jsize n, const T* buf) {                                   \
    (void)e; check(a, KIND, s, n, "Set" #NAME "ArrayRegion"); memcpy((T*)a->data + s, buf, (size_t)n * sizeof(T)); \
  }
REGION(Int, jint, 4)
REGION(Long, jlong, 8)
REGION(Byte, jbyte, 1)
static jobject m_new_direct(JNIEnv* e, void* p, jlong cap) {
  (void)e;
  jobject o = (jobject)calloc(1, sizeof(*o));
  o->kind = 0, o->len = cap, o->data = p;
  return o;
}
static void* m_addr(JNIEnv* e, jobject b) { (void)e; return (b && b->kind == 0) ? b->data : NULL; }
static jlong m_cap(JNIEnv* e, jobject b) { (void)e; return (b && b->kind == 0) ? b->len : -1; }

static const struct JNINativeInterface_ TABLE = {m_len,      m_get_Int,  m_set_Int,    m_get_Long, m_set_Long,
                                                 m_get_Byte, m_set_Byte, m_new_direct, m_addr,     m_cap};
static JNIEnv ENV = &TABLE;

/* ---- what the python test calls ------------------------------------------------------------------------ */
JNIEnv* mock_env(void) { return &ENV; }
jobject mock_new_array(int kind, jlong len, const void* init) {
  jobject o = (jobject)calloc(1, sizeof(*o));
  o->kind = kind, o->len = len, o->owns = 1;
  o->data = calloc((size_t)(len > 0 ? len : 1), (size_t)kind);
  if (init && len > 0) memcpy(o->data, init, (size_t)len * (size_t)kind);
  return o;
}
jobject mock_new_direct(jlong capacity) {
  jobject o = m_new_direct(&ENV, calloc((size_t)(capacity > 0 ? capacity : 1), 1), capacity);
  o->owns = 1;
  return o;
}
void* mock_data(jobject o) { return o ? o->data : NULL; }
void mock_free(jobject o) {
  if (!o) return;
  if (o->owns) free(o->data);
  free(o);
}
