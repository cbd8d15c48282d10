/* A small MOCK of <jni.h> for an image without a JDK: the JNI types and the JNIEnv functions
 * frankenpaxos_amd/jni/fpx_jni.c uses, with their real signatures (JNI specification, chapter 4), backed by
 * tests/jni_stub/mock_jvm.c -- "Java arrays" are heap blocks with a length, direct buffers an address with a
 * capacity.  Enough to COMPILE AND RUN the shim from tests/test_jni_shim.py; not a JVM. */
#ifndef FPX_TEST_JNI_STUB_H
#define FPX_TEST_JNI_STUB_H
#include <stdint.h>
typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef jint jsize;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jbyteArray;
#define JNIEXPORT
#define JNICALL
#define JNI_ABORT 2
struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ {
  jsize (*GetArrayLength)(JNIEnv* env, jarray array);
  void (*GetIntArrayRegion)(JNIEnv* env, jintArray array, jsize start, jsize len, jint* buf);
  void (*SetIntArrayRegion)(JNIEnv* env, jintArray array, jsize start, jsize len, const jint* buf);
  void (*GetLongArrayRegion)(JNIEnv* env, jlongArray array, jsize start, jsize len, jlong* buf);
  void (*SetLongArrayRegion)(JNIEnv* env, jlongArray array, jsize start, jsize len, const jlong* buf);
  void (*GetByteArrayRegion)(JNIEnv* env, jbyteArray array, jsize start, jsize len, jbyte* buf);
  void (*SetByteArrayRegion)(JNIEnv* env, jbyteArray array, jsize start, jsize len, const jbyte* buf);
  jobject (*NewDirectByteBuffer)(JNIEnv* env, void* address, jlong capacity);
  void* (*GetDirectBufferAddress)(JNIEnv* env, jobject buf);
  jlong (*GetDirectBufferCapacity)(JNIEnv* env, jobject buf);
};
#endif
