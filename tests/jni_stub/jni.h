/* Minimal stand-in for <jni.h>, ONLY so that tests/test_jni_shim.py can type-check
 * frankenpaxos_amd/jni/fpx_jni.c against include/fpx.h in an image without a JDK.  It declares the
 * handful of JNI types and the four JNIEnv functions the shim uses, with their real signatures
 * (JNI specification, "Get/ReleasePrimitiveArrayCritical").  Not a JNI implementation. */
#ifndef FPX_TEST_JNI_STUB_H
#define FPX_TEST_JNI_STUB_H
#include <stdint.h>
typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
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
  void* (*GetPrimitiveArrayCritical)(JNIEnv* env, jarray array, jboolean* isCopy);
  void (*ReleasePrimitiveArrayCritical)(JNIEnv* env, jarray array, void* carray, jint mode);
  jobject (*NewDirectByteBuffer)(JNIEnv* env, void* address, jlong capacity);
  void* (*GetDirectBufferAddress)(JNIEnv* env, jobject buf);
};
#endif
