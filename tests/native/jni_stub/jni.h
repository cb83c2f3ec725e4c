// jni.h — NOT the JDK's header.  A declaration-only stand-in with the handful of JNI types and JNIEnv members that
// java/jni/dge_jni.cpp uses, so that `g++ -fsyntax-only` can check the shim's C++ against include/dge.h in an image
// without a JDK (tests/test_abi.py).  It proves nothing about a real JVM; a maintainer builds the shim against $JAVA_HOME.
#pragma once
#include <stdint.h>
typedef int32_t jint; typedef int64_t jlong; typedef double jdouble; typedef float jfloat; typedef uint8_t jboolean; typedef jint jsize;
class _jobject {}; typedef _jobject* jobject; typedef jobject jclass; typedef jobject jstring; typedef jobject jarray;
typedef jarray jintArray; typedef jarray jlongArray; typedef jarray jdoubleArray; typedef jarray jfloatArray; typedef jarray jobjectArray;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
struct JNIEnv {
    jclass FindClass(const char*);
    jint ThrowNew(jclass, const char*);
    jsize GetArrayLength(jarray);
    jint* GetIntArrayElements(jintArray, jboolean*);       void ReleaseIntArrayElements(jintArray, jint*, jint);
    jlong* GetLongArrayElements(jlongArray, jboolean*);    void ReleaseLongArrayElements(jlongArray, jlong*, jint);
    jdouble* GetDoubleArrayElements(jdoubleArray, jboolean*); void ReleaseDoubleArrayElements(jdoubleArray, jdouble*, jint);
    jdoubleArray NewDoubleArray(jsize);  void SetDoubleArrayRegion(jdoubleArray, jsize, jsize, const jdouble*);
    jfloatArray NewFloatArray(jsize);    void SetFloatArrayRegion(jfloatArray, jsize, jsize, const jfloat*);
    void SetIntArrayRegion(jintArray, jsize, jsize, const jint*);
    jobject GetObjectArrayElement(jobjectArray, jsize);
    const char* GetStringUTFChars(jstring, jboolean*);     void ReleaseStringUTFChars(jstring, const char*);
};
