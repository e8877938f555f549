/* UNVERIFIED (no JDK / jni.h in the build image).  JNI shim for bindings/java/.../ZstdDecompressor.java over libzsmi.so.
 * gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I include zsmi_jni.c -L zstandard_amd/lib -lzsmi -o libzsmi_jni.so */
#include <jni.h>
#include <stdint.h>
#include "zsmi.h"

JNIEXPORT jlong JNICALL Java_com_epam_deltix_zstd_ZstdDecompressor_nDecompress(JNIEnv *env, jclass cls, jbyteArray in, jint inOff, jint inLen,
                                                                               jbyteArray out, jint outOff, jint maxLen)
{
    (void)cls;
    jbyte *pi = (*env)->GetPrimitiveArrayCritical(env, in, NULL);
    if (!pi) return -(jlong)ZSMI_error_memory_allocation;
    jbyte *po = (*env)->GetPrimitiveArrayCritical(env, out, NULL);
    if (!po) { (*env)->ReleasePrimitiveArrayCritical(env, in, pi, JNI_ABORT); return -(jlong)ZSMI_error_memory_allocation; }
    /* the Java decoder takes exactly one frame per call and needs no trailing bytes (ZstdFrameDecompressor.java:157-225) */
    size_t const r = zsmi_decompress(po + outOff, (size_t)maxLen, pi + inOff, (size_t)inLen);
    (*env)->ReleasePrimitiveArrayCritical(env, out, po, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, in, pi, JNI_ABORT);
    return zsmi_isError(r) ? -(jlong)zsmi_getErrorCode(r) : (jlong)r;
}

JNIEXPORT jlong JNICALL Java_com_epam_deltix_zstd_ZstdDecompressor_nGetDecompressedSize(JNIEnv *env, jclass cls, jbyteArray in, jint off, jint len)
{
    (void)cls;
    jbyte *pi = (*env)->GetPrimitiveArrayCritical(env, in, NULL);
    if (!pi) return -1;
    unsigned long long const r = zsmi_getDecompressedSize(pi + off, (size_t)len);
    /* zsmi (as the C# reference, ZStdDecompress.cs:621) says 0 for "unknown"; the Java reference says -1 (ZstdFrameDecompressor.java:922).
     * A frame whose header states content size 0 also reads as 0: tell them apart by the frame header descriptor: single-segment or
     * FCS-field flag set (bits 5-7 of byte 4) means the size is stated. */
    jlong res = (jlong)r;
    if (r == 0 && !(len >= 5 && (((uint8_t)pi[off + 4] >> 5) != 0))) res = -1;
    (*env)->ReleasePrimitiveArrayCritical(env, in, pi, JNI_ABORT);
    return res;
}

JNIEXPORT jstring JNICALL Java_com_epam_deltix_zstd_ZstdDecompressor_nErrorName(JNIEnv *env, jclass cls, jlong code)
{
    (void)cls;
    return (*env)->NewStringUTF(env, zsmi_getErrorName((size_t)0 - (size_t)code));
}
