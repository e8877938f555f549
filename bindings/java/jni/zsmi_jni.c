/* UNVERIFIED (no JDK / jni.h in the build image).  JNI shim for bindings/java/.../ZstdDecompressor.java over libzsmi.so.
 * gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I include zsmi_jni.c -L zstandard_amd/lib -lzsmi -o libzsmi_jni.so */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include "zsmi.h"

/* The GPU call (context work, launches, a stream synchronise) runs on native buffers: JNI forbids blocking or long calls between
 * Get/ReleasePrimitiveArrayCritical (garbage collection stalls for every Java thread, some VMs deadlock), so the arrays are copied
 * with Get/SetByteArrayRegion, which hold nothing across the call. */
JNIEXPORT jlong JNICALL Java_com_epam_deltix_zstd_ZstdDecompressor_nDecompress(JNIEnv *env, jclass cls, jbyteArray in, jint inOff, jint inLen,
                                                                               jbyteArray out, jint outOff, jint maxLen)
{
    (void)cls;
    jbyte *src = (jbyte *)malloc(inLen > 0 ? (size_t)inLen : 1), *dst = (jbyte *)malloc(maxLen > 0 ? (size_t)maxLen : 1);
    jlong res;
    if (!src || !dst) { free(src); free(dst); return -(jlong)ZSMI_error_memory_allocation; }
    (*env)->GetByteArrayRegion(env, in, inOff, inLen, src);
    if ((*env)->ExceptionCheck(env)) { free(src); free(dst); return -(jlong)ZSMI_error_GENERIC; }
    {
        /* the Java decoder takes exactly one frame per call and needs no trailing bytes (ZstdFrameDecompressor.java:157-225) */
        size_t const r = zsmi_decompress(dst, (size_t)maxLen, src, (size_t)inLen);
        if (zsmi_isError(r)) res = -(jlong)zsmi_getErrorCode(r);
        else { (*env)->SetByteArrayRegion(env, out, outOff, (jsize)r, dst); res = (jlong)r; }
    }
    free(src); free(dst);
    return res;
}

JNIEXPORT jlong JNICALL Java_com_epam_deltix_zstd_ZstdDecompressor_nGetDecompressedSize(JNIEnv *env, jclass cls, jbyteArray in, jint off, jint len)
{
    (void)cls;
    /* the frame header is at most 18 bytes (magic 4, descriptor 1, window 1, dictionary id 4, content size 8): a copy on the stack */
    jbyte hdr[18];
    jint const n = len < 18 ? len : 18;
    if (n > 0) (*env)->GetByteArrayRegion(env, in, off, n, hdr);
    if ((*env)->ExceptionCheck(env)) return -1;
    {
        unsigned long long const r = zsmi_getDecompressedSize(hdr, (size_t)(n > 0 ? n : 0));
        /* zsmi (as the C# reference, ZStdDecompress.cs:621) says 0 for "unknown"; the Java reference says -1 (ZstdFrameDecompressor.java:922).
         * A frame whose header states content size 0 also reads as 0: tell them apart by the frame header descriptor: single-segment or
         * FCS-field flag set (bits 5-7 of byte 4) means the size is stated. */
        if (r == 0 && !(n >= 5 && (((uint8_t)hdr[4] >> 5) != 0))) return -1;
        return (jlong)r;
    }
}

JNIEXPORT jstring JNICALL Java_com_epam_deltix_zstd_ZstdDecompressor_nErrorName(JNIEnv *env, jclass cls, jlong code)
{
    (void)cls;
    return (*env)->NewStringUTF(env, zsmi_getErrorName((size_t)0 - (size_t)code));
}
