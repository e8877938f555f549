// UNVERIFIED (no JDK in the build image).  Drop-in for the reference's com.epam.deltix.zstd.ZstdDecompressor
// (java/src/main/java/com/epam/deltix/zstd/ZstdDecompressor.java:18-34): same two public methods; the frame is decoded
// on the GPU through the JNI shim bindings/java/jni/zsmi_jni.c over libzsmi.so (include/zsmi.h).
package com.epam.deltix.zstd;

public class ZstdDecompressor {
    static {
        System.loadLibrary("zsmi_jni");
    }

    // result >= 0: bytes written; < 0: -(error code of csharp/src/ZStdErrors.cs:61-90)
    private static native long nDecompress(byte[] input, int inputOffset, int inputLength, byte[] output, int outputOffset, int maxOutputLength);
    private static native long nGetDecompressedSize(byte[] input, int offset, int length);
    private static native String nErrorName(long code);

    // replaces ZstdDecompressor.java:22-29.  The reference throws RuntimeException("<reason>: offset=<n>") (Util.java:32-40)
    public int decompress(final byte[] input, final int inputOffset, final int inputLength,
                          final byte[] output, final int outputOffset, final int maxOutputLength) {
        checkRange(input, inputOffset, inputLength);
        checkRange(output, outputOffset, maxOutputLength);
        final long r = nDecompress(input, inputOffset, inputLength, output, outputOffset, maxOutputLength);
        if (r < 0)
            throw new RuntimeException(nErrorName(-r) + ": offset=" + inputOffset);
        return (int) r;
    }

    // replaces ZstdDecompressor.java:31-33 (ZstdFrameDecompressor.getDecompressedSize :922): -1 when the header holds no content size
    public static long getDecompressedSize(final byte[] input, final int offset, final int length) {
        checkRange(input, offset, length);
        return nGetDecompressedSize(input, offset, length);
    }

    private static void checkRange(final byte[] a, final int off, final int len) {
        if (a == null) throw new NullPointerException();
        if (off < 0 || len < 0 || off > a.length - len) throw new IndexOutOfBoundsException("offset=" + off + " length=" + len + " array=" + a.length);
    }
}
