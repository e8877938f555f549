// UNVERIFIED (no JDK in the build image).  JUnit port of the reference's only Java test
// (java/src/test/java/com/epam/deltix/zstd/TestDecompress.java:6-20): the same 51-byte frame (this repo's tests/golden/java_a2z.zst),
// here WITH a content assertion (the reference only checks "does not throw", which includes its XXH64 check).
package com.epam.deltix.zstd;

import org.junit.Assert;
import org.junit.Test;

public class TestDecompress {
    private static final byte[] FRAME = { (byte) 0x28, (byte) 0xb5, (byte) 0x2f, (byte) 0xfd, (byte) 0xa4, (byte) 0xa0, (byte) 0x86, (byte) 0x01, (byte) 0x00, (byte) 0x1d, (byte) 0x01, (byte) 0x00, (byte) 0xd8, (byte) 0x61, (byte) 0x62, (byte) 0x63, (byte) 0x64, (byte) 0x65, (byte) 0x66, (byte) 0x67, (byte) 0x68, (byte) 0x69, (byte) 0x6a, (byte) 0x6b, (byte) 0x6c, (byte) 0x6d, (byte) 0x6e, (byte) 0x6f, (byte) 0x70, (byte) 0x71, (byte) 0x72, (byte) 0x73, (byte) 0x74, (byte) 0x75, (byte) 0x76, (byte) 0x77, (byte) 0x78, (byte) 0x79, (byte) 0x7a, (byte) 0x61, (byte) 0x01, (byte) 0x00, (byte) 0x0b, (byte) 0x1a, (byte) 0x76, (byte) 0x3e, (byte) 0xc7, (byte) 0xf8, (byte) 0x33, (byte) 0xa4, (byte) 0x5a };

    @Test
    public void decompressGoldenFrame() {
        Assert.assertEquals(100000L, ZstdDecompressor.getDecompressedSize(FRAME, 0, FRAME.length));
        final byte[] out = new byte[100000];
        final int n = new ZstdDecompressor().decompress(FRAME, 0, FRAME.length, out, 0, out.length);
        Assert.assertEquals(100000, n);
        for (int i = 0; i < n; i++)
            Assert.assertEquals((byte) ('a' + i % 26), out[i]);
    }

    @Test(expected = RuntimeException.class)
    public void corruptFrameThrows() {
        final byte[] bad = FRAME.clone();
        bad[0] ^= 0x7f;
        new ZstdDecompressor().decompress(bad, 0, bad.length, new byte[100000], 0, 100000);
    }
}
