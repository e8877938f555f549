#!/usr/bin/env python3
"""Extract the reference's own golden vectors into byte fixtures.

Run in the build container (needs /root/reference); the resulting *.zst / *.bin
files are committed so nothing reads /root/reference at test time.

  csharp_alphabet.zst  <- byte array literal in csharp/test/TestDecompress.cs:58-90
  csharp_alphabet.bin  <- expected output, regenerated the way AlphabetDataPrepare
                          (TestDecompress.cs:28-51) does, with xorshift128+ as in
                          csharp/test/XorShift128Plus.cs:45-53, seed (42, 24)
  java_a2z.zst         <- byte array literal in java/src/test/java/com/epam/deltix/zstd/TestDecompress.java:8-10
  java_a2z.bin         <- 100000 bytes of repeating a..z (the frame's content size
                          and XXH64 pin it; the Java test asserts only "no throw")
"""
import os, re, hashlib, sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
M64 = (1 << 64) - 1


def xorshift128plus(s0, s1):
    s = [s0, s1]
    def nxt():
        x, y = s[0], s[1]
        s[0] = y
        x ^= (x << 23) & M64
        s[1] = x ^ y ^ (x >> 17) ^ (y >> 26)
        return (s[1] + y) & M64
    return nxt


def alphabet_data():
    alpha = "abcdefghijklmnopqrstuvwxyz"
    n = len(alpha)
    alpha2 = alpha + alpha
    rnd = xorshift128plus(42, 24)
    out = []
    for _ in range(256):
        i = rnd() % n
        l = rnd() % n
        out.append(alpha2[i:i + l])
    return "".join(out).encode("ascii")


def main():
    cs = open(os.path.join(REF, "csharp/test/TestDecompress.cs")).read()
    body = cs[cs.index("byte[] compressedData = {"):]
    body = body[:body.index("};")]
    frame = bytes(int(h, 16) for h in re.findall(r"0x([0-9A-Fa-f]{2})", body))
    assert len(frame) == 484
    assert hashlib.sha256(frame).hexdigest().startswith("de152098")
    open(os.path.join(HERE, "csharp_alphabet.zst"), "wb").write(frame)
    data = alphabet_data()
    assert len(data) == 3409 and hashlib.sha256(data).hexdigest().startswith("2de908e2"), len(data)
    open(os.path.join(HERE, "csharp_alphabet.bin"), "wb").write(data)

    jv = open(os.path.join(REF, "java/src/test/java/com/epam/deltix/zstd/TestDecompress.java")).read()
    body = jv[jv.index("compressedData = {") + len("compressedData = {"):]
    body = body[:body.index("};")]
    frame = bytes(int(v) & 0xFF for v in re.findall(r"-?\d+", body))
    assert len(frame) == 51 and hashlib.sha256(frame).hexdigest().startswith("498a4593")
    open(os.path.join(HERE, "java_a2z.zst"), "wb").write(frame)
    data = (b"abcdefghijklmnopqrstuvwxyz" * 3847)[:100000]
    assert hashlib.sha256(data).hexdigest().startswith("bc634ceb")
    open(os.path.join(HERE, "java_a2z.bin"), "wb").write(data)
    print("ok")


if __name__ == "__main__":
    main()
