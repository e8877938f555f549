#!/usr/bin/env python3
"""Generate frames with upstream libzstd (1.4.8 in this image) that exercise the
format constructs the reference's own two test vectors never reach:
Huffman literals (1 and 4 streams, direct and FSE-compressed weights), RLE literals,
RLE / repeat / predefined sequence modes, raw and RLE blocks, multi-block frames,
concatenated and skippable frames, frame checksum, empty frames, frames without a
content size.

Output: libzstd_fixtures.npz  (frame_<name> = compressed bytes, data_<name> = content)
Inputs are deterministic (seeded numpy / closed forms), so the file is reproducible.
libzstd is an independent implementation of the same format, not the reference.
"""
import ctypes, os, struct
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
Z = ctypes.CDLL("libzstd.so.1")
Z.ZSTD_compressBound.restype = ctypes.c_size_t
Z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
Z.ZSTD_createCCtx.restype = ctypes.c_void_p
Z.ZSTD_freeCCtx.argtypes = [ctypes.c_void_p]
Z.ZSTD_CCtx_setParameter.restype = ctypes.c_size_t
Z.ZSTD_CCtx_setParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
Z.ZSTD_compress2.restype = ctypes.c_size_t
Z.ZSTD_compress2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
Z.ZSTD_isError.argtypes = [ctypes.c_size_t]
Z.ZSTD_decompress.restype = ctypes.c_size_t
Z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
Z.ZSTD_compressStream2.restype = ctypes.c_size_t

ZSTD_c_compressionLevel, ZSTD_c_windowLog = 100, 101
ZSTD_c_contentSizeFlag, ZSTD_c_checksumFlag = 200, 201


def zcompress(data: bytes, level=3, checksum=0, content_size=1, window_log=0) -> bytes:
    c = Z.ZSTD_createCCtx()
    Z.ZSTD_CCtx_setParameter(c, ZSTD_c_compressionLevel, level)
    Z.ZSTD_CCtx_setParameter(c, ZSTD_c_checksumFlag, checksum)
    Z.ZSTD_CCtx_setParameter(c, ZSTD_c_contentSizeFlag, content_size)
    if window_log:
        Z.ZSTD_CCtx_setParameter(c, ZSTD_c_windowLog, window_log)
    cap = Z.ZSTD_compressBound(len(data))
    out = ctypes.create_string_buffer(cap)
    r = Z.ZSTD_compress2(c, out, cap, data, len(data))
    assert not Z.ZSTD_isError(r), r
    Z.ZSTD_freeCCtx(c)
    return out.raw[:r]


def words_text(n, seed, vocab=2000, zipf=1.2):
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", dtype=np.uint8)
    vocab_words = [bytes(rng.choice(letters, size=int(rng.integers(2, 10)))) for _ in range(vocab)]
    out = bytearray()
    ranks = rng.zipf(zipf, size=n // 3 + 16)
    for r in ranks:
        out += vocab_words[(int(r) - 1) % vocab] + (b" " if rng.random() < 0.9 else b".\n")
        if len(out) >= n:
            break
    return bytes(out[:n])


def build_cases():
    rng = np.random.default_rng(1234)
    cases = {}
    t64 = words_text(65536, 1)
    cases["text64k_l3"] = (t64, dict(level=3))
    cases["text64k_l1"] = (t64, dict(level=1))
    cases["text128k_l1_ck"] = (words_text(131072, 2), dict(level=1, checksum=1))
    cases["text300k_l19"] = (words_text(300000, 3, vocab=300), dict(level=19))      # multi-block, repeat modes
    cases["small_text_l3"] = (words_text(700, 4, vocab=40), dict(level=3))          # 1-stream Huffman
    cases["tiny_l3"] = (b"hello hello hello hello world", dict(level=3))
    cases["empty"] = (b"", dict(level=3))
    cases["one_byte"] = (b"Z", dict(level=3, checksum=1))
    cases["zeros_1m"] = (bytes(1 << 20), dict(level=3))                             # RLE literals, RLE blocks
    cases["zeros_64k"] = (bytes(65536), dict(level=3))
    cases["random_200k"] = (rng.integers(0, 256, 200000, dtype=np.uint8).tobytes(), dict(level=3))   # raw blocks
    # few distinct bytes, random order: Huffman with tiny alphabet (direct weights), few matches
    cases["nibbles_40k"] = (rng.integers(0, 4, 40000, dtype=np.uint8).tobytes(), dict(level=3))
    cases["skewed_bytes_100k"] = (np.minimum(rng.geometric(0.08, 100000), 255).astype(np.uint8).tobytes(), dict(level=5))
    # periodic data: every sequence has the same codes -> RLE sequence tables
    cases["period_7_50k"] = ((b"abcdefg" * 8000)[:50000], dict(level=3))
    pat = bytearray()
    for i in range(3000):
        pat += b"0123456789ABCDEF" + bytes([65 + i % 26])
    cases["records_51k"] = (bytes(pat), dict(level=3))
    cases["nocontentsize"] = (words_text(50000, 5), dict(level=3, content_size=0))   # window descriptor present
    cases["bigwindow_l3"] = (words_text(200000, 6, vocab=5000), dict(level=3, checksum=1))
    # long literal runs + long matches (length codes with many extra bits)
    blk = rng.integers(0, 256, 20000, dtype=np.uint8).tobytes()
    cases["longmatch_60k"] = (blk + blk + blk, dict(level=3))
    # > 0x7F00 sequences in a block needs >= 32512 sequences in 128 KiB: alternate 2-byte literal / 3-4 byte match
    a = bytearray()
    base = rng.integers(0, 256, 4096, dtype=np.uint8).tobytes()
    a += base
    while len(a) < 131072:
        o = int(rng.integers(0, 4090))
        a += base[o:o + 4] + bytes([int(rng.integers(0, 256))])
    cases["manyseq_128k_l19"] = (bytes(a[:131072]), dict(level=19))
    # no sequences at all: Huffman-only block (nbSeq == 0)
    cases["noseq_30k_l1"] = (rng.integers(0, 16, 30000, dtype=np.uint8).tobytes(), dict(level=1))
    # multi-block, extremely regular: RLE sequence tables for LL / OF / ML, treeless (repeat) literals
    rec = bytearray()
    for i in range(30000):
        rec += b"0123456789ABCDEF" + bytes([65 + i % 26])
    cases["records_510k_l3"] = (bytes(rec), dict(level=3))
    # multi-block small vocabulary at level 19: repeat mode for a sequence table
    cases["text600k_v50_l19"] = (words_text(600000, 9, vocab=50), dict(level=19))
    return cases


def main():
    out = {}
    cases = build_cases()
    for name, (data, kw) in cases.items():
        frame = zcompress(data, **kw)
        out["frame_" + name] = np.frombuffer(frame, dtype=np.uint8)
        out["data_" + name] = np.frombuffer(data, dtype=np.uint8)
    # concatenated frames with a skippable frame in between (DecompressMultiFrame, ZStdDecompress.cs:2111-2160)
    a, b = cases["small_text_l3"][0], cases["tiny_l3"][0]
    skip = struct.pack("<II", 0x184D2A53, 11) + b"skip me pls"
    multi = zcompress(a, level=3) + skip + zcompress(b, level=1, checksum=1) + zcompress(b"", level=1)
    out["frame_multi_skippable"] = np.frombuffer(multi, dtype=np.uint8)
    out["data_multi_skippable"] = np.frombuffer(a + b, dtype=np.uint8)
    for k in list(out):
        if k.startswith("frame_"):
            f, d = out[k].tobytes(), out["data_" + k[6:]].tobytes()
            buf = ctypes.create_string_buffer(max(len(d), 1))
            r = Z.ZSTD_decompress(buf, len(d), f, len(f))
            assert r == len(d) and buf.raw[:r] == d, k
            print(f"{k[6:]:24s} {len(d):8d} -> {len(f):7d}")
    np.savez_compressed(os.path.join(HERE, "libzstd_fixtures.npz"), **out)


if __name__ == "__main__":
    main()
