"""The reference's SECOND decoder (Java, aircompressor lineage) is stricter than the C# one that oracle D restates.  No JVM exists in
this image, so its acceptance rules are stated here and checked on the frames this codec emits (oracle E's frames, which the HIP encoder
reproduces byte for byte: tests/test_gpu_codec.py), by a small frame walker written for this test.

Rules, from java/src/main/java/com/epam/deltix/zstd/ZstdFrameDecompressor.java:
  :843-920  readFrameHeader: no dictionary id (:891); a frame WITH a window descriptor has windowSize = base + base/8*mantissa,
            a single-segment frame has windowSize = -1 (:857, the field is never set from the content size)
  :309      windowSize <= MAX_WINDOW_SIZE = 8 MiB (:38) -- so a single-segment frame of ANY content size passes (-1 <= 8 MiB); only a
            windowed frame can fail.  Every frame of this codec is single-segment.
  :157-225  one frame per call, no skippable frames; the output buffer must be exactly the content (checksum over outputLimit, :216)
  :36,:279  blocks: compressed block size 3 .. 128 KiB
  :62       offset code <= 28; :60-61 literal-length code <= 35, match-length code <= 52; table logs <= 9 / 9 / 8 (:64-66)
  Huffman.java:45  table log <= 12, and only 1- or 4-stream literals with the sizes of the format
"""
import numpy as np
import pytest

import _oracle as O
import _data as D

MAX_WINDOW = 1 << 23
LL_DEFAULT_MAX, ML_DEFAULT_MAX, OF_DEFAULT_MAX = 35, 52, 28


class Bits:
    """forward bit reader over bytes (FSE table descriptions are read forward: EntropyCommon.cs:79-188 / FseTableReader.java:28)"""
    def __init__(self, b, pos):
        self.b, self.pos, self.bit = b, pos, 0

    def peek(self, n):
        v = int.from_bytes(self.b[self.pos:self.pos + 8].ljust(8, b"\0"), "little") >> self.bit
        return v & ((1 << n) - 1)

    def skip(self, n):
        self.bit += n; self.pos += self.bit >> 3; self.bit &= 7

    def end(self):
        return self.pos + (1 if self.bit else 0)


def read_ncount(b, pos, max_symbol):
    """-> (table log, highest symbol with a count, position behind the description)"""
    r = Bits(b, pos)
    table_log = r.peek(4) + 5; r.skip(4)
    remaining, threshold, nbits = (1 << table_log) + 1, 1 << table_log, table_log + 1
    sym, prev0, top = 0, False, 0
    while remaining > 1 and sym <= max_symbol:
        if prev0:
            n0 = sym
            while r.peek(2) == 3:
                n0 += 3; r.skip(2)
            n0 += r.peek(2); r.skip(2)
            assert n0 <= max_symbol + 1
            sym = n0
        mx = 2 * threshold - 1 - remaining
        if r.peek(nbits - 1) < mx:
            count = r.peek(nbits - 1); r.skip(nbits - 1)
        else:
            count = r.peek(nbits)
            if count >= threshold:
                count -= mx
            r.skip(nbits)
        count -= 1
        remaining -= -count if count < 0 else count
        if count != 0:
            top = sym
        sym += 1
        prev0 = count == 0
        while remaining < threshold:
            nbits -= 1; threshold >>= 1
    assert remaining == 1
    return table_log, top, r.end()


def walk_frame(frame, content_size):
    """asserts the Java decoder's rules on one frame; returns (blocks, compressed blocks)"""
    assert frame[:4] == b"\x28\xb5\x2f\xfd"                                   # verifyMagic :928 (no skippable frames)
    fhd = frame[4]
    single, dict_desc, fcs_desc, checksum = (fhd >> 5) & 1, fhd & 3, fhd >> 6, (fhd >> 2) & 1
    assert dict_desc == 0                                                     # :891 "Custom dictionaries not supported"
    assert not (fhd & 0x08)
    pos = 5
    window = -1
    if not single:
        wd = frame[pos]; pos += 1
        base = 1 << (10 + (wd >> 3)); window = base + (base // 8) * (wd & 7)
    assert window <= MAX_WINDOW                                               # :309 (a single-segment frame: -1)
    fcs = {0: 1 if single else 0, 1: 2, 2: 4, 3: 8}[fcs_desc]
    stated = int.from_bytes(frame[pos:pos + fcs], "little") + (256 if fcs_desc == 1 else 0) if fcs else None
    pos += fcs
    assert stated == content_size                                             # the caller sizes the output buffer from it (:216 needs the exact size)
    nblocks = ncomp = 0
    produced = 0
    while True:
        h = int.from_bytes(frame[pos:pos + 3], "little"); pos += 3
        last, btype, bsize = h & 1, (h >> 1) & 3, h >> 3
        nblocks += 1
        assert btype in (0, 1, 2)
        if btype == 0:
            assert bsize <= 128 * 1024; produced += bsize; pos += bsize
        elif btype == 1:
            assert bsize <= 128 * 1024; produced += bsize; pos += 1
        else:
            assert 3 <= bsize <= 128 * 1024                                   # :279-283
            ncomp += 1
            walk_compressed_block(frame, pos, bsize)
            pos += bsize
        if last:
            break
    if checksum:
        pos += 4
    assert pos == len(frame)                                                  # one frame, nothing behind it
    return nblocks, ncomp


def walk_compressed_block(frame, pos, bsize):
    end = pos + bsize
    b0 = frame[pos]
    ltype, sfmt = b0 & 3, (b0 >> 2) & 3
    if ltype in (0, 1):                                                       # raw / RLE literals :692-760
        if sfmt in (0, 2): hs, lsize = 1, b0 >> 3
        elif sfmt == 1: hs, lsize = 2, int.from_bytes(frame[pos:pos + 2], "little") >> 4
        else: hs, lsize = 3, int.from_bytes(frame[pos:pos + 3], "little") >> 4
        assert lsize <= 128 * 1024
        pos += hs + (lsize if ltype == 0 else 1)
    else:
        assert ltype == 2                                                     # 3 = repeat: "Dictionary is corrupted" unless a table is loaded; never emitted
        if sfmt in (0, 1):
            v = int.from_bytes(frame[pos:pos + 3], "little"); hs, lsize, csize = 3, (v >> 4) & 0x3FF, (v >> 14) & 0x3FF
        elif sfmt == 2:
            v = int.from_bytes(frame[pos:pos + 4], "little"); hs, lsize, csize = 4, (v >> 4) & 0x3FFF, v >> 18
        else:
            v = int.from_bytes(frame[pos:pos + 5], "little"); hs, lsize, csize = 5, (v >> 4) & 0x3FFFF, v >> 22
        assert lsize <= 128 * 1024 and csize <= bsize
        if sfmt != 0:
            assert csize >= 10                                                # 4 streams: 6-byte jump table + a byte each (Huffman.java:155)
        # Huffman table description: header byte < 128: FSE-compressed weights, else (byte - 127) weights, 4 bits each
        hb = frame[pos + hs]
        nweights = None
        if hb >= 128:
            nweights = hb - 127
            weights = []
            for i in range(nweights):
                byte = frame[pos + hs + 1 + i // 2]
                weights.append(byte >> 4 if i % 2 == 0 else byte & 15)
            assert max(weights) <= 12
            total = sum((1 << w) >> 1 for w in weights)
            log = total.bit_length()                                          # the last weight completes the power of two above
            assert log <= 12                                                  # Huffman.java:45 / Huf.cs:148
        pos += hs + csize
    assert pos < end
    # sequences header :317-360
    nb = frame[pos]; pos += 1
    if nb == 0:
        assert pos == end
        return
    if nb >= 128:
        if nb == 255: nbseq = int.from_bytes(frame[pos:pos + 2], "little") + 0x7F00; pos += 2
        else: nbseq = ((nb - 128) << 8) + frame[pos]; pos += 1
    else:
        nbseq = nb
    assert nbseq > 0
    modes = frame[pos]; pos += 1
    assert modes & 3 == 0
    for name, mode, dmax, max_log in (("LL", modes >> 6, LL_DEFAULT_MAX, 9), ("OF", (modes >> 4) & 3, OF_DEFAULT_MAX, 8), ("ML", (modes >> 2) & 3, ML_DEFAULT_MAX, 9)):
        assert mode != 3, "repeat mode in a block that has no earlier table"  # this codec never emits repeat mode
        limit = 28 if name == "OF" else dmax
        if mode == 1:
            assert frame[pos] <= limit; pos += 1                              # RLE: the one symbol
        elif mode == 2:
            log, top, pos = read_ncount(frame, pos, dmax if name != "OF" else 31)
            assert log <= max_log and top <= limit
    assert pos < end                                                          # the bitstream has at least a byte


def _check(data, level):
    f = O.compress(data, level)
    assert O.decompress(f, len(data)) == data
    return walk_frame(f, len(data))


@pytest.mark.parametrize("level", [1, 3, 4])
def test_emitted_frames_meet_the_java_decoders_rules(level):
    inputs = D.mixed_inputs()
    seen_comp = 0
    for name, data in inputs.items():
        if len(data) == 0:
            continue                                                          # (an empty frame: the Java decoder returns 0 before reading it when the output is empty, :164)
        nblocks, ncomp = _check(data, level)
        seen_comp += ncomp
    assert seen_comp > 10


def test_frame_above_8_mib_is_single_segment_and_passes():
    """zsmi_compress on a large input emits ONE single-segment frame: the Java decoder's window check (:309) sees windowSize = -1 for it
    (:857), whatever the content size; the C# decoder needs windowLog <= 30 only for windowed frames (ZStdDecompress.cs:468)."""
    data = D.zipf_log((9 << 20) + 12345, seed_lo=3).tobytes()
    f = O.compress(data, 3)
    assert (f[4] >> 5) & 1 == 1
    nblocks, ncomp = walk_frame(f, len(data))
    assert nblocks == (len(data) + 65535) // 65536 and ncomp == nblocks
    assert O.decompress(f, len(data)) == data


def test_walker_rejects_what_the_java_decoder_rejects():
    """the walker is not vacuous: a windowed frame above 8 MiB, a dictionary id, a second frame behind the first all fail"""
    data = D.zipf_log(70000, seed_lo=9).tobytes()
    f = bytearray(O.compress(data, 3))
    walk_frame(bytes(f), len(data))
    with pytest.raises(AssertionError):
        walk_frame(bytes(f) + bytes(f), len(data))
    g = bytearray(f); g[4] |= 1
    with pytest.raises(AssertionError):
        walk_frame(bytes(g), len(data))
    # a windowed header with window 16 MiB: FHD without the single-segment bit, window descriptor exponent 14
    fcs = f[5:9] if (f[4] >> 6) == 2 else None
    assert fcs is not None
    w = bytes(f[:4]) + bytes([f[4] & ~0x20 & 0xFF, 14 << 3]) + bytes(f[5:])
    with pytest.raises(AssertionError):
        walk_frame(w, len(data))
