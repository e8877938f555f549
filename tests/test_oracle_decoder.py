"""Oracle D (oracle/zso_decoder.c = C restatement of the reference decoder) against
 (1) the reference's own two golden vectors (csharp/test/TestDecompress.cs:58-90,
     java/src/test/java/com/epam/deltix/zstd/TestDecompress.java:8-10), and
 (2) frames made by upstream libzstd 1.4.8 (tests/golden/gen_fixtures.py) for the constructs
     the reference's tests never reach.  CPU only."""
import ctypes, hashlib, os, struct
import numpy as np
import pytest
import _oracle as O
import _data as D

G = D.GOLDEN


def test_reference_csharp_vector():
    frame = open(os.path.join(G, "csharp_alphabet.zst"), "rb").read()
    want = open(os.path.join(G, "csharp_alphabet.bin"), "rb").read()
    assert hashlib.sha256(frame).hexdigest() == "de1520980bf1932a100d2a0d9b4fb32689b4b9ef0cad677aa3f86aa04bac4a1e"
    assert hashlib.sha256(want).hexdigest() == "2de908e221d2bbee8d7a1f2ce63c443f5ca05a649b92dd083fd9f42a513fab2c"
    L = O.lib()
    assert L.zso_getDecompressedSize(frame, len(frame)) == 3409          # TestDecompress.cs:92-93
    got, st = O.decode_stats(frame, 3409)
    assert got == want                                                    # TestDecompress.cs:98-99
    assert st[0] == 1 and st[10] == 1 and st[14] == 1 and st[18] == 1 and st[25] == 1   # raw literals, 3 FSE tables, checksum
    assert (L.zso_xxh64(want, len(want), 0) & 0xFFFFFFFF) == 0xE9A0B233


def test_reference_java_vector():
    frame = open(os.path.join(G, "java_a2z.zst"), "rb").read()
    want = open(os.path.join(G, "java_a2z.bin"), "rb").read()
    assert hashlib.sha256(frame).hexdigest() == "498a4593f75a87cd84ad2660826943531dbdd8990fcda6036c14472aed321489"
    L = O.lib()
    assert L.zso_getDecompressedSize(frame, len(frame)) == 100000
    got, st = O.decode_stats(frame, 100000)
    assert got == want
    assert st[8] == 1 and st[12] == 1 and st[16] == 1                     # predefined tables
    assert (L.zso_xxh64(want, len(want), 0) & 0xFFFFFFFF) == 0x5AA433F8


@pytest.mark.parametrize("name", sorted(D.fixtures().keys()))
def test_libzstd_fixture(name):
    frame, want = D.fixtures()[name]
    assert O.decompress(frame, len(want)) == want


def test_fixture_coverage():
    """together the fixtures reach the format constructs listed in SURVEY.md §4 as untested by the reference"""
    tot = np.zeros(32, dtype=np.uint64)
    for frame, want in D.fixtures().values():
        _, st = O.decode_stats(frame, len(want))
        tot += st
    must = {0: "raw literals", 1: "rle literals", 2: "huffman literals", 3: "treeless literals", 4: "1-stream", 5: "4-stream",
            6: "direct weights", 7: "fse weights", 8: "LL predefined", 9: "LL rle", 10: "LL fse", 11: "LL repeat",
            12: "OF predefined", 13: "OF rle", 14: "OF fse", 15: "OF repeat", 16: "ML predefined", 17: "ML rle", 18: "ML fse",
            19: "ML repeat", 20: "raw block", 21: "rle block", 22: "compressed block", 24: "skippable", 25: "checksum",
            26: "nbSeq==0", 27: ">= 0x7F00 sequences", 28: "repcode", 29: "multi-block"}
    missing = [v for k, v in must.items() if tot[k] == 0]
    assert not missing, missing


def test_both_huffman_decoders_are_reached():
    """the reference picks its single- or double-symbol Huffman decoder per literal section (SelectDecoder,
    HufDecompress.cs:1082-1095); oracle D restates both and the dispatch.  The fixtures and this repo's own frames reach both."""
    L = O.lib()
    L.zso_statsGet40.argtypes = [ctypes.c_void_p]
    def x4_and_huff(frames):
        x4 = huf = 0
        for frame, want in frames:
            L.zso_statsReset()
            assert O.decompress(frame, len(want)) == want
            st = np.zeros(40, dtype=np.uint32); L.zso_statsGet40(st.ctypes.data_as(ctypes.c_void_p))
            x4 += int(st[34]); huf += int(st[5])
        return x4, huf
    x4, huf = x4_and_huff(D.fixtures().values())
    assert 0 < x4 < huf, (x4, huf)                       # some 4-stream sections by the X4 decoder, some by the X2 one
    data = D.zipf_log(1 << 20).tobytes()
    own = [(O.compress(data[i:i + 65536], 3), data[i:i + 65536]) for i in range(0, len(data), 65536)]
    x4o, hufo = x4_and_huff(own)
    assert hufo > 0


def test_get_decompressed_size_semantics():
    L = O.lib()
    f = D.fixtures()
    assert L.zso_getDecompressedSize(f["nocontentsize"][0], len(f["nocontentsize"][0])) == 0   # unknown -> 0 (:621)
    assert L.zso_getDecompressedSize(b"\x28\xb5\x2f", 3) == 0                                    # too short -> 0
    assert L.zso_getDecompressedSize(b"abcdefgh", 8) == 0                                        # bad magic -> 0
    skip = struct.pack("<II", 0x184D2A50, 4) + b"abcd"
    assert L.zso_getDecompressedSize(skip, len(skip)) == 0                                       # skippable -> 0 (:523-526)
    assert L.zso_getDecompressedSize(f["empty"][0], len(f["empty"][0])) == 0


def _code(frame, cap):
    with pytest.raises(O.OracleError) as e:
        O.decompress(frame, cap)
    return e.value.code


def test_error_codes():
    f = D.fixtures()
    frame, want = f["text64k_l3"]
    assert _code(frame, len(want) - 1) == 70                    # dstSize_tooSmall
    assert _code(frame[:-5], len(want)) in (72, 20)             # truncated
    assert _code(b"\x00" * 16, 16) == 10                        # prefix_unknown
    assert _code(frame + b"\x01", len(want)) == 72              # trailing garbage -> srcSize_wrong (:2156)
    ck, ckw = f["one_byte"]
    bad = bytearray(ck); bad[-1] ^= 0xFF
    assert _code(bytes(bad), len(ckw)) == 22                    # checksum_wrong
    res = bytearray(frame); res[4] |= 0x08
    assert _code(bytes(res), len(want)) == 14                   # reserved bit (:461)
    cs = open(os.path.join(G, "csharp_alphabet.zst"), "rb").read()
    for pos in (20, 100, 200, 300, 470):
        b = bytearray(cs); b[pos] ^= 0x55
        try:
            out = O.decompress(bytes(b), 3409)
            assert False, "corruption must not pass the checksum"
        except O.OracleError as e:
            assert e.code in (20, 22, 70, 72, 1)


def test_truncations_never_crash():
    frame, want = D.fixtures()["small_text_l3"]
    assert O.decompress(b"", 16) == b""          # empty input: the frame loop never runs, returns 0 (:2111, :2155)
    for cut in range(1, len(frame)):
        try:
            O.decompress(frame[:cut], len(want))
            assert False, cut
        except O.OracleError:
            pass


def test_datagen_anchor():
    """the Zipf log generator matches SURVEY.md §8(d)'s prototype (sha256 prefix of the first 8 MiB)"""
    a = D.zipf_log(8 << 20, single=True)
    assert hashlib.sha256(a.tobytes()).hexdigest().startswith("8ac8791e969ec1be")
