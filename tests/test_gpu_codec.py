"""Parity tests proper: the HIP codec (through the C ABI of libzsmi.so) against the CPU oracle.
 - decode: k_decode_frames vs oracle D on the reference's golden vectors, the libzstd fixtures and damaged frames
 - encode: HIP frames must (a) decode bit-exactly under oracle D (the restated reference decoder),
           (b) be byte-identical to oracle E (the scalar statement of the same algorithm),
           (c) decode under upstream libzstd when present, (d) keep the stated ratio tolerance.
Run with -m gpu on an MI355X."""
import ctypes, os, sys
import numpy as np
import pytest
import _oracle as O
import _data as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

ERR = 0xFFFFFF88


@pytest.fixture(scope="module")
def codec():
    from zstandard_amd import BatchCodec
    bc = BatchCodec()
    yield bc
    bc.close()


def _u8(b):
    return np.frombuffer(b, dtype=np.uint8) if len(b) else np.zeros(1, dtype=np.uint8)


def _compress_many(codec, chunks, level=3):
    src = np.concatenate([_u8(c)[:len(c)] for c in chunks] + [np.zeros(1, np.uint8)])
    sizes = np.array([len(c) for c in chunks], dtype=np.uint32)
    offs = np.zeros(len(chunks), dtype=np.uint64); offs[1:] = np.cumsum(sizes.astype(np.uint64))[:-1]
    arena, do, dsz = codec.compress_host(src, offs, sizes, level)
    out = []
    for i in range(len(chunks)):
        assert dsz[i] < ERR, (i, hex(int(dsz[i])))
        out.append(arena[int(do[i]):int(do[i]) + int(dsz[i])].tobytes())
    return out


def _decompress_many(codec, frames, caps):
    src = np.concatenate([_u8(f)[:len(f)] for f in frames] + [np.zeros(1, np.uint8)])
    sizes = np.array([len(f) for f in frames], dtype=np.uint32)
    offs = np.zeros(len(frames), dtype=np.uint64); offs[1:] = np.cumsum(sizes.astype(np.uint64))[:-1]
    arena, do, dsz = codec.decompress_host(src, offs, sizes, np.array(caps, dtype=np.uint32))
    return [(int(dsz[i]), arena[int(do[i]):int(do[i]) + (int(dsz[i]) if dsz[i] < ERR else 0)].tobytes()) for i in range(len(frames))]


# ------------------------------------------------------------------ decode
def test_decode_reference_golden_vectors(codec):
    """csharp/test/TestDecompress.cs:53-99 and java/.../TestDecompress.java:6-20 through the HIP decoder"""
    from zstandard_amd import ZStdDecompress, ZstdDecompressor
    for n in ("csharp_alphabet", "java_a2z"):
        frame = open(os.path.join(D.GOLDEN, n + ".zst"), "rb").read()
        want = open(os.path.join(D.GOLDEN, n + ".bin"), "rb").read()
        size = ZStdDecompress.GetDecompressedSize(frame)
        assert size == len(want)
        dst = bytearray(size)
        r = ZStdDecompress.Decompress(dst, frame)
        assert r == len(want) and bytes(dst) == want
        out = bytearray(size)
        assert ZstdDecompressor().decompress(frame, 0, len(frame), out, 0, size) == size and bytes(out) == want
        assert ZstdDecompressor.getDecompressedSize(frame, 0, len(frame)) == size


def test_decode_libzstd_fixtures(codec):
    fx = D.fixtures()
    names = sorted(fx)
    res = _decompress_many(codec, [fx[k][0] for k in names], [max(len(fx[k][1]), 1) for k in names])
    for k, (sz, got) in zip(names, res):
        assert sz == len(fx[k][1]), (k, hex(sz))
        assert got == fx[k][1], k


def test_decode_errors_match_oracle(codec):
    """damaged frames: the HIP decoder reports an error wherever oracle D does, with the same code for the
    frame-level checks (magic, reserved bit, checksum, capacity, trailing bytes)"""
    frame, want = D.fixtures()["text64k_l3"]
    ck, ckw = D.fixtures()["one_byte"]
    bad_ck = bytearray(ck); bad_ck[-1] ^= 0xFF
    res_bit = bytearray(frame); res_bit[4] |= 0x08
    cases = [(frame, len(want) - 1, 70), (b"\x00" * 16, 16, 10), (frame + b"\x01", len(want), 72),
             (bytes(bad_ck), 1, 22), (bytes(res_bit), len(want), 14), (frame[:-5], len(want), None), (frame[:200], len(want), None)]
    cs = open(os.path.join(D.GOLDEN, "csharp_alphabet.zst"), "rb").read()
    for pos in (20, 100, 200, 300, 470):
        b = bytearray(cs); b[pos] ^= 0x55
        cases.append((bytes(b), 3409, None))
    res = _decompress_many(codec, [c[0] for c in cases], [max(c[1], 1) for c in cases])
    for (fr, cap, code), (sz, _) in zip(cases, res):
        assert sz > ERR, "must be an error"
        try:
            O.decompress(fr, cap); oracle_code = 0
        except O.OracleError as e:
            oracle_code = e.code
        assert oracle_code != 0
        if code is not None:
            assert (0x100000000 - sz) == code == oracle_code


def test_decode_damaged_own_frames_match_oracle(codec):
    """frames of the batch shape (one frame, one compressed block: the decoder's fast path) with one byte damaged at random
    places: the HIP decoder fails exactly where oracle D fails, and where D still decodes, the bytes agree"""
    data = D.zipf_log(1 << 20, seed_lo=31)
    rng = np.random.default_rng(17)
    frames, caps = [], []
    for i in range(8):
        c = data[i * 32768:(i + 1) * 32768].tobytes()
        f = O.compress(c, 3)
        frames.append(f); caps.append(len(c))                                   # undamaged: must decode
        for _ in range(40):
            b = bytearray(f); pos = int(rng.integers(0, len(f))); b[pos] ^= int(rng.integers(1, 256))
            frames.append(bytes(b)); caps.append(len(c))
    res = _decompress_many(codec, frames, caps)
    nerr = 0
    for fr, cap, (sz, got) in zip(frames, caps, res):
        try:
            want = O.decompress(fr, cap)
        except O.OracleError:
            want = None
        if want is None:
            nerr += 1
            assert sz > ERR
        else:
            assert sz == len(want) and got == want
    # 104 of the 320 damaged frames are rejected by oracle D for these seeds and this round's encoder (the rest decode to other bytes: both
    # decoders agree on them).  The count is a property of the frames (oracle E) and of D's acceptance rules: whoever changes either re-pins it
    # knowingly (round 2 loosened a bound here without saying why; the X4 last-symbol rule, HufDecompress.cs:369-385, had made D accept more).
    assert nerr == 104


def test_decode_truncations(codec):
    frame, want = D.fixtures()["small_text_l3"]
    frames = [frame[:c] for c in range(1, len(frame))]
    res = _decompress_many(codec, frames, [len(want)] * len(frames))
    for c, (sz, _) in enumerate(res, start=1):
        assert sz > ERR, c


# ------------------------------------------------------------------ encode
INPUTS = D.mixed_inputs()


@pytest.mark.parametrize("level", [1, 3])
def test_encode_roundtrip_and_bit_exact_vs_oracle(codec, level):
    names = sorted(INPUTS)
    frames = _compress_many(codec, [INPUTS[k] for k in names], level)
    for k, f in zip(names, frames):
        data = INPUTS[k]
        assert O.lib().zso_getDecompressedSize(f, len(f)) == len(data), k
        assert O.decompress(f, len(data)) == data, k                      # reference decoder semantics accept it, bit-exact
        assert f == O.compress(data, level), k                            # same bytes as the scalar statement
        if O.libzstd():
            assert O.zstd_decompress(f, len(data)) == data, k
    # and through this codec's own decoder
    res = _decompress_many(codec, frames, [max(len(INPUTS[k]), 1) for k in names])
    for k, (sz, got) in zip(names, res):
        assert sz == len(INPUTS[k]) and got == INPUTS[k], k


def test_encode_one_shot_api(codec):
    from zstandard_amd import ZstdCompressor, ZStdDecompress
    data = INPUTS["log_200001"]
    f = ZstdCompressor(3).compress(data)
    assert f == O.compress(data, 3)
    dst = bytearray(len(data))
    assert ZStdDecompress.Decompress(dst, f) == len(data) and bytes(dst) == data
    small = bytearray(10)
    assert ZStdDecompress.IsError(ZStdDecompress.Decompress(small, f))


def test_encode_batch_64k_chunks_log(codec):
    """headline shape: 64 KiB chunks of the Zipf log stream; bit-exact vs oracle E; ratio within 1 % of libzstd L3"""
    data = D.zipf_log(8 << 20)
    cs = 65536
    n = len(data) // cs
    offs = np.arange(n, dtype=np.uint64) * cs
    sizes = np.full(n, cs, dtype=np.uint32)
    arena, do, dsz = codec.compress_host(data, offs, sizes, 3)
    assert (dsz < ERR).all()
    ea, eo, es = O.compress_batch(data, offs, sizes, 3, 8)
    assert (dsz == es).all()
    for i in range(n):
        assert (arena[int(do[i]):int(do[i]) + int(dsz[i])] == ea[int(eo[i]):int(eo[i]) + int(es[i])]).all(), i
    # decode all with this codec and compare with the input
    frames = np.concatenate([arena[int(do[i]):int(do[i]) + int(dsz[i])] for i in range(n)])
    fo = np.zeros(n, dtype=np.uint64); fo[1:] = np.cumsum(dsz.astype(np.uint64))[:-1]
    out, oo, osz = codec.decompress_host(frames, fo, dsz, sizes)
    assert (osz == cs).all()
    assert (out[:n * cs] == data[:n * cs]).all()
    if O.libzstd():
        z = sum(len(O.zstd_compress(data[i * cs:(i + 1) * cs].tobytes(), 3)) for i in range(0, n, 4))
        e = int(dsz[::4].sum())
        assert e <= z * 1.01, (e, z)


@pytest.mark.parametrize("level", [1, 3])
def test_encode_batch_128k_chunks_log(codec, level):
    """BASELINE configs 3 and 5 shape: 128 KiB chunks (one LZ unit of two blocks, the second copying from the first);
    bit-exact vs oracle E; ratio within 1 % of libzstd at the same level and chunk size"""
    data = D.zipf_log(8 << 20, seed_lo=21)
    cs = 131072
    n = len(data) // cs
    offs = np.arange(n, dtype=np.uint64) * cs
    sizes = np.full(n, cs, dtype=np.uint32)
    arena, do, dsz = codec.compress_host(data, offs, sizes, level)
    assert (dsz < ERR).all()
    ea, eo, es = O.compress_batch(data, offs, sizes, level, 8)
    assert (dsz == es).all()
    for i in range(n):
        assert (arena[int(do[i]):int(do[i]) + int(dsz[i])] == ea[int(eo[i]):int(eo[i]) + int(es[i])]).all(), i
    for i in range(0, n, 8):
        f = arena[int(do[i]):int(do[i]) + int(dsz[i])].tobytes()
        out, st = O.decode_stats(f, cs)
        assert out == data[i * cs:(i + 1) * cs].tobytes()
    frames = np.concatenate([arena[int(do[i]):int(do[i]) + int(dsz[i])] for i in range(n)])
    fo = np.zeros(n, dtype=np.uint64); fo[1:] = np.cumsum(dsz.astype(np.uint64))[:-1]
    out, oo, osz = codec.decompress_host(frames, fo, dsz, sizes)
    assert (osz == cs).all() and (out[:n * cs] == data[:n * cs]).all()
    if O.libzstd():
        z = sum(len(O.zstd_compress(data[i * cs:(i + 1) * cs].tobytes(), level)) for i in range(0, n, 4))
        e = int(dsz[::4].sum())
        assert e <= z * 1.01, (e, z)


def test_encode_128k_chunks_and_ragged(codec):
    data = D.zipf_log(3 << 20, seed_lo=77)
    rng = np.random.default_rng(5)
    sizes = np.concatenate([np.full(8, 131072), rng.integers(0, 200000, 12)]).astype(np.uint32)
    offs = np.zeros(len(sizes), dtype=np.uint64); offs[1:] = np.cumsum(sizes.astype(np.uint64))[:-1]
    assert int(offs[-1]) + int(sizes[-1]) <= len(data)
    arena, do, dsz = codec.compress_host(data, offs, sizes, 3)
    for i in range(len(sizes)):
        f = arena[int(do[i]):int(do[i]) + int(dsz[i])].tobytes()
        c = data[int(offs[i]):int(offs[i]) + int(sizes[i])].tobytes()
        assert O.decompress(f, len(c)) == c
        assert f == O.compress(c, 3)


def test_many_tiny_and_odd_chunks(codec):
    """ragged batch: 3000 chunks of 0..3000 bytes, plus sizes around the block / frame-header boundaries"""
    data = D.zipf_log(6 << 20, seed_lo=99)
    rng = np.random.default_rng(11)
    sizes = np.concatenate([rng.integers(0, 3000, 3000), [255, 256, 257, 65535, 65536, 65537, 65791, 65792, 131071, 131072, 131073, 262144, 0, 1, 15, 16, 17]]).astype(np.uint32)
    offs = np.zeros(len(sizes), dtype=np.uint64); offs[1:] = np.cumsum(sizes.astype(np.uint64))[:-1]
    assert int(offs[-1]) + int(sizes[-1]) <= len(data)
    arena, do, dsz = codec.compress_host(data, offs, sizes, 3)
    assert (dsz < ERR).all()
    ea, eo, es = O.compress_batch(data, offs, sizes, 3, 8)
    assert (dsz == es).all()
    bad = [i for i in range(len(sizes)) if not (arena[int(do[i]):int(do[i]) + int(dsz[i])] == ea[int(eo[i]):int(eo[i]) + int(es[i])]).all()]
    assert not bad, bad[:5]
    frames = np.concatenate([arena[int(do[i]):int(do[i]) + int(dsz[i])] for i in range(len(sizes))])
    fo = np.zeros(len(sizes), dtype=np.uint64); fo[1:] = np.cumsum(dsz.astype(np.uint64))[:-1]
    out, oo, osz = codec.decompress_host(frames, fo, dsz, np.maximum(sizes, 1))
    assert (osz == sizes).all()
    for i in range(len(sizes)):
        assert (out[int(oo[i]):int(oo[i]) + int(sizes[i])] == data[int(offs[i]):int(offs[i]) + int(sizes[i])]).all(), i


def test_one_shot_large_frame(codec):
    """one frame of 5 MiB + 123 bytes (81 blocks) through the reference-shaped API; decodes under oracle D and this decoder"""
    from zstandard_amd import ZstdCompressor, ZStdDecompress
    data = D.zipf_log((5 << 20) + 123, seed_lo=5).tobytes()
    f = ZstdCompressor(3).compress(data)
    assert f == O.compress(data, 3)
    assert O.decompress(f, len(data)) == data
    dst = bytearray(len(data))
    assert ZStdDecompress.Decompress(dst, f) == len(data) and bytes(dst) == data
    # incompressible and constant inputs take the raw / RLE block paths
    rnd = np.random.default_rng(2).integers(0, 256, 300000, dtype=np.uint8).tobytes()
    for blob in (rnd, bytes(300000), b"\\xAB" * 70001):
        f = ZstdCompressor(3).compress(blob)
        assert f == O.compress(blob, 3) and O.decompress(f, len(blob)) == blob



def test_pack_frames_device():
    """zsmi_packFramesDevice: frames left at their worst-case offsets by the batch compressor end up back to back, in order
    (ragged sizes, so that every copy alignment occurs).  Device buffers come from torch, which has to be loaded before
    libzsmi.so in its process: the check runs as a script of its own (tools/pack_check.py)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pack_check.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_checksummed_one_block_frames(codec):
    """frames with a content checksum (libzstd ZSTD_c_checksumFlag) of the one-block shape: the decoder's fast path checks the
    XXH64 itself (k_dec_checksum); a wrong checksum is checksum_wrong (22) as in the reference (ZStdDecompress.cs:2076-2083)"""
    if not O.libzstd():
        pytest.skip("libzstd not available")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gpu_fuzz_decode import zstd_compress_checked
    data = D.zipf_log(1 << 20, seed_lo=23)
    chunks = [data[i * 30000:(i + 1) * 30000].tobytes() for i in range(12)]
    frames = [zstd_compress_checked(c, 3 + (i % 5)) for i, c in enumerate(chunks)]
    assert all((f[4] >> 2) & 1 for f in frames)                                     # the checksum flag is set
    bad_sum = [bytes(f[:-1]) + bytes([f[-1] ^ 0x40]) for f in frames[:4]]            # stored checksum damaged
    res = _decompress_many(codec, frames + bad_sum, [len(c) for c in chunks] + [len(c) for c in chunks[:4]])
    for (sz, got), c in zip(res[:len(frames)], chunks):
        assert sz == len(c) and got == c
    for (sz, _), f, c in zip(res[len(frames):], bad_sum, chunks):
        assert sz > ERR and (0x100000000 - sz) == 22
        with pytest.raises(O.OracleError) as e:
            O.decompress(f, len(c))
        assert e.value.code == 22


# ------------------------------------------------------------------ mixed corpus, BASELINE configs, in-flight limits
@pytest.mark.parametrize("cs,level", [(65536, 3), (131072, 1), (131072, 3)])
def test_mixed_corpus_matches_oracle_and_ratio(codec, cs, level):
    """every class of the mixed corpus (tests/_corpus.py): HIP frames byte-identical to oracle E, decode under oracle D,
    and the ratio contract per class: compressed size <= 1.01 x libzstd's at the same level and chunk size"""
    import _corpus as C
    worst = {}
    for name, data in C.corpus(1 << 20).items():
        chunks = [data[i:i + cs] for i in range(0, len(data), cs)]
        frames = _compress_many(codec, chunks, level)
        ea, eo, es = O.compress_batch(np.frombuffer(data, dtype=np.uint8), np.arange(0, len(data), cs, dtype=np.uint64),
                                      np.array([len(c) for c in chunks], dtype=np.uint32), level, 8)
        for i, (f, c) in enumerate(zip(frames, chunks)):
            assert f == ea[int(eo[i]):int(eo[i]) + int(es[i])].tobytes(), (name, i)
        assert O.decompress(frames[0], len(chunks[0])) == chunks[0] and O.decompress(frames[-1], len(chunks[-1])) == chunks[-1]
        got = _decompress_many(codec, frames, [len(c) for c in chunks])
        assert all(g == (len(c), c) for g, c in zip(got, chunks)), name
        if O.libzstd():
            worst[name] = round(sum(len(f) for f in frames) / sum(len(O.zstd_compress(c, level)) for c in chunks), 4)
    assert all(v <= 1.01 for v in worst.values()), worst


def test_one_mib_of_zeros_level3(codec):
    """BASELINE config 1 (1 MiB zero-filled buffer, level 3): through the one-shot call (one frame of 16 RLE blocks of 64 KiB:
    73 bytes; libzstd cuts 128 KiB blocks and needs 50) and as a batch of 16 chunks; both against oracle E / D"""
    from zstandard_amd import ZstdCompressor, ZStdDecompress
    data = bytes(1 << 20)
    f = ZstdCompressor(3).compress(data)
    assert f == O.compress(data, 3) and len(f) == 73
    assert O.decompress(f, len(data)) == data
    out = bytearray(len(data))
    assert ZStdDecompress.Decompress(out, f) == len(data) and bytes(out) == data
    if O.libzstd():
        assert O.zstd_decompress(f, len(data)) == data
    chunks = [data[i:i + 65536] for i in range(0, len(data), 65536)]
    frames = _compress_many(codec, chunks, 3)
    assert all(fr == O.compress(c, 3) for fr, c in zip(frames, chunks)) and len(frames[0]) == 11
    assert all(g == (65536, c) for g, c in zip(_decompress_many(codec, frames, [65536] * 16), chunks))


_CHILD = r'''
import sys, os, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import _oracle as O, _data as D
from zstandard_amd import BatchCodec
ERR = 0xFFFFFF88
bc = BatchCodec()
data = D.zipf_log(24 << 20, seed_lo=4242)
rng = np.random.default_rng(21)
# 150 chunks of 0 .. 3 blocks: the compress sub-batch loop (> 64 blocks in flight) turns over several times and meets chunks of
# every block count at the boundaries; 150 frames, then 700 small frames: the decode launch loop (> 64 items in flight)
sizes = np.concatenate([rng.integers(0, 190000, 150), [65536] * 70, [131072] * 10]).astype(np.uint32)
offs = np.zeros(len(sizes), dtype=np.uint64); offs[1:] = np.cumsum(sizes.astype(np.uint64))[:-1]
assert int(offs[-1]) + int(sizes[-1]) <= len(data)
for level in (3, 1):
    arena, do, dsz = bc.compress_host(data, offs, sizes, level)
    assert (dsz < ERR).all()
    ea, eo, es = O.compress_batch(data, offs, sizes, level, 8)
    assert (dsz == es).all(), np.nonzero(dsz != es)[0][:5]
    for i in range(len(sizes)):
        assert (arena[int(do[i]):int(do[i]) + int(dsz[i])] == ea[int(eo[i]):int(eo[i]) + int(es[i])]).all(), i
frames = np.concatenate([arena[int(do[i]):int(do[i]) + int(dsz[i])] for i in range(len(sizes))])
fo = np.zeros(len(sizes), dtype=np.uint64); fo[1:] = np.cumsum(dsz.astype(np.uint64))[:-1]
out, oo, osz = bc.decompress_host(frames, fo, dsz, np.maximum(sizes, 1))
assert (osz == sizes).all()
for i in range(len(sizes)):
    assert (out[int(oo[i]):int(oo[i]) + int(sizes[i])] == data[int(offs[i]):int(offs[i]) + int(sizes[i])]).all(), i
small = rng.integers(1, 5000, 700).astype(np.uint32)
so = np.zeros(len(small), dtype=np.uint64); so[1:] = np.cumsum(small.astype(np.uint64))[:-1]
a2, d2, s2 = bc.compress_host(data, so, small, 3)
fr = np.concatenate([a2[int(d2[i]):int(d2[i]) + int(s2[i])] for i in range(len(small))])
f2 = np.zeros(len(small), dtype=np.uint64); f2[1:] = np.cumsum(s2.astype(np.uint64))[:-1]
# every 7th frame damaged: its status must be the oracle's error code, its neighbours untouched
frd = fr.copy()
for i in range(0, len(small), 7):
    if s2[i] > 12: frd[int(f2[i]) + int(s2[i]) // 2] ^= 0x5A
out2, o2, z2 = bc.decompress_host(frd, f2, s2, small)
for i in range(len(small)):
    fb = frd[int(f2[i]):int(f2[i]) + int(s2[i])].tobytes()
    try:
        want = O.decompress(fb, int(small[i])); code = 0
    except O.OracleError as e:
        want = None; code = e.code
    if code: assert int(z2[i]) == (1 << 32) - code, (i, hex(int(z2[i])), code)
    else: assert int(z2[i]) == len(want) and out2[int(o2[i]):int(o2[i]) + len(want)].tobytes() == want, i
print("CHILD-OK")
'''


def test_in_flight_limits_cross_both_ways():
    """ZSMI_BLOCKS_IN_FLIGHT=64 / ZSMI_ITEMS_IN_FLIGHT=64 in a child process: the compress sub-batch loop and the decode
    launch loop (zsmi_api.hip) run many turns on a small batch; every item compared with the oracle"""
    import subprocess
    env = dict(os.environ, ZSMI_BLOCKS_IN_FLIGHT="64", ZSMI_ITEMS_IN_FLIGHT="64")
    r = subprocess.run([sys.executable, "-c", _CHILD, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CHILD-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_decode_two_block_frames_fast_path_and_its_fallbacks(codec):
    """frames of two compressed blocks take the decoder's fast kernels (block slot 1); shapes next to them must still decode:
    three blocks, a raw or RLE second block, one-block frames in the same batch, a damaged second block (oracle's error code)"""
    data = D.zipf_log(4 << 20, seed_lo=515)
    rng = np.random.default_rng(8)
    noise = rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()
    chunks = [data[0:131072].tobytes(), data[131072:131072 + 100000].tobytes(), data[300000:300000 + 65537].tobytes(), data[400000:400000 + 65536].tobytes(),
              data[500000:500000 + 196608].tobytes(),                                   # three blocks: general kernel
              data[700000:700000 + 65536].tobytes() + noise[:65536],                    # second block raw
              data[800000:800000 + 65536].tobytes() + bytes(40000),                     # second block RLE
              noise[:65536] + data[900000:900000 + 65536].tobytes(),                    # first block raw
              data[1000000:1000000 + 70000].tobytes(), data[1100000:1100000 + 131071].tobytes()]
    frames = [O.compress(c, 3) for c in chunks]
    bad = bytearray(frames[0]); bad[len(bad) - 200] ^= 0x41                              # inside the second block of a two-block frame
    frames.append(bytes(bad)); chunks.append(chunks[0])
    res = _decompress_many(codec, frames, [len(c) for c in chunks])
    for i, (f, c, (sz, got)) in enumerate(zip(frames, chunks, res)):
        try:
            want = O.decompress(f, len(c))
        except O.OracleError as e:
            assert sz == (1 << 32) - e.code, (i, hex(sz), e.code)
            continue
        assert sz == len(want) and got == want, i
    # libzstd's own two-block frames (tables may repeat in the second block: the general kernel takes those)
    if O.libzstd():
        zf = [O.zstd_compress(c, 3) for c in chunks[:4]]
        for (sz, got), c in zip(_decompress_many(codec, zf, [len(c) for c in chunks[:4]]), chunks[:4]):
            assert sz == len(c) and got == c


def test_kernel_timing_modes(codec):
    """zsmi_enableKernelTiming: 1 = events around every launch, 2 = only around the dominant kernel of each direction
    (what bench.py's timed region carries), 0 = none; the data path is the same in all three"""
    chunks = [D.zipf_log(65536, seed_lo=900 + i).tobytes() for i in range(40)] + [D.zipf_log(131072, seed_lo=990).tobytes()]
    ref = _compress_many(codec, chunks)
    try:
        codec.enable_timing(2)
        fr = _compress_many(codec, chunks)
        t2c = codec.kernel_times()
        back = _decompress_many(codec, fr, [len(c) for c in chunks])
        t2d = codec.kernel_times()
        codec.enable_timing(True)
        fr1 = _compress_many(codec, chunks)
        t1c = codec.kernel_times()
    finally:
        codec.enable_timing(False)
    assert fr == ref and fr1 == ref
    assert [b for _, b in back] == chunks
    assert t2c and all(k.startswith("k_lz_walk") for k in t2c), t2c
    assert set(t2d) == {"k_dec_execute"}, t2d
    assert {"k_lz_candidates", "k_lz_walk", "k_encode_sequences", "k_encode_literals", "k_assemble_frames"} <= set(t1c), t1c
    assert all(s > 0 and n > 0 for s, n in list(t2c.values()) + list(t1c.values()))
    assert codec.kernel_times() == {}


def test_decode_literal_spread_across_windows(codec):
    """the fast execute kernel spreads a block's literals through 32 KiB windows of a bit per output byte: one-block frames whose block is
    up to 128 KiB (libzstd's block size: four windows), with long literal runs, long matches and matches that cross window borders;
    every frame also under oracle D"""
    if not O.libzstd():
        pytest.skip("libzstd not present")
    rng = np.random.default_rng(77)
    log = D.zipf_log(1 << 20, seed_lo=4242).tobytes()
    noise = rng.integers(0, 256, 1 << 17, dtype=np.uint8).tobytes()
    chunks = []
    for n in (131072, 131071, 98304 + 5, 65536 + 32768, 32768 + 1, 32768, 32767, 100000):
        chunks.append(log[:n])                                                      # text: short literal runs everywhere
        chunks.append((noise[:20000] + log[:30000] + noise[20000:33000] + log[:40000] + bytes(30000))[:n])   # long literal runs, long matches, a run of zeros
        chunks.append((log[:1000] * 200)[:n])                                       # period 1000: matches far longer than a window
        chunks.append((noise[:32760] + noise[:32760] + noise[100:40000] + log[:40000])[:n])                  # a match that straddles window borders
    frames = [O.zstd_compress(c, 3) for c in chunks]
    try:
        codec.enable_timing(True)
        res = _decompress_many(codec, frames, [len(c) for c in chunks])
        kt = codec.kernel_times()
    finally:
        codec.enable_timing(False)
    assert "k_dec_execute" in kt
    for i, (f, c, (sz, got)) in enumerate(zip(frames, chunks, res)):
        assert O.decompress(f, len(c)) == c
        assert sz == len(c) and got == c, (i, len(c), sz)


def _long_match_inputs():
    """data whose matches are far longer than a walk range: found piecewise by consecutive walkers, joined by the stitch"""
    rng = np.random.default_rng(5)
    per = rng.integers(0, 256, 1000, dtype=np.uint8).tobytes()
    z = bytearray(300000)
    for i in rng.integers(0, len(z), 100):
        z[int(i)] = 1 + int(rng.integers(0, 255))
    base = rng.integers(32, 127, 40000, dtype=np.uint8).tobytes()
    return {
        "period1000_64k": (per * 66)[:65536], "period1000_128k": (per * 132)[:131072], "period1000_300k": (per * 301)[:300000],
        "zeros_noise_300k": bytes(z), "zeros_noise_64k": bytes(z[:65536]),
        "repeat_40k_x3": base * 3, "repeat_shifted": base[:30000] + b"x" + base[:30000] + b"yz" + base[5:29000],
        "period7_100k": (b"abcdefg" * 15000)[:100000], "period300_edge": (bytes(rng.integers(0, 256, 300, dtype=np.uint8)) * 500)[:131071],
    }


@pytest.mark.gpu
@pytest.mark.parametrize("level", [1, 3, 4])
def test_long_matches_are_joined_as_in_the_oracle(codec, level):
    """period-1000 data, zeros with sparse noise, long repeats: HIP == oracle E byte for byte, frames decode; the joined matches keep the
    sequence count small (a 64 KiB block of period-1000 data is a handful of sequences, not one per walk range)"""
    inputs = _long_match_inputs()
    names = sorted(inputs)
    frames = _compress_many(codec, [inputs[k] for k in names], level)
    for k, f in zip(names, frames):
        data = inputs[k]
        assert f == O.compress(data, level), k
        assert O.decompress(f, len(data)) == data, k
    assert len(frames[names.index("period1000_64k")]) < 1400                   # 1000 literal bytes + one long match (round 2's parse: 1.14 x libzstd here)


@pytest.mark.parametrize("level", [1, 3])
def test_units_of_one_repeated_byte_skip_the_parse(codec, level):
    """k_lz_candidates and k_lz_walk leave a unit of one repeated byte alone (its blocks are RLE blocks whatever the parse says):
    the frames are oracle E's for such units, for units that miss the condition by one byte anywhere, and for their neighbours in the batch"""
    rng = np.random.default_rng(11)
    text = D.zipf_log(70000, seed_lo=21).tobytes()
    chunks = [b"\x00" * 65536, b"\xff" * 65536, b"a" * 131072, b"a" * 65536 + b"b" * 65536, b"q" * 65520, b"q" * 65521, b"z" * 16, b"z" * 15,
              b"\x00" * 65535 + b"\x01", b"\x01" + b"\x00" * 65535, b"\x00" * 32768 + b"\x01" + b"\x00" * 32767, b"\x00" * 131071 + b"\x01",
              b"\x00" * 65536 + b"\x01" + b"\x00" * 65535, text, b"k" * 200000, b"k" * 196608]
    for _ in range(8):                                                            # one differing byte at a random place
        c = bytearray(b"\x07" * 65536); c[int(rng.integers(0, 65536))] = 8; chunks.append(bytes(c))
    frames = _compress_many(codec, chunks, level)
    for i, (c, f) in enumerate(zip(chunks, frames)):
        assert f == O.compress(c, level), i
        assert O.decompress(f, len(c)) == c, i
    assert len(frames[0]) < 16 and len(frames[2]) < 24


def _matchless_inputs():
    """noise with 0 .. 12 planted 24-byte repeats (each leaves ~17 - 20 candidate positions: the rule's 32 / 64 per unit are crossed on the way),
    skewed noise (Huffman pays, no match does), noise in front of text and text in front of noise inside one unit"""
    rng = np.random.default_rng(23)
    text = D.zipf_log(140000, seed_lo=77).tobytes()
    out = {}
    for size in (65536, 131072, 40000, 2047, 2048, 4096):
        for plants in (0, 1, 2, 3, 4, 6, 8, 12):
            c = bytearray(rng.integers(0, 256, size, dtype=np.uint8).tobytes())
            for k in range(plants):
                b = int(rng.integers(size // 2, size - 64)); a = b - int(rng.integers(100, min(2000, size // 2 - 64)))     # (near: a slot of the 2^13-slot tables is overwritten every ~8 KiB of noise)
                c[b:b + 24] = c[a:a + 24]
            out["noise_%d_%d" % (size, plants)] = bytes(c)
    out["skewed_65536"] = np.minimum(rng.geometric(0.05, 65536), 255).astype(np.uint8).tobytes()
    # the literals kernel's short cuts around the Huffman build, on both sides of their tests: almost flat counts (the most frequent value just under /
    # over twice the rarest), noise over 200 values and a few rare ones (no flat counts, but code bits >= literal count), over 128 values (7 bits: pays),
    # and noise behind the last byte of the source buffer's block (literals read in place: no bytes behind the block may be touched)
    for name, top in (("nearly_flat_under", 1.9), ("nearly_flat_over", 2.3)):
        cnt = np.full(256, 200); cnt[7] = int(200 * top); v = np.repeat(np.arange(256, dtype=np.uint8), cnt)
        out[name] = rng.permutation(v)[:65536].tobytes() if len(v) >= 65536 else rng.permutation(np.concatenate([v, rng.integers(0, 256, 65536 - len(v), dtype=np.uint8)])).tobytes()
    a = rng.integers(0, 200, 65536, dtype=np.uint8); a[rng.integers(0, 65536, 40)] = rng.integers(200, 256, 40, dtype=np.uint8); out["noise_200_values"] = a.tobytes()
    out["noise_128_values"] = rng.integers(0, 128, 65536, dtype=np.uint8).tobytes()
    for size in (65, 300, 1023, 4099, 16385, 65533):
        out["noise_tail_%d" % size] = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
    out["noise_then_text"] = rng.integers(0, 256, 65536, dtype=np.uint8).tobytes() + text[:65536]
    out["text_then_noise"] = text[:50000] + rng.integers(0, 256, 81072, dtype=np.uint8).tobytes()
    return out


@pytest.mark.parametrize("level", [1, 3])
def test_matchless_units_skip_the_parse(codec, level):
    """a unit with fewer than n / 2048 candidate positions is not parsed (oracle E: findCandidates / parseBlock; k_lz_candidates counts, k_lz_walk
    leaves the unit): the frames are oracle E's on both sides of the threshold and decode; pure noise becomes raw blocks"""
    inputs = _matchless_inputs()
    names = sorted(inputs)
    frames = _compress_many(codec, [inputs[k] for k in names], level)
    for k, f in zip(names, frames):
        assert f == O.compress(inputs[k], level), k
        assert O.decompress(f, len(inputs[k])) == inputs[k], k
    assert len(frames[names.index("noise_65536_0")]) == 65536 + 4 + 1 + 2 + 3              # header + one raw block
    assert len(frames[names.index("noise_then_text")]) < 65536 + 40000                      # the unit's second block still finds its matches


def _byte_run_inputs():
    """runs of one byte of every length and alignment against the candidate kernel's shapes: 8 (16) positions a hasher lane, 64 a step, 512 (1024)
    a group; runs of the same byte next to each other, runs that meet a unit's end, runs inside text and inside noise"""
    import _corpus as C
    rng = np.random.default_rng(41)
    text = C.CLASSES["json"](1 << 18)
    out = {}
    for size in (65536, 131072, 50000):
        for kind in ("text", "noise"):
            b = bytearray(text[:size]) if kind == "text" else bytearray(rng.integers(0, 256, size, dtype=np.uint8).tobytes())
            pos = int(rng.integers(0, 40))
            while pos < size:
                ln = int(rng.choice([3, 7, 8, 9, 15, 16, 17, 23, 24, 31, 33, 63, 64, 65, 100, 127, 129, 500, 511, 513, 1025, 3000]))
                val = int(rng.choice([0, 0, 32, 255, int(rng.integers(0, 256))]))
                b[pos:pos + ln] = bytes([val]) * min(ln, size - pos)
                pos += ln + int(rng.choice([0, 1, 2, 5, 8, 40, 300]))                 # (0: the next run follows at once, often with another byte)
            out["%s_%d" % (kind, size)] = bytes(b)
    z = bytearray(65536); z[65535] = 1; out["zeros_but_the_last"] = bytes(z)
    z = bytearray(65536); z[0] = 1; out["zeros_but_the_first"] = bytes(z)
    z = bytearray(131072); z[70000] = 9; out["big_zeros_but_one"] = bytes(z)
    out["two_bytes_alternating_runs"] = (b"\x00" * 37 + b"\x01" * 91) * 512
    return out


@pytest.mark.parametrize("level", [1, 3])
def test_runs_of_one_byte_are_parsed_as_in_the_oracle(codec, level):
    """k_lz_candidates keeps the inner positions of a run of one byte away from the run's hash slot (they would meet there in one LDS exchange,
    64 lanes on one address) and gives them the distance 1 the sequential loop of oracle E's findCandidates finds: the frames are oracle E's"""
    inputs = _byte_run_inputs()
    names = sorted(inputs)
    frames = _compress_many(codec, [inputs[k] for k in names], level)
    for k, f in zip(names, frames):
        assert f == O.compress(inputs[k], level), k
        assert O.decompress(f, len(inputs[k])) == inputs[k], k


def test_decode_calls_on_both_sides_of_the_launch_shape_rules(codec):
    """the decoder picks its launch shapes by the call's size (zsmi_api.hip): the entropy stage as one launch or one per kernel by the rounds of workgroups
    either takes, the execute kernel at 8 wavefronts a SIMD where that makes one round of two (7169 .. 8192 items on 256 CUs).  Calls of 7100, 7600, 8192 and
    8300 small frames (and 20480 / 22528: apart / fused) must all restore their chunks; mixed content so every kernel has work."""
    log = D.zipf_log(9000 * 1500 + 4096, seed_lo=91).tobytes()
    rng = np.random.default_rng(77)
    for n in (7100, 7600, 8192, 8300, 20480, 22528):
        sizes = rng.integers(600, 1500, n)
        starts = rng.integers(0, len(log) - 1500, n)
        chunks = [log[int(a):int(a) + int(z)] for a, z in zip(starts, sizes)]
        frames = _compress_many(codec, chunks, 3)
        got = _decompress_many(codec, frames, [len(c) for c in chunks])
        bad = [i for i, ((sz, data), c) in enumerate(zip(got, chunks)) if sz != len(c) or data != c]
        assert not bad, (n, bad[:5])


def test_decode_wide_alphabets_flat_huffman_table(codec):
    """literals over all 256 byte values with a long tail of rare ones: more 9-bit prefixes hold 10 / 11-bit codes than the fast path's two-level
    Huffman table has sub-tables, so k_dec_prep emits the flat 2^11 table and k_dec_huffman's flat class decodes them (before round 3's end such
    frames - ELF sections, binary tables - fell to the general kernel).  Frames of this codec and, where present, of upstream libzstd; one- and
    two-block frames; a table of fewer than 11 bits; against the input and oracle D."""
    rng = np.random.default_rng(17)
    def skewed(n, decay, seed):
        r = np.random.default_rng(seed)
        p = decay ** np.arange(256); p /= p.sum()
        perm = r.permutation(256).astype(np.uint8)
        return perm[r.choice(256, size=n, p=p)].tobytes()
    chunks = [skewed(65536, 0.975, 1), skewed(131072, 0.98, 2), skewed(40000, 0.96, 3), skewed(3000, 0.985, 4), skewed(65536, 0.99, 5), skewed(20000, 0.97, 6)]
    frames = _compress_many(codec, chunks, 3)
    if O.libzstd():
        frames += [O.zstd_compress(c, 3) for c in chunks]
        chunks = chunks + chunks
    got = _decompress_many(codec, frames, [len(c) for c in chunks])
    for i, (g, c, f) in enumerate(zip(got, chunks, frames)):
        assert g == (len(c), c), i
        assert O.decompress(f, len(c)) == c, i


def test_decode_large_frames_of_many_blocks(codec):
    """a call that is mostly large frames (3 - 16 blocks each: chunks of 150 KiB - 1 MiB) reserves that many block slots and decodes them on the
    fast path; frames of more blocks, frames with raw / RLE blocks among the compressed ones and small frames in the same call must come out
    right whichever kernel takes them.  Against the input and oracle D."""
    rng = np.random.default_rng(23)
    text = D.zipf_log(3 << 20, seed_lo=31).tobytes()
    noise = rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()
    chunks = [text[:150000], text[100000:100000 + 262144], text[:1 << 20], text[5000:5000 + 1000000], text[:(1 << 20) + 1],
              text[:200000] + noise + text[200000:400000], text[:65536 * 3] + bytes(65536) + text[:70000], text[:40000], text[:65536 * 5 + 17]]
    frames = _compress_many(codec, chunks, 3)
    got = _decompress_many(codec, frames, [len(c) for c in chunks])
    for i, (g, c, f) in enumerate(zip(got, chunks, frames)):
        assert g == (len(c), c), i
    assert O.decompress(frames[2], len(chunks[2])) == chunks[2] and O.decompress(frames[6], len(chunks[6])) == chunks[6]
    # the same frames one by one and among many small ones (the call then keeps its one or two slots: the general kernel takes the large frames)
    small = [text[i * 30000:(i + 1) * 30000] for i in range(40)]
    frames2 = _compress_many(codec, small + chunks[:3], 3)
    got2 = _decompress_many(codec, frames2, [len(c) for c in small + chunks[:3]])
    assert all(g == (len(c), c) for g, c in zip(got2, small + chunks[:3]))


def test_decode_libzstd_frames_of_many_blocks_with_repeated_tables(codec):
    """upstream libzstd's multi-block frames (blocks of 128 KiB; later blocks repeat the Huffman table - "treeless" literals - and sequence tables,
    ZStdDecompress.cs:696-697, 1062-1064) in a call of mostly large frames: k_dec_prep copies a repeated table into the block's own slot, so the
    frames stay on the fast path; levels 1 - 19, text and binary-ish data, against the input and oracle D."""
    if not O.libzstd():
        pytest.skip("no libzstd here")
    rng = np.random.default_rng(29)
    text = D.zipf_log(2 << 20, seed_lo=41).tobytes()
    skew = np.random.default_rng(3)
    p = 0.97 ** np.arange(256); p /= p.sum()
    binary = skew.permutation(256).astype(np.uint8)[skew.choice(256, size=600000, p=p)].tobytes()
    chunks, frames = [], []
    for i, (data, n, lvl) in enumerate([(text, 300000, 1), (text, 400000, 3), (text, 1 << 20, 3), (text, 700000, 5), (text, 262144, 9), (text, 393216, 19),
                                        (binary, 400000, 3), (binary, 600000, 1), (text[:200000] + binary[:200000] + text[:150000], 550000, 3), (text, 131073, 3)]):
        a = int(rng.integers(0, len(data) - n + 1)); c = data[a:a + n]
        chunks.append(c); frames.append(O.zstd_compress(c, lvl))
    got = _decompress_many(codec, frames, [len(c) for c in chunks])
    for i, (g, c, f) in enumerate(zip(got, chunks, frames)):
        assert g == (len(c), c), i
    assert O.decompress(frames[1], len(chunks[1])) == chunks[1] and O.decompress(frames[6], len(chunks[6])) == chunks[6]


def test_intended_shapes_stay_on_the_decode_fast_path():
    """tools/fastpath_check.py (its own process: the library with the debug hooks): every shape the fast path is meant to take is decoded
    THERE - a silent fall-back to the general kernel decodes correctly, 4 x slower, and no other test would notice."""
    import json, subprocess
    # (the tool builds the debug-hook library itself when it is missing or stale: on a GPU run this test never skips)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fastpath_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert len(res) >= 5, res
    for label, (fast, n) in res.items():
        assert fast >= 0, (label, "wrong output")
        assert fast == n, (label, fast, n)


def test_decode_scratch_follows_the_capacities_and_is_given_back():
    """round 4: the fast path's slots are sized by the call's largest capacity (literals min(cap, 128 KiB), sequences cap / 3) and the general kernel's literal
    buffers belong to a pool of wavefronts: a call with 32 KiB capacities holds far less than 263 KiB an item, a call with generous capacities more, and a
    later small call gives the excess back (zsmi_decodeScratchBytes); the output is right every time"""
    from zstandard_amd import BatchCodec
    bc = BatchCodec()
    try:
        text = D.zipf_log(2048 * 32768, seed_lo=5).tobytes()
        chunks = [text[i * 32768:(i + 1) * 32768] for i in range(2048)]
        frames = _compress_many(bc, chunks, 3)
        def scratch_after(caps):
            out = _decompress_many(bc, frames, caps)
            assert all(sz == 32768 and data == c for (sz, data), c in zip(out, chunks))
            return int(bc.L.zsmi_decodeScratchBytes(bc.ctx))
        small = scratch_after([32768] * 2048)
        pool = 2048 * ((1 << 17) + 64)                                       # the general kernel's pool: at most one buffer a wavefront, never more wavefronts than items
        assert small <= 2048 * 140 * 1024 + pool + (64 << 20), small            # ~125 KiB an item + pool (+ reserve slack); round 3: 263 KiB an item
        big = scratch_after([1 << 20] * 2048)                                # generous capacities: slots for 16 blocks of 128 KiB
        assert big > 4 * small, (small, big)
        again = scratch_after([32768] * 2048)
        assert again < big // 2, (big, again)                                # the excess went back
    finally:
        bc.close()


def test_shutdown_releases_the_one_shot_contexts_and_they_come_back():
    """zsmi_shutdown (for embedders that unload the library: nothing is released from an exit-time destructor) frees the per-device contexts of the
    one-shot calls; the next one-shot call makes a new one"""
    from zstandard_amd import ZStdDecompress, _lib
    frame = open(os.path.join(D.GOLDEN, "csharp_alphabet.zst"), "rb").read()
    want = open(os.path.join(D.GOLDEN, "csharp_alphabet.bin"), "rb").read()
    for _ in range(2):
        dst = bytearray(len(want))
        assert ZStdDecompress.Decompress(dst, frame) == len(want) and bytes(dst) == want
        _lib.lib().zsmi_shutdown()
    dst = bytearray(len(want))
    assert ZStdDecompress.Decompress(dst, frame) == len(want) and bytes(dst) == want
