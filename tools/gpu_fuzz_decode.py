#!/usr/bin/env python3
"""Development aid (GPU box): randomised decoder run.  Frames made by upstream libzstd (levels 1 .. 19, the frame shapes this
codec's own encoder never emits: long matches, overlapping copies, repeat tables, several blocks, checksums off) and by this
codec's encoder are decoded by the HIP kernels (fast path + general kernel) and must give back the input; the same frames
with one byte flipped must fail or decode exactly as under oracle D."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data as D, _oracle as O
from zstandard_amd import BatchCodec

def pieces(rng, total):
    log = D.zipf_log(total)
    yield log
    yield np.tile(np.frombuffer(D.alphabet_data(), dtype=np.uint8), total // 3409 + 1)[:total]
    yield np.repeat(rng.integers(0, 256, total // 300 + 1, dtype=np.uint8), rng.integers(1, 600, total // 300 + 1))[:total]
    yield rng.choice(np.array([97, 98, 99, 32, 10], dtype=np.uint8), total, p=[0.5, 0.2, 0.1, 0.15, 0.05])
    per = rng.integers(0, 256, 37, dtype=np.uint8); yield np.tile(per, total // 37 + 1)[:total]          # short period: overlapping copies
    yield rng.integers(0, 256, total, dtype=np.uint8)
    # 256 symbols with a long tail of rare ones: Huffman tables with many 10- and 11-bit codes (the fast path's two-level table
    # and its fall-back when the long codes need more sub-tables than it holds)
    pz = 1.0 / np.arange(1, 257) ** 1.1; yield rng.choice(256, total, p=pz / pz.sum()).astype(np.uint8)
    pg = 0.97 ** np.arange(256); yield rng.choice(256, total, p=pg / pg.sum()).astype(np.uint8)

def dictionary_rounds(bc, rounds):
    """frames compressed by libzstd WITH a dictionary (raw content of several sizes; a trained one): the batch call with that
    dictionary must restore every chunk"""
    import ctypes
    Z = ctypes.CDLL("libzstd.so.1")
    sz, vp, cp = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_char_p
    Z.ZSTD_compressBound.restype = sz; Z.ZSTD_compressBound.argtypes = [sz]
    Z.ZSTD_createCCtx.restype = vp
    Z.ZSTD_compress_usingDict.restype = sz; Z.ZSTD_compress_usingDict.argtypes = [vp, vp, sz, cp, sz, cp, sz, ctypes.c_int]
    Z.ZSTD_isError.restype = ctypes.c_uint; Z.ZSTD_isError.argtypes = [sz]
    Z.ZDICT_trainFromBuffer.restype = sz; Z.ZDICT_trainFromBuffer.argtypes = [vp, sz, cp, ctypes.POINTER(sz), ctypes.c_uint]
    Z.ZDICT_isError.restype = ctypes.c_uint; Z.ZDICT_isError.argtypes = [sz]
    cctx = Z.ZSTD_createCCtx()
    def comp(data, dic, level):
        cap = Z.ZSTD_compressBound(len(data)); out = ctypes.create_string_buffer(cap)
        r = Z.ZSTD_compress_usingDict(cctx, out, cap, data, len(data), dic, len(dic), level); assert not Z.ZSTD_isError(r)
        return out.raw[:r]
    bad = 0
    for rd in range(rounds):
        rng = np.random.default_rng(500 + rd)
        srcs = list(pieces(rng, 1 << 20))
        base = srcs[rd % 5].tobytes()
        if rd % 2 == 0:
            dic = base[: int(rng.choice([17, 300, 5000, 100000]))]                      # raw content
        else:
            samples = [base[i * 700:i * 700 + 900] for i in range(1000)]
            buf = b"".join(samples); sizes = (sz * len(samples))(*[len(x) for x in samples])
            dbuf = ctypes.create_string_buffer(16384)
            r = Z.ZDICT_trainFromBuffer(dbuf, 16384, buf, sizes, len(samples))
            dic = dbuf.raw[:r] if not Z.ZDICT_isError(r) else base[:4000]
        chunks, frames = [], []
        for i in range(int(rng.integers(40, 120))):
            n = int(rng.choice([rng.integers(1, 600), rng.integers(600, 20000), rng.integers(20000, 300000)]))
            a = int(rng.integers(0, len(base) - n))
            c = base[a:a + n]
            chunks.append(c); frames.append(comp(c, dic, int(rng.integers(1, 20))))
        sizes = np.array([len(c) for c in chunks], dtype=np.uint32)
        fsz = np.array([len(f) for f in frames], dtype=np.uint32)
        blob = np.frombuffer(b"".join(frames), dtype=np.uint8)
        fo = np.zeros(len(frames), dtype=np.uint64); fo[1:] = np.cumsum(fsz.astype(np.uint64))[:-1]
        out, oo, osz = bc.decompress_host(blob, fo, fsz, sizes, dic)
        for i, c in enumerate(chunks):
            if int(osz[i]) != len(c) or out[int(oo[i]):int(oo[i]) + len(c)].tobytes() != c:
                bad += 1; print(f"dictionary round {rd} frame {i} ({len(c)} B, dictionary {len(dic)} B): wrong output, size/status {osz[i]:#x}")
            elif i % 9 == 0 and O.decompress_using_dict(frames[i], len(c), dic) != c:
                bad += 1; print(f"dictionary round {rd} frame {i}: oracle D differs")
        print(f"dictionary round {rd}: {len(frames)} frames, dictionary of {len(dic)} B ({'formatted' if dic[:4] == bytes([0x37, 0xA4, 0x30, 0xEC]) else 'raw content'}): ok so far = {bad == 0}", flush=True)
    return bad


_zc = None
def zstd_compress_checked(data, level):
    """libzstd frame WITH a content checksum (ZSTD_c_checksumFlag): the fast path's k_dec_checksum, or the general kernel's :2076"""
    global _zc
    import ctypes
    if _zc is None:
        Z = ctypes.CDLL("libzstd.so.1"); sz, vp = ctypes.c_size_t, ctypes.c_void_p
        Z.ZSTD_createCCtx.restype = vp
        Z.ZSTD_CCtx_setParameter.restype = sz; Z.ZSTD_CCtx_setParameter.argtypes = [vp, ctypes.c_int, ctypes.c_int]
        Z.ZSTD_compress2.restype = sz; Z.ZSTD_compress2.argtypes = [vp, vp, sz, ctypes.c_char_p, sz]
        Z.ZSTD_compressBound.restype = sz; Z.ZSTD_compressBound.argtypes = [sz]
        Z.ZSTD_isError.restype = ctypes.c_uint; Z.ZSTD_isError.argtypes = [sz]
        _zc = (Z, Z.ZSTD_createCCtx())
    Z, cctx = _zc
    assert not Z.ZSTD_isError(Z.ZSTD_CCtx_setParameter(cctx, 100, level))      # ZSTD_c_compressionLevel
    assert not Z.ZSTD_isError(Z.ZSTD_CCtx_setParameter(cctx, 201, 1))          # ZSTD_c_checksumFlag
    cap = Z.ZSTD_compressBound(len(data)); out = ctypes.create_string_buffer(cap)
    r = Z.ZSTD_compress2(cctx, out, cap, data, len(data)); assert not Z.ZSTD_isError(r)
    return out.raw[:r]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    if not O.libzstd():
        print("libzstd not found: nothing to do"); return 0
    bc = BatchCodec(0); bad = 0
    for rd in range(rounds):
        rng = np.random.default_rng(77 + rd)
        srcs = list(pieces(rng, 1 << 20))
        chunks, frames = [], []
        for i in range(int(rng.integers(60, 200))):
            s = srcs[int(rng.integers(0, len(srcs)))]
            n = int(rng.choice([rng.integers(1, 2000), rng.integers(2000, 70000), rng.integers(70000, 400000)]))
            a = int(rng.integers(0, len(s) - n))
            c = s[a:a + n].tobytes()
            f = zstd_compress_checked(c, int(rng.integers(1, 20))) if i % 3 == 0 else O.zstd_compress(c, int(rng.integers(1, 20)))
            chunks.append(c); frames.append(f)
        sizes = np.array([len(c) for c in chunks], dtype=np.uint32)
        fsz = np.array([len(f) for f in frames], dtype=np.uint32)
        blob = np.frombuffer(b"".join(frames), dtype=np.uint8)
        fo = np.zeros(len(frames), dtype=np.uint64); fo[1:] = np.cumsum(fsz.astype(np.uint64))[:-1]
        out, oo, osz = bc.decompress_host(blob, fo, fsz, sizes)
        for i, c in enumerate(chunks):
            if int(osz[i]) != len(c) or out[int(oo[i]):int(oo[i]) + len(c)].tobytes() != c:
                bad += 1; print(f"round {rd} frame {i} ({len(c)} B): wrong output, size/status {osz[i]:#x}")
        # damaged copies: same verdict and bytes as oracle D
        dmg = []
        for i, f in enumerate(frames[:40]):
            b = bytearray(f); p = int(rng.integers(0, len(b))); b[p] ^= 1 << int(rng.integers(0, 8)); dmg.append(bytes(b))
        dsz = np.array([len(f) for f in dmg], dtype=np.uint32)
        dblob = np.frombuffer(b"".join(dmg), dtype=np.uint8)
        dfo = np.zeros(len(dmg), dtype=np.uint64); dfo[1:] = np.cumsum(dsz.astype(np.uint64))[:-1]
        caps = sizes[:len(dmg)]
        out2, oo2, osz2 = bc.decompress_host(dblob, dfo, dsz, caps)
        for i, f in enumerate(dmg):
            try:
                want = O.decompress(f, int(caps[i])); werr = None
            except O.OracleError as e:
                want = None; werr = e.code
            got = int(osz2[i])
            if want is None:
                if got <= 0xFFFFFF88:
                    bad += 1; print(f"round {rd} damaged {i}: oracle error {werr}, HIP {got:#x}")
            elif got != len(want) or out2[int(oo2[i]):int(oo2[i]) + got].tobytes() != want:
                bad += 1; print(f"round {rd} damaged {i}: oracle decodes {len(want)} B, HIP {got:#x}")
        print(f"round {rd}: {len(frames)} frames, {int(sizes.sum()) >> 10} KiB: ok so far = {bad == 0}", flush=True)
    bad += dictionary_rounds(bc, max(2, rounds // 2))
    print("FUZZ-DECODE", "PASS" if bad == 0 else f"FAIL ({bad})")
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
