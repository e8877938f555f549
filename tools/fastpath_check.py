#!/usr/bin/env python3
"""GPU check (run by tests/test_gpu_codec.py::test_intended_shapes_stay_on_the_decode_fast_path, in a process of its own because it loads
the library built with the debug hooks): frames of the shapes the decoder's fast path is meant to take are decoded, and the per-item
descriptors (ZsFastDesc.fast) are read back.  A shape that silently falls back to the general kernel decodes correctly and 4 x slower -
only this count shows it (ELF-class frames did so for two rounds).  Prints one JSON object: shape -> [items on the fast path, items]."""
import os; os.environ["ZSMI_DEBUG_LIB"] = "1"
import sys, ctypes, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _data as D, _oracle as O, _corpus as C
from zstandard_amd import BatchCodec, _lib


def desc_layout(Z):
    """words of a ZsFastDesc and the word index of `fast`: asked of the library itself (zsmi_dbg_descLayout), so a field added to the descriptor
    cannot make this check read the wrong word"""
    lay = (ctypes.c_uint32 * 6)()
    Z.zsmi_dbg_descLayout(lay)
    return int(lay[0]), int(lay[1])



def main():
    if _lib.built_fingerprint() != _lib.source_fingerprint():
        _lib.build()                                      # (a stale debug build would check yesterday's kernels)
    bc = BatchCodec(0); Z = _lib.lib()
    DESC_WORDS, FAST_AT = desc_layout(Z)
    assert 8 <= DESC_WORDS <= 256 and FAST_AT < DESC_WORDS
    rng = np.random.default_rng(5)
    text = D.zipf_log(6 << 20, seed_lo=41).tobytes()
    noise = rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
    elf = dict(C.corpus(1 << 20)).get("elf", None)
    res = {}

    def run(label, chunks, own=True, lvl=3):
        if own:
            src = np.frombuffer(b"".join(chunks), dtype=np.uint8); sizes = np.array([len(c) for c in chunks], dtype=np.uint32)
            offs = np.zeros(len(chunks), dtype=np.uint64); offs[1:] = np.cumsum(sizes.astype(np.uint64))[:-1]
            arena, do, dsz = bc.compress_host(src, offs, sizes, lvl)
            frames = [arena[int(do[i]):int(do[i]) + int(dsz[i])].tobytes() for i in range(len(chunks))]
        else:
            frames = [O.zstd_compress(c, lvl) for c in chunks]
        sizes = np.array([len(c) for c in chunks], dtype=np.uint32); fsz = np.array([len(f) for f in frames], dtype=np.uint32)
        blob = np.frombuffer(b"".join(frames), dtype=np.uint8); fo = np.zeros(len(frames), dtype=np.uint64); fo[1:] = np.cumsum(fsz.astype(np.uint64))[:-1]
        out, oo, osz = bc.decompress_host(blob, fo, fsz, sizes)
        ok = all(int(osz[i]) == len(c) and out[int(oo[i]):int(oo[i]) + len(c)].tobytes() == c for i, c in enumerate(chunks))
        n = len(chunks)
        buf = np.zeros(n * DESC_WORDS, dtype=np.uint32)
        rc = Z.zsmi_dbg_copyScratch(bc.ctx, 10, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)); assert rc == 0, rc
        res[label] = [int((buf.reshape(-1, DESC_WORDS)[:n, FAST_AT] == 1).sum()) if ok else -1, n]

    run("own 32 KiB", [text[i * 32768:(i + 1) * 32768] for i in range(64)])
    run("own 128 KiB", [text[i * 40000:i * 40000 + 131072] for i in range(32)])
    run("own 1 MiB", [text[i * 70000:i * 70000 + (1 << 20)] for i in range(16)])
    run("own, raw and RLE blocks among compressed ones", [text[i * 50000:i * 50000 + 200000] + noise[i * 1000:i * 1000 + 70000] + bytes(70000) + text[:100000] for i in range(16)])
    skew = np.minimum(rng.geometric(0.05, 1 << 20), 255).astype(np.uint8).tobytes()
    run("own, compressed blocks of literals only (no sequences: Huffman pays, no match does)", [skew[i * 65536:(i + 1) * 65536] for i in range(16)])
    run("own, raw blocks only", [noise[i * 100:i * 100 + 65536] for i in range(16)])
    run("own, RLE blocks only: 1 MiB of zeros (BASELINE config 1)", [bytes(1 << 20)])
    run("own, raw blocks only, 300 KB", [noise[i * 100:i * 100 + 300000] for i in range(8)])
    if elf is not None:
        run("own, ELF class (wide alphabets: flat Huffman table)", [elf[i * 65536:(i + 1) * 65536] for i in range(16)])
    if O.libzstd():
        run("libzstd 32 KiB", [text[i * 32768:(i + 1) * 32768] for i in range(64)], own=False)
        run("libzstd 300 KB level 1", [text[i * 70000:i * 70000 + 300000] for i in range(16)], own=False, lvl=1)
        run("libzstd 1 MiB level 3 (repeated tables)", [text[i * 70000:i * 70000 + (1 << 20)] for i in range(16)], own=False)
        run("libzstd 400 KB level 9", [text[i * 70000:i * 70000 + 400000] for i in range(16)], own=False, lvl=9)
    print(json.dumps(res))
    return 0


if __name__ == "__main__":
    sys.exit(main())
