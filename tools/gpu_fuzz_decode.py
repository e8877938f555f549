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
            f = O.zstd_compress(c, int(rng.integers(1, 20)))
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
    print("FUZZ-DECODE", "PASS" if bad == 0 else f"FAIL ({bad})")
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
