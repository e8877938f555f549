#!/usr/bin/env python3
"""Development aid (GPU box): randomised parity run.  Batches of chunks of random sizes (1 B .. 400 KiB, clustered around
the kernels' boundaries) cut from mixed content are compressed by the HIP path and by oracle E; every frame must be
byte-identical and decode back (oracle D); the HIP decoder must restore every chunk."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data as D, _oracle as O
from zstandard_amd import BatchCodec

def content(rng, total):
    parts = []
    log = D.zipf_log(total // 2 + 4096)
    parts.append(log)
    parts.append(rng.integers(0, 256, total // 8, dtype=np.uint8))                       # incompressible
    parts.append(np.repeat(rng.integers(0, 256, total // 4096 + 1, dtype=np.uint8), 512)[: total // 8])   # runs
    parts.append(np.tile(np.frombuffer(D.alphabet_data(), dtype=np.uint8), total // 8 // 3409 + 1)[: total // 8])
    parts.append(rng.choice(np.array([65, 66, 67, 10], dtype=np.uint8), total // 8, p=[0.7, 0.2, 0.05, 0.05]))   # skewed
    pz = 1.0 / np.arange(1, 257) ** 1.1
    parts.append(rng.choice(256, total // 8, p=pz / pz.sum()).astype(np.uint8))          # long tail of rare symbols: 10- and 11-bit Huffman codes
    if os.environ.get("FUZZ_CORPUS"):                                                     # pieces of the mixed corpus' binary / run-heavy classes
        import _corpus as C
        for name in ("elf", "repetitive", "bintable", "hipso"):
            one = np.frombuffer(C.CLASSES[name](4 << 20), dtype=np.uint8)
            a = int(rng.integers(0, len(one) - total // 8)) if len(one) > total // 8 else 0
            parts.append(one[a:a + total // 8])
    return np.concatenate(parts)

def sizes_for(rng, n):
    edges = np.array([1, 2, 3, 4, 5, 15, 16, 17, 255, 256, 257, 8191, 8192, 8193, 65533, 65535, 65536, 65537, 65540,
                      131071, 131072, 131073, 131076, 196608, 262144, 262147])
    s = np.where(rng.random(n) < 0.4, rng.choice(edges, n) + rng.integers(-3, 4, n), rng.integers(1, 400000, n))
    return np.clip(s, 1, 400000).astype(np.uint32)

def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    bc = BatchCodec(0)
    bad = 0
    for rd in range(rounds):
        rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1000")) + rd)
        n = int(rng.integers(50, 400))
        sizes = sizes_for(rng, n)
        data = content(rng, int(sizes.sum()) + 400000)
        starts = rng.integers(0, len(data) - 400000, n).astype(np.uint64)
        level = 1 if rd % 3 == 2 else 3
        arena, do, dsz = bc.compress_host(data, starts, sizes, level)
        ea, eo, esz = O.compress_batch(data, starts, sizes, level, threads=8)
        frames = []
        for i in range(n):
            g = arena[int(do[i]):int(do[i]) + int(dsz[i])].tobytes(); e = ea[int(eo[i]):int(eo[i]) + int(esz[i])].tobytes()
            if g != e:
                bad += 1; print(f"round {rd} chunk {i} size {sizes[i]} level {level}: frames differ ({len(g)} vs {len(e)})")
            elif rd < 2 and i % 7 == 0 and O.decompress(g, int(sizes[i])) != data[int(starts[i]):int(starts[i]) + int(sizes[i])].tobytes():
                bad += 1; print(f"round {rd} chunk {i}: oracle D does not restore the chunk")
            frames.append(g)
        blob = np.frombuffer(b"".join(frames), dtype=np.uint8)
        fo = np.zeros(n, dtype=np.uint64); fo[1:] = np.cumsum(dsz.astype(np.uint64))[:-1]
        out, oo, osz = bc.decompress_host(blob, fo, dsz, sizes)
        for i in range(n):
            if int(osz[i]) != int(sizes[i]) or out[int(oo[i]):int(oo[i]) + int(sizes[i])].tobytes() != data[int(starts[i]):int(starts[i]) + int(sizes[i])].tobytes():
                bad += 1; print(f"round {rd} chunk {i} size {sizes[i]}: HIP decoder result differs (size {osz[i]})")
        print(f"round {rd}: {n} chunks, level {level}, {int(sizes.sum()) >> 20} MiB: ok so far = {bad == 0}", flush=True)
    print("FUZZ", "PASS" if bad == 0 else f"FAIL ({bad})")
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
