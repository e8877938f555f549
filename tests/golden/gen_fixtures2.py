#!/usr/bin/env python3
"""Second batch of libzstd 1.4.8 frames: the constructs the first batch (gen_fixtures.py) left unpinned --
RLE literals, OF / ML repeat-mode sequence tables, blocks with >= 0x7F00 sequences (3-byte sequence count).
Inputs were found by search (which libzstd level emits what is checked below with the oracle's coverage counters);
they are deterministic (seeded numpy).  Output: libzstd_fixtures2.npz (frame_<name>, data_<name>).
libzstd is an independent implementation of the same format, not the reference."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import gen_fixtures as G
import _oracle as O


def euler_tokens(rng, K, tokbytes, nbytes):
    """K random tokens in an order where no ordered pair of tokens occurs twice: every match is one token long"""
    toks = [bytes(rng.integers(0, 256, tokbytes, dtype=np.uint8)) for _ in range(K)]
    succ = [list(rng.permutation(K)) for _ in range(K)]
    cur = 0; out = bytearray(toks[0])
    while len(out) < nbytes:
        while succ[cur] == []:
            cur = int(rng.integers(0, K))
        nxt = int(succ[cur].pop()); out += toks[nxt]; cur = nxt
    return bytes(out[:nbytes])


def build_cases():
    cases = {}
    rng = np.random.default_rng(123)
    # second block = slices of the first block glued with one and the same byte: every literal of that block is 'z'
    A = rng.integers(0, 256, 131072, dtype=np.uint8).tobytes()          # one full block (stored raw); libzstd's block = 128 KiB
    B = bytearray()
    while len(B) < 60000:
        ln = int(rng.integers(24, 80)); o = int(rng.integers(0, 131072 - ln))
        B += A[o:o + ln] + b"z"
    cases["rle_literals_l5"] = (A + bytes(B), dict(level=5), 1)
    # 16 KiB window = 16 KiB blocks, each block made of slices of the block before it: few, similar sequences per block,
    # so repeating the previous table is cheaper than describing a new one (LL, OF and ML repeat modes in one frame)
    W = 1 << 14
    for seed in range(100, 140):                                       # first seed whose frame uses OF and ML repeat modes
        rng3 = np.random.default_rng(seed)
        d = bytearray(rng3.integers(0, 256, W, dtype=np.uint8).tobytes())
        while len(d) < 4 * W:
            blk = bytearray(); prev = bytes(d[-W:])
            while len(blk) < W:
                ln = int(rng3.integers(24, 60)); o = int(rng3.integers(0, W - ln - 70))
                blk += prev[o:o + ln] + b"z"
            d += blk[:W]
        f = G.zcompress(bytes(d), level=7, window_log=14)
        _, st = O.decode_stats(f, len(d))
        if st[11] and st[15] and st[19]:
            break
    cases["repeat_tables_w14_l7"] = (bytes(d), dict(level=7, window_log=14), 15)
    rng2 = np.random.default_rng(77)
    cases["longnbseq_l16"] = (euler_tokens(rng2, 210, 3, 131072), dict(level=16), 27)        # ~43 000 sequences in one block
    cases["longnbseq_edge_l13"] = (euler_tokens(rng2, 250, 4, 131072), dict(level=13), 27)   # just above 0x7F00
    return cases


def main():
    out = {}
    for name, (data, kw, stat) in build_cases().items():
        frame = G.zcompress(data, **kw)
        got, st = O.decode_stats(frame, len(data))
        assert got == data, name
        assert st[stat] > 0, (name, "construct not emitted by this libzstd", stat)
        out["frame_" + name] = np.frombuffer(frame, dtype=np.uint8)
        out["data_" + name] = np.frombuffer(data, dtype=np.uint8)
        print(f"{name:24s} {len(data):8d} -> {len(frame):7d}  stat[{stat}] = {int(st[stat])}  nseq {int(st[30])}")
    np.savez_compressed(os.path.join(HERE, "libzstd_fixtures2.npz"), **out)


if __name__ == "__main__":
    main()
