#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP encoder with oracle E on a few inputs (development aid, GPU box)."""
import sys, os, ctypes
os.environ["ZSMI_DEBUG_LIB"] = "1"          # the library built with -DZSMI_DEBUG_HOOKS (zstandard_amd/_lib.py)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O, _data as D
from zstandard_amd import BatchCodec, _lib

def main():
    names = sys.argv[1:] or ["log_131072", "log_65536", "log_1000", "zeros_65536", "rand_70000", "alphabet", "log_200001", "records", "sixbit_2000", "empty", "one", "skewed"]
    inputs = D.mixed_inputs()
    bc = BatchCodec()
    L = O.lib(); Z = _lib.lib()
    L.zso_debugCandidates.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int]
    L.zso_debugWalk.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int]
    for name in names:
        data = inputs[name]
        src = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
        arena, do, dsz = bc.compress_host(src, [0], [len(data)], 3)
        frame = arena[:int(dsz[0])].tobytes() if dsz[0] < 0xFFFFFF88 else None
        ref = O.compress(data, 3)
        ok_rt = None
        if frame is not None:
            try:
                ok_rt = O.decompress(frame, len(data)) == data
            except O.OracleError as e:
                ok_rt = f"oracle error {e.code}"
        print(f"{name:14s} n={len(data):7d} gpu={dsz[0]:10d} E={len(ref):7d} same={frame == ref} roundtrip={ok_rt}")
        if frame != ref and len(data) >= 16:
            n0 = min(len(data), 131072)          # first LZ unit
            blk = data[:n0]
            dist_e = np.zeros(n0, dtype=np.uint32); L.zso_debugCandidates(dist_e.ctypes.data_as(ctypes.c_void_p), blk, n0, 3)
            dist_lo = np.zeros(131072, dtype=np.uint16); Z.zsmi_dbg_copyScratch(bc.ctx, 0, dist_lo.ctypes.data_as(ctypes.c_void_p), 131072 * 2)
            dist_g = dist_lo[:n0].astype(np.uint32)
            if n0 > 65536:
                hi = np.zeros(16384, dtype=np.uint8); Z.zsmi_dbg_copyScratch(bc.ctx, 4, hi.ctypes.data_as(ctypes.c_void_p), 16384)
                dist_g |= np.unpackbits(hi, bitorder="little")[:n0].astype(np.uint32) << 16
                dist_g[dist_lo[:n0] == 0] = 0
            if n0 <= 65536: dist_g = dist_lo[:n0].astype(np.uint32)
            hashable = max(n0 - 7, 0)                                         # (behind the hashable positions the scratch holds whatever: the walk cuts them off)
            bad = np.nonzero(dist_e[:hashable] != dist_g[:hashable])[0]
            print(f"   dist mismatches (unit 0): {len(bad)}", bad[:8], dist_e[bad[:8]], dist_g[bad[:8]])
            # the parse: sequences per block as the stitch kernel leaves them (64 output ranges x 256 records) against oracle E's
            nblk = (n0 + 65535) // 65536
            seq_e = np.zeros(3 * 65536 * nblk, dtype=np.uint32); ns_e = np.zeros(2, dtype=np.uint32)
            L.zso_debugWalk(seq_e.ctypes.data_as(ctypes.c_void_p), ns_e.ctypes.data_as(ctypes.c_void_p), blk, n0, 3)
            hdr_g = np.zeros(nblk * 64 * 4, dtype=np.uint32); Z.zsmi_dbg_copyScratch(bc.ctx, 2, hdr_g.ctypes.data_as(ctypes.c_void_p), hdr_g.nbytes)
            seq_g = np.zeros(nblk * 64 * 256 * 2, dtype=np.uint32); Z.zsmi_dbg_copyScratch(bc.ctx, 1, seq_g.ctypes.data_as(ctypes.c_void_p), seq_g.nbytes)
            hdr_g = hdr_g.reshape(nblk, 64, 4); seq_g = seq_g.reshape(nblk, 64, 256, 2); o = 0
            for b in range(nblk):
                want = seq_e[3 * o:3 * (o + int(ns_e[b]))].reshape(-1, 3); o += int(ns_e[b])
                got = []
                for g in range(64):
                    x, y = seq_g[b, g, :int(hdr_g[b, g, 0]), 0], seq_g[b, g, :int(hdr_g[b, g, 0]), 1]
                    got.append(np.stack([y >> 16, (x >> 11) & 0x1FFFF, (y & 0xFFFF) | (((x >> 28) & 1) << 16)], axis=1))
                got = np.concatenate(got)
                mlen = min(len(got), len(want)); d = np.nonzero((got[:mlen] != want[:mlen]).any(axis=1))[0]
                if len(got) != len(want) or len(d):
                    k = int(d[0]) if len(d) else mlen
                    print(f"   block {b}: sequences E {len(want)} G {len(got)}, first difference at {k}: E(start,ml,off)={want[k:k + 2].tolist()} G={got[k:k + 2].tolist()}")
            if frame is not None:
                m = next((i for i in range(min(len(frame), len(ref))) if frame[i] != ref[i]), None)
                print("   first byte diff at", m, "gpu", frame[max(0,(m or 0)-4):(m or 0)+12].hex(), "E", ref[max(0,(m or 0)-4):(m or 0)+12].hex())
    # decode check
    fx = D.fixtures()
    for name in ["text64k_l3", "small_text_l3", "tiny_l3", "zeros_1m", "random_200k", "multi_skippable", "one_byte", "text300k_l19"]:
        frame, want = fx[name]
        src = np.frombuffer(frame, dtype=np.uint8)
        arena, do, dsz = bc.decompress_host(src, [0], [len(frame)], [max(len(want), 1)])
        got = arena[:int(dsz[0])].tobytes() if dsz[0] < 0xFFFFFF88 else None
        print(f"decode {name:18s} size={dsz[0]:10d} want={len(want)} ok={got == want}")
    for n in ("csharp_alphabet", "java_a2z"):
        frame = open(os.path.join(D.GOLDEN, n + ".zst"), "rb").read(); want = open(os.path.join(D.GOLDEN, n + ".bin"), "rb").read()
        arena, do, dsz = bc.decompress_host(np.frombuffer(frame, dtype=np.uint8), [0], [len(frame)], [len(want)])
        print(f"decode golden {n}: size={dsz[0]} ok={arena[:int(dsz[0])].tobytes() == want}")

if __name__ == "__main__":
    main()
