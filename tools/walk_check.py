#!/usr/bin/env python3
"""Development aid (GPU box): the walk kernel's output (sequences per block, read back from the scratch) against oracle E's parse, for a batch of
N chunks of the Zipf log -- the entropy kernels are NOT launched (ZSMI_STOP_AFTER_WALK, debug-hooks library), so a wrong parse cannot take
anything else down.  usage: walk_check.py [chunks] [chunk_size] [level]"""
import os, sys, ctypes
os.environ["ZSMI_DEBUG_LIB"] = "1"; os.environ["ZSMI_STOP_AFTER_WALK"] = "1"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O, _data as D
from zstandard_amd import BatchCodec, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cs = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
level = int(sys.argv[3]) if len(sys.argv) > 3 else 3
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
data = D.zipf_log(n * cs)
bc = BatchCodec(0); Z = _lib.lib(); L = O.lib()
L.zso_debugWalk.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int]
offs = np.arange(n, dtype=np.uint64) * cs; sizes = np.full(n, cs, dtype=np.uint32)
for _ in range(reps):
    bc.compress_host(data, offs, sizes, level)           # (the frames are not written: sizes come back as they were)
bpc = (cs + 65535) // 65536                              # blocks per chunk
nb = n * bpc
hdr = np.zeros(nb * 64 * 4, dtype=np.uint32); rc = Z.zsmi_dbg_copyScratch(bc.ctx, 2, hdr.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(hdr.nbytes)); assert rc == 0, rc
seq = np.zeros(nb * 64 * 256 * 2, dtype=np.uint32); rc = Z.zsmi_dbg_copyScratch(bc.ctx, 1, seq.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(seq.nbytes)); assert rc == 0, rc
hdr = hdr.reshape(nb, 64, 4); seq = seq.reshape(nb, 64, 256, 2)
bad = 0
step = max(1, n // 64)
for ci in list(range(0, n, step)) + [n - 1]:
    chunk = data[ci * cs:(ci + 1) * cs].tobytes()
    for u0 in range(0, cs, 131072):
        unit = chunk[u0:u0 + 131072]
        se = np.zeros(3 * 65536, dtype=np.uint32); ne = np.zeros(2, dtype=np.uint32)
        L.zso_debugWalk(se.ctypes.data_as(ctypes.c_void_p), ne.ctypes.data_as(ctypes.c_void_p), unit, len(unit), level)
        o = 0
        for bi in range((len(unit) + 65535) // 65536):
            b = ci * bpc + u0 // 65536 + bi
            want = se[3 * o:3 * (o + int(ne[bi]))].reshape(-1, 3); o += int(ne[bi])
            got = []
            for g in range(64):
                nsq, first = int(hdr[b, g, 0]), int(hdr[b, g, 3])
                if nsq > 256 or first != 0: got = None; break
                x, y = seq[b, g, :nsq, 0], seq[b, g, :nsq, 1]
                got.append(np.stack([y >> 16, (x >> 11) & 0x1FFFF, (y & 0xFFFF) | (((x >> 28) & 1) << 16)], axis=1))
            got = np.concatenate(got) if got is not None else None
            if got is None or got.shape != want.shape or (got != want).any():
                bad += 1
                if bad <= 5:
                    k = None
                    if got is not None:
                        m = min(len(got), len(want)); d = np.nonzero((got[:m] != want[:m]).any(axis=1))[0]; k = int(d[0]) if len(d) else m
                    print(f"chunk {ci} block {b}: sequences E {len(want)} G {None if got is None else len(got)} first difference at {k}", None if k is None or got is None else (want[k:k + 2].tolist(), got[k:k + 2].tolist()))
print(f"{n} chunks of {cs} B, level {level}: blocks checked with a different parse: {bad}")
sys.exit(1 if bad else 0)
