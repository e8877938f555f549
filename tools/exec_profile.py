#!/usr/bin/env python3
"""Development aid (GPU box, library built with -DZS_EXEC_PROFILE): where k_dec_execute's wavefronts spend their time,
from s_memtime stamps per item (fast-path items only)."""
import os; os.environ["ZSMI_DEBUG_LIB"] = "1"          # the library built with -DZSMI_DEBUG_HOOKS (zstandard_amd/_lib.py); build it with ZSMI_HIPCC_FLAGS="-DZS_EXEC_PROFILE -DZS_EXEC_PREEXPAND=0" (the stamps sit in the tile-by-tile form)
import sys, os, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data as D
from zstandard_amd import BatchCodec, _lib
n, cs = 4096, 32768
host = D.zipf_log(n * cs)
bc = BatchCodec(0); Z = _lib.lib()
offs = np.arange(n, dtype=np.uint64) * cs; sizes = np.full(n, cs, dtype=np.uint32)
arena, do, dsz = bc.compress_host(host, offs, sizes, 3)
frames = np.concatenate([arena[int(do[i]):int(do[i]) + int(dsz[i])] for i in range(n)])
fo = np.zeros(n, dtype=np.uint64); fo[1:] = np.cumsum(dsz.astype(np.uint64))[:-1]
out, oo, osz = bc.decompress_host(frames, fo, dsz, sizes)
assert (osz == cs).all() and (out[:n * cs] == host).all()
stride = (1 << 17) + 64
buf = np.zeros(n * stride, dtype=np.uint8)
rc = Z.zsmi_dbg_copyScratch(bc.ctx, 5, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n * stride)); assert rc == 0, rc
prof = np.stack([buf[i * stride + (1 << 17): i * stride + (1 << 17) + 64].view(np.uint64) for i in range(n)]).astype(np.float64)
m = prof.mean(axis=0)
names = ["literals (+ scans, checks)", "matches from before the tile", "matches inside the tile", "records, bits, recent offsets", "whole item", "sequences"]
for k in (3, 0, 1, 2, 4):
    print(f"{names[k]:32s} {m[k]:12.0f} ticks  {100 * m[k] / m[4]:5.1f} %")
print("sequences per item: %.0f, tiles: %.1f" % (m[5], m[5] / 64))
