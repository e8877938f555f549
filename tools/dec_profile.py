#!/usr/bin/env python3
"""Development aid (GPU box, library built with -DZS_DEC_PROFILE): cycles per decoder phase, averaged over the frames."""
import os; os.environ["ZSMI_DEBUG_LIB"] = "1"          # the library built with -DZSMI_DEBUG_HOOKS (zstandard_amd/_lib.py)
import sys, os, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data as D
from zstandard_amd import BatchCodec, _lib
n, cs = 2048, 32768
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
prof = np.stack([buf[i * stride + (1 << 17): i * stride + (1 << 17) + 64].view(np.uint64) for i in range(n)])
names = ["literals", "seq tables", "seq decode", "seq execute", "checksum", "whole item", "  huf table", "  huf symbol loops"]
m = prof.mean(axis=0)
for k, nm in enumerate(names):
    print(f"{nm:12s} {m[k]:12.0f} ticks  {100 * m[k] / m[5]:5.1f} %")
print("literals per frame (mean):", float(np.mean([0])) )
print("whole item = %.0f s_memtime ticks (shader-clock cycles on this part: k_dec_prep at 163 K ticks an item and 16 items a CU at a time is its measured 0.97 ms per 57344 frames at ~2.4 GHz)" % m[5])
