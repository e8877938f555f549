#!/usr/bin/env python3
"""Development aid (GPU box, library built with -DZS_K1_PROFILE): time per phase of the small-unit candidates kernel,
per wavefront (range), from s_memtime stamps (100 MHz)."""
import sys, os, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data as D
from zstandard_amd import BatchCodec, _lib
n, cs = 4096, 65536
host = D.zipf_log(n * cs)
bc = BatchCodec(0); Z = _lib.lib()
offs = np.arange(n, dtype=np.uint64) * cs; sizes = np.full(n, cs, dtype=np.uint32)
for _ in range(2):
    arena, do, dsz = bc.compress_host(host, offs, sizes, 3)
buf = np.zeros(n * 8192, dtype=np.uint8)
rc = Z.zsmi_dbg_copyScratch(bc.ctx, 4, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n * 8192)); assert rc == 0, rc
t = buf.view(np.uint64).reshape(n, 1024)[:, :16 * 8].reshape(n, 16, 8)[:, :, :5].astype(np.int64)
t0 = t[:, :, 0].min(axis=1, keepdims=True)
rel = (t - t0[:, :, None]) / 100.0            # us from the workgroup's first stamp
print("wave  range half | clear-done  A-end  B-begin  B-end   (us, mean over %d units)" % n)
for w in range(16):
    m = rel[:, w, :].mean(axis=0)
    print(f"{w:4d}  {w % 8:5d} {w // 8:4d} | {m[1]:9.1f} {m[2]:7.1f} {m[3]:8.1f} {m[4]:7.1f}")
print("workgroup: phase A %.1f us, phase B %.1f us, total %.1f us" % ((rel[:, :, 3].max(axis=1) - rel[:, :, 1].min(axis=1)).mean(), (rel[:, :, 4].max(axis=1) - rel[:, :, 3].min(axis=1)).mean(), rel[:, :, 4].max(axis=1).mean()))
