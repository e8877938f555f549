#!/usr/bin/env python3
"""Development aid (GPU box; library built with -DZSMI_DEBUG_HOOKS -DZS_PREP_PROFILE, named by ZSMI_LIB_FILE): s_memtime ticks per phase of
k_dec_prep, averaged over the frames of a decode call."""
import sys, os, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data as D
from zstandard_amd import BatchCodec, _lib
n, cs = 8192, 32768
host = D.zipf_log(n * cs)
bc = BatchCodec(0); Z = _lib.lib()
offs = np.arange(n, dtype=np.uint64) * cs; sizes = np.full(n, cs, dtype=np.uint32)
arena, do, dsz = bc.compress_host(host, offs, sizes, 3)
frames = np.concatenate([arena[int(do[i]):int(do[i]) + int(dsz[i])] for i in range(n)])
fo = np.zeros(n, dtype=np.uint64); fo[1:] = np.cumsum(dsz.astype(np.uint64))[:-1]
out, oo, osz = bc.decompress_host(frames, fo, dsz, sizes)
assert (osz == cs).all()
buf = np.zeros(n * 4096, dtype=np.uint8)
rc = Z.zsmi_dbg_copyScratch(bc.ctx, 9, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n * 4096)); assert rc == 0, rc
prof = np.stack([buf[i * 4096 + 2048: i * 4096 + 2048 + 104].view(np.uint64) for i in range(n)]).astype(np.float64)
names = ["headers up to the Huffman description", "huf: stage the description", "huf: weight counts (readNCount)", "huf: weight FSE table", "huf: weights (two FSE states)",
         "huf: ranks, start cells", "huf: table fill", "seq: between tables", "seq: stage a description", "seq: parse (readNCount)", "seq: build", "seq: emit 16-bit cells", "whole wavefront"]
m = prof.mean(axis=0)
for k, nm in enumerate(names):
    print(f"{nm:40s} {m[k]:10.0f} ticks  {100 * m[k] / m[12]:5.1f} %")
print("whole item = %.0f s_memtime ticks (shader-clock cycles on this part: k_dec_prep at 163 K ticks an item and 16 items a CU at a time is its measured 0.97 ms per 57344 frames at ~2.4 GHz)" % m[12])
