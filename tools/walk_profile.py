#!/usr/bin/env python3
"""Development aid (GPU box, library built with -DZSMI_DEBUG_HOOKS -DZS_WALK_PROFILE): where k_lz_walk's wavefronts spend their time,
from s_memtime stamps per wavefront (tools/build_variants.sh prof:"-DZSMI_DEBUG_HOOKS -DZS_WALK_PROFILE ..." ; ZSMI_LIB_FILE=... ZSMI_DEBUG_LIB=1)."""
import os; os.environ["ZSMI_DEBUG_LIB"] = "1"
import sys, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data as D
from zstandard_amd import BatchCodec, _lib
n, cs = 4096, 65536
waves = int(os.environ.get("WAVES", "8"))
data = D.zipf_log(n * cs)
dsrc = torch.from_numpy(data).cuda()
bc = BatchCodec(0); Z = _lib.lib()
off = np.arange(n, dtype=np.uint64) * cs; sz = np.full(n, cs, dtype=np.uint32)
bound = int(Z.zsmi_compressBound(cs)); doff = np.arange(n, dtype=np.uint64) * bound
ddst = torch.empty(n * bound, dtype=torch.uint8, device="cuda"); dsz = torch.empty(n, dtype=torch.int32, device="cuda")
bc.compress_device(dsrc.data_ptr(), off, sz, ddst.data_ptr(), doff, dsz.data_ptr(), 3); bc.sync()
base = n * 256 * 16                                         # behind the blocks' range results
buf = np.zeros(base + n * waves * 80, dtype=np.uint8)
rc = Z.zsmi_dbg_copyScratch(bc.ctx, 8, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(buf))); assert rc == 0, rc
prof = buf[base:].view(np.uint64).reshape(n, waves, 10).astype(np.float64)
m = prof.mean(axis=(0, 1))
names = ["staging", "grab / loop head / barrier wait", "recent offsets + window", "candidate picks + distance loads", "scoring", "whole length", "record + bookkeeping", "-", "steps per wavefront", "active lanes per step"]
tot = m[:8].sum()
for k in range(8):
    print(f"{names[k]:36s} {m[k]:10.0f} ticks {100 * m[k] / tot:5.1f} %")
print(f"steps per wavefront {m[8]:.1f}, active lanes per step {m[9] / max(m[8], 1):.1f}, ticks per step {(m[2:7].sum()) / max(m[8], 1):.0f}")
