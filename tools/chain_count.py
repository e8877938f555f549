#!/usr/bin/env python3
"""Development aid (GPU box): how often a segment of the sequences kernel's FSE state chains fails its seam check and is run again (k_encode_sequences,
entropy_kernels.hip), per data class.  Needs a library built with -DZSMI_DEBUG_HOOKS -DZS_CHAIN_COUNT (the kernel then leaves its rounds of repair in
ZsBlockMeta.pad[0] >> 16 and K | Sb << 8 | nseq << 16 in pad[1]): ZSMI_LIB_FILE=<that library> [CLS=<class of tests/_corpus.py>] python tools/chain_count.py"""
import os, sys, ctypes
os.environ["ZSMI_DEBUG_LIB"] = "1"
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data as D
from zstandard_amd import BatchCodec, _lib
n, cs = 1024, 65536
cls = os.environ.get("CLS", "zipf")
if cls == "zipf":
    data = D.zipf_log(n * cs)
else:
    import _corpus as C
    one = np.frombuffer(C.CLASSES[cls](16 << 20), dtype=np.uint8)
    data = np.tile(one, (n * cs + len(one) - 1) // len(one))[:n * cs].copy()
dsrc = torch.from_numpy(data).cuda()
bc = BatchCodec(0); Z = _lib.lib()
off = np.arange(n, dtype=np.uint64) * cs; sz = np.full(n, cs, dtype=np.uint32)
bound = int(Z.zsmi_compressBound(cs)); doff = np.arange(n, dtype=np.uint64) * bound
ddst = torch.empty(n * bound, dtype=torch.uint8, device="cuda"); dsz = torch.empty(n, dtype=torch.int32, device="cuda")
bc.compress_device(dsrc.data_ptr(), off, sz, ddst.data_ptr(), doff, dsz.data_ptr(), 3); bc.sync()
buf = np.zeros(n * 32, dtype=np.uint8)
rc = Z.zsmi_dbg_copyScratch(bc.ctx, 3, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(buf))); assert rc == 0, rc
m = buf.view(np.uint32).reshape(n, 8)
fails = m[:, 6] & 0xFFFF; rounds = m[:, 6] >> 16      # (fails: not counted by the kernel - lane 0's ballot; rounds of repair are)
K = m[:, 7] & 0xFF; Sb = (m[:, 7] >> 8) & 0xFF; ns = m[:, 7] >> 16
print(cls, "blocks", n, "K", np.bincount(K), "Sb mean", Sb.mean(), "nseq mean", ns.mean())
print("repair rounds histogram:", np.bincount(rounds, minlength=22).tolist())
print("nseq of blocks with >= 8 rounds:", np.sort(ns[rounds >= 8])[::max(1, int((rounds >= 8).sum()) // 12)].tolist())
print("failed seams per block: mean %.2f max %d; repair rounds mean %.2f max %d; blocks with none: %d" % (fails.mean(), fails.max(), rounds.mean(), rounds.max(), (fails == 0).sum()))
