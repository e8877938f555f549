#!/usr/bin/env python3
"""Development aid (GPU box, debug library): which data classes leave the decode fast path, and at which kernel (ZsFastDesc.why).
usage: python tools/dec_why.py [class ...]   (classes of tests/_corpus.py; 4096 frames of 64 KiB each, level 3)"""
import os; os.environ["ZSMI_DEBUG_LIB"] = "1"
import sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _corpus as C
from zstandard_amd import BatchCodec, _lib
cs, n = int(os.environ.get("CS", "65536")), int(os.environ.get("N", "4096"))
bc = BatchCodec(0, torch.cuda.current_stream().cuda_stream); Z = _lib.lib()
classes = dict(C.corpus(int(os.environ.get("BYTES", str(1 << 20)))))
_lay = (ctypes.c_uint32 * 6)(); Z.zsmi_dbg_descLayout(_lay)
DESC_WORDS, FAST_AT, WHY_AT, NBSEQ_AT, LITTYPE_AT, HUFLOG_AT = (int(x) for x in _lay)     # the library's own layout of a ZsFastDesc
WHY = {0: "-", 1: "Huffman stream did not end exactly", 2: "sequence stream exhausted", 3: "offset code > 28", 4: "a sequence failed pass A's checks", 5: "last literals do not fit", 6: "content size differs", 7: "execute (other)"}
for name in (sys.argv[1:] or list(classes)):
    one = np.frombuffer(classes[name], dtype=np.uint8)
    host = np.tile(one, (n * cs + len(one) - 1) // len(one))[:n * cs]
    d_src = torch.from_numpy(host.copy()).cuda()
    offs = np.arange(n, dtype=np.uint64) * cs; sizes = np.full(n, cs, dtype=np.uint32)
    bound = int(bc.L.zsmi_compressBound(cs)); stride = (bound + 255) // 256 * 256
    doffs = np.arange(n, dtype=np.uint64) * stride
    d_dst = torch.empty(n * stride, dtype=torch.uint8, device="cuda"); d_sizes = torch.zeros(n, dtype=torch.int32, device="cuda")
    bc.compress_device(d_src.data_ptr(), offs, sizes, d_dst.data_ptr(), doffs, d_sizes.data_ptr(), 3); torch.cuda.synchronize()
    csz = d_sizes.cpu().numpy().astype(np.uint32)
    d_out = torch.empty(n * cs, dtype=torch.uint8, device="cuda"); d_osz = torch.zeros(n, dtype=torch.int32, device="cuda")
    bc.decompress_device(d_dst.data_ptr(), doffs, csz, d_out.data_ptr(), offs, sizes, d_osz.data_ptr()); torch.cuda.synchronize()
    buf = np.zeros(n * DESC_WORDS, dtype=np.uint32)
    rc = Z.zsmi_dbg_copyScratch(bc.ctx, 10, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)); assert rc == 0, rc
    d = buf.reshape(n, DESC_WORDS)
    fast, why = d[:, FAST_AT], d[:, WHY_AT]
    left = np.nonzero(fast == 0)[0]
    hist = {WHY.get(int(k), int(k)): int(v) for k, v in zip(*np.unique(why[left], return_counts=True))} if len(left) else {}
    print(f"{name:12s} frames {n}  left the fast path: {len(left)}  {hist}  first: {left[:4].tolist()}  nbSeq of those: {d[left[:4], NBSEQ_AT].tolist()} litType {d[left[:4], LITTYPE_AT].tolist()} hufLog {d[left[:4], HUFLOG_AT].tolist()}")
