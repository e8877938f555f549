#!/usr/bin/env python3
"""development aid: compare the repcode bits the sequences kernel leaves in the sequence records with the sequential rules"""
import sys, os, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O, _data as D
from zstandard_amd import BatchCodec, _lib

def seq_ref(offs, lls, init):
    r = list(init); out = []
    for o, l in zip(offs, lls):
        if l:
            if o == r[0]: v = 1
            elif o == r[1]: v = 2; r[1] = r[0]; r[0] = o
            elif o == r[2]: v = 3; r[2] = r[1]; r[1] = r[0]; r[0] = o
            else: v = 0; r[2] = r[1]; r[1] = r[0]; r[0] = o
        else:
            if o == r[1]: v = 1; r[1] = r[0]; r[0] = o
            elif o == r[2]: v = 2; r[2] = r[1]; r[1] = r[0]; r[0] = o
            else: v = 0; r[2] = r[1]; r[1] = r[0]; r[0] = o
        out.append(v)
    return out

inputs = D.mixed_inputs()
bc = BatchCodec(); Z = _lib.lib()
for name in sys.argv[1:]:
    data = inputs[name][:65536]
    src = np.frombuffer(data, dtype=np.uint8)
    bc.compress_host(src, [0], [len(data)], 3)
    hdr = np.zeros(16, dtype=np.uint32); Z.zsmi_dbg_copyScratch(bc.ctx, 2, hdr.ctypes.data_as(ctypes.c_void_p), 64)
    seq = np.zeros(8 * 2048 * 4, dtype=np.uint16); Z.zsmi_dbg_copyScratch(bc.ctx, 1, seq.ctypes.data_as(ctypes.c_void_p), 8 * 2048 * 8)
    sg = seq.reshape(8, 2048, 4)
    offs, lls, vals = [], [], []
    carry = 0
    for r in range(8):
        ns, tr = int(hdr[2 * r]), int(hdr[2 * r + 1])
        for k in range(ns):
            ll = int(sg[r, k, 0]) + (carry if k == 0 else 0)
            offs.append(int(sg[r, k, 2])); lls.append(ll); vals.append(int(sg[r, k, 1]) >> 14)
        carry = tr if ns else carry + tr
    ref = seq_ref(offs, lls, (1, 4, 8))
    bad = [i for i in range(len(ref)) if ref[i] != vals[i]]
    print(name, "nseq", len(ref), "mismatches", len(bad), bad[:10])
    for i in bad[:3]:
        lo = max(0, i - 4)
        print("   around", i, "offs", offs[lo:i + 2], "lls", lls[lo:i + 2], "ref", ref[lo:i + 2], "gpu", vals[lo:i + 2])
