#!/usr/bin/env python3
"""Generates tests/golden/libzstd_fixtures_dict.npz: frames compressed WITH a dictionary by upstream libzstd (the reference has
no encoder and no public dictionary entry point: ZStdDecompress.cs:2162 is internal), for the dictionary row of SURVEY 8f.
  raw_*      : raw-content dictionary (any bytes: the segment in front of the frame, RefDictContent :2366)
  trained_*  : formatted dictionary (magic 0xEC30A437, entropy tables + recent offsets + content: LoadEntropy :2378), made by
               ZDICT_trainFromBuffer; the frames carry its dictID, their first blocks may use the dictionary's tables
               (treeless literals, repeat-mode sequence tables) and offsets reaching into its content.
Each entry: dict, frame, the expected content.  Run from the repo root: python tests/golden/gen_fixtures_dict.py"""
import ctypes, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _data as D

Z = ctypes.CDLL("libzstd.so.1")
sz, vp, cp = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_char_p
Z.ZSTD_compressBound.restype = sz; Z.ZSTD_compressBound.argtypes = [sz]
Z.ZSTD_createCCtx.restype = vp
Z.ZSTD_compress_usingDict.restype = sz; Z.ZSTD_compress_usingDict.argtypes = [vp, vp, sz, cp, sz, cp, sz, ctypes.c_int]
Z.ZSTD_createDCtx.restype = vp
Z.ZSTD_decompress_usingDict.restype = sz; Z.ZSTD_decompress_usingDict.argtypes = [vp, vp, sz, cp, sz, cp, sz]
Z.ZSTD_isError.restype = ctypes.c_uint; Z.ZSTD_isError.argtypes = [sz]
Z.ZDICT_trainFromBuffer.restype = sz; Z.ZDICT_trainFromBuffer.argtypes = [vp, sz, cp, ctypes.POINTER(sz), ctypes.c_uint]
Z.ZDICT_isError.restype = ctypes.c_uint; Z.ZDICT_isError.argtypes = [sz]

def comp(data, dic, level):
    cctx = Z.ZSTD_createCCtx(); cap = Z.ZSTD_compressBound(len(data)); out = ctypes.create_string_buffer(cap)
    r = Z.ZSTD_compress_usingDict(cctx, out, cap, data, len(data), dic, len(dic), level); assert not Z.ZSTD_isError(r)
    f = out.raw[:r]
    dctx = Z.ZSTD_createDCtx(); back = ctypes.create_string_buffer(max(len(data), 1))
    r2 = Z.ZSTD_decompress_usingDict(dctx, back, len(data), f, len(f), dic, len(dic)); assert r2 == len(data) and back.raw[:r2] == data
    return f

log = D.zipf_log(1 << 20).tobytes()
out = {}
raw = log[:6000]
for name, a, n, lvl in (("raw_small_l3", 200000, 1500, 3), ("raw_text_l19", 300000, 40000, 19), ("raw_two_blocks_l5", 400000, 150000, 5)):
    c = log[a:a + n]
    out[name + "_dict"] = np.frombuffer(raw, dtype=np.uint8); out[name + "_frame"] = np.frombuffer(comp(c, raw, lvl), dtype=np.uint8); out[name + "_want"] = np.frombuffer(c, dtype=np.uint8)
# trained dictionary: 2000 samples of ~600 bytes
samples = [log[i * 500:i * 500 + 600] for i in range(2000)]
buf = b"".join(samples); sizes = (sz * len(samples))(*[len(x) for x in samples])
dcap = 8192; dbuf = ctypes.create_string_buffer(dcap)
r = Z.ZDICT_trainFromBuffer(dbuf, dcap, buf, sizes, len(samples)); assert not Z.ZDICT_isError(r), r
trained = dbuf.raw[:r]; assert trained[:4] == bytes([0x37, 0xA4, 0x30, 0xEC])
for name, a, n, lvl in (("trained_tiny_l3", 700000, 300, 3), ("trained_small_l3", 710000, 2500, 3), ("trained_text_l19", 720000, 30000, 19), ("trained_two_blocks_l3", 500000, 140000, 3)):
    c = log[a:a + n]
    out[name + "_dict"] = np.frombuffer(trained, dtype=np.uint8); out[name + "_frame"] = np.frombuffer(comp(c, trained, lvl), dtype=np.uint8); out[name + "_want"] = np.frombuffer(c, dtype=np.uint8)
np.savez_compressed(os.path.join(HERE, "libzstd_fixtures_dict.npz"), **out)
print({k: len(v) for k, v in out.items() if k.endswith("_frame")}, "dict sizes", len(raw), len(trained))
