import ctypes, sys
import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _corpus as C, _data as D
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle', '_build', 'libzso_asan.so'))
L.zso_compress.restype = ctypes.c_size_t; L.zso_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
L.zso_decompress.restype = ctypes.c_size_t; L.zso_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
L.zso_compressBound.restype = ctypes.c_size_t; L.zso_compressBound.argtypes = [ctypes.c_size_t]
import numpy as np
rng = np.random.default_rng(1)
n = 0
items = list(D.mixed_inputs().values())
for name, data in C.corpus(1 << 19).items():
    for cs in (65536, 131072, 200000, 33333):
        items += [data[i:i + cs] for i in range(0, len(data), cs)][:3]
for c in items:
    for level in (1, 3):
        cap = L.zso_compressBound(len(c)); out = ctypes.create_string_buffer(cap)
        r = L.zso_compress(out, cap, c, len(c), level)
        assert r < (1 << 62)
        back = ctypes.create_string_buffer(max(len(c), 1))
        d = L.zso_decompress(back, len(c), out.raw[:r], r)
        assert d == len(c) and back.raw[:d] == c
        # damaged copies through the decoder (both Huffman decoders, error paths)
        fr = bytearray(out.raw[:r])
        for _ in range(6):
            b = bytearray(fr); b[int(rng.integers(0, len(b)))] ^= int(rng.integers(1, 256))
            L.zso_decompress(back, len(c), bytes(b), len(b))
        n += 1
print("asan/ubsan clean over", n, "compress + decode rounds")
