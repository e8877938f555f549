"""ctypes access to the CPU oracle (oracle/_build/libzso.so) and, when present, to upstream
libzstd.  TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the zstandard_amd package."""
import ctypes, os, subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(ROOT, "oracle", "_build", "libzso.so")


def build_oracle(force=False):
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build_oracle())
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        L.zso_decompress.restype = sz; L.zso_decompress.argtypes = [vp, sz, vp, sz]
        L.zso_compress.restype = sz; L.zso_compress.argtypes = [vp, sz, vp, sz, ctypes.c_int]
        L.zso_compressBound.restype = sz; L.zso_compressBound.argtypes = [sz]
        L.zso_getDecompressedSize.restype = ctypes.c_ulonglong; L.zso_getDecompressedSize.argtypes = [vp, sz]
        L.zso_isError.restype = ctypes.c_uint; L.zso_isError.argtypes = [sz]
        L.zso_errorCode.restype = ctypes.c_uint; L.zso_errorCode.argtypes = [sz]
        L.zso_xxh64.restype = ctypes.c_uint64; L.zso_xxh64.argtypes = [vp, sz, ctypes.c_uint64]
        L.zso_statsGet.argtypes = [vp]
        L.zso_compressBatch.restype = ctypes.c_int
        L.zso_compressBatch.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_uint32, ctypes.c_int, ctypes.c_int]
        L.zso_decompressBatch.restype = ctypes.c_int
        L.zso_decompressBatch.argtypes = [vp, vp, vp, vp, vp, vp, vp, ctypes.c_uint32, ctypes.c_int]
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, code):
        super().__init__(f"oracle error code {code}")
        self.code = code


def decompress(frame: bytes, capacity=None) -> bytes:
    L = lib()
    if capacity is None:
        capacity = int(L.zso_getDecompressedSize(frame, len(frame)))
    out = ctypes.create_string_buffer(max(capacity, 1))
    r = L.zso_decompress(out, capacity, frame, len(frame))
    if L.zso_isError(r):
        raise OracleError(L.zso_errorCode(r))
    return out.raw[:r]


def decompress_using_dict(frame: bytes, capacity: int, dictionary: bytes) -> bytes:
    """oracle D with a dictionary (ZSTD_decompress_usingDict, ZStdDecompress.cs:2162): raw-content or formatted"""
    L = lib()
    L.zso_decompress_usingDict.restype = ctypes.c_size_t
    L.zso_decompress_usingDict.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    out = ctypes.create_string_buffer(max(capacity, 1))
    r = L.zso_decompress_usingDict(out, capacity, frame, len(frame), dictionary, len(dictionary))
    if L.zso_isError(r):
        raise OracleError(L.zso_errorCode(r))
    return out.raw[:r]


def compress(data: bytes, level=3) -> bytes:
    L = lib()
    cap = L.zso_compressBound(len(data))
    out = ctypes.create_string_buffer(cap)
    r = L.zso_compress(out, cap, data, len(data), level)
    if L.zso_isError(r):
        raise OracleError(L.zso_errorCode(r))
    return out.raw[:r]


def decode_stats(frame: bytes, capacity: int):
    L = lib()
    L.zso_statsReset()
    out = decompress(frame, capacity)
    st = np.zeros(32, dtype=np.uint32)
    L.zso_statsGet(st.ctypes.data_as(ctypes.c_void_p))
    return out, st


def compress_batch(src: np.ndarray, offsets: np.ndarray, sizes: np.ndarray, level=3, threads=1):
    """src uint8 array; returns (arena, dst_offsets, dst_sizes)"""
    L = lib()
    n = len(sizes)
    bounds = np.array([L.zso_compressBound(int(s)) for s in np.unique(sizes)])
    bmap = dict(zip(np.unique(sizes).tolist(), bounds.tolist()))
    caps = np.array([bmap[int(s)] for s in sizes], dtype=np.uint64)
    doff = np.zeros(n, dtype=np.uint64)
    doff[1:] = np.cumsum(caps)[:-1]
    arena = np.empty(int(caps.sum()), dtype=np.uint8)
    dsz = np.zeros(n, dtype=np.uint32)
    vp = ctypes.c_void_p
    rc = L.zso_compressBatch(arena.ctypes.data_as(vp), doff.ctypes.data_as(vp), dsz.ctypes.data_as(vp),
                             src.ctypes.data_as(vp), offsets.astype(np.uint64).ctypes.data_as(vp),
                             sizes.astype(np.uint32).ctypes.data_as(vp), n, level, threads)
    if rc:
        raise OracleError(-1)
    return arena, doff, dsz


# ---- optional yardstick: upstream libzstd (independent implementation, NOT the reference) ----
_z = None


def libzstd():
    global _z
    if _z is None:
        try:
            Z = ctypes.CDLL("libzstd.so.1")
        except OSError:
            _z = False
            return None
        sz, vp = ctypes.c_size_t, ctypes.c_void_p
        Z.ZSTD_compressBound.restype = sz; Z.ZSTD_compressBound.argtypes = [sz]
        Z.ZSTD_compress.restype = sz; Z.ZSTD_compress.argtypes = [vp, sz, vp, sz, ctypes.c_int]
        Z.ZSTD_decompress.restype = sz; Z.ZSTD_decompress.argtypes = [vp, sz, vp, sz]
        Z.ZSTD_isError.restype = ctypes.c_uint; Z.ZSTD_isError.argtypes = [sz]
        Z.ZSTD_versionNumber.restype = ctypes.c_uint
        _z = Z
    return _z or None


def zstd_compress(data: bytes, level=3) -> bytes:
    Z = libzstd()
    cap = Z.ZSTD_compressBound(len(data))
    out = ctypes.create_string_buffer(cap)
    r = Z.ZSTD_compress(out, cap, data, len(data), level)
    assert not Z.ZSTD_isError(r)
    return out.raw[:r]


def zstd_decompress(frame: bytes, capacity: int):
    Z = libzstd()
    out = ctypes.create_string_buffer(max(capacity, 1))
    r = Z.ZSTD_decompress(out, capacity, frame, len(frame))
    if Z.ZSTD_isError(r):
        return None
    return out.raw[:r]
