"""Host-side mirror of the reference's public surface, over the C ABI of libzsmi.so.

  ZStdDecompress      static class of csharp/src/ZStdDecompress.cs:37-42 (Decompress :2182-2191, GetDecompressedSize :590-622):
                      never raises on corrupt input, returns the size or the reference's error value (uint)(-code).
  ZstdDecompressor    Java class java/src/main/java/com/epam/deltix/zstd/ZstdDecompressor.java:18-34: raises RuntimeError
                      like Util.java:32-40.
  ZstdCompressor      the compressor the reference lacks (north_star), same calling conventions.
  BatchCodec          the batch hot path on device memory (torch tensors are only a handle to device memory here).

Everything computes on the GPU through libzsmi.so; nothing here falls back to a CPU codec.
"""
import ctypes
import numpy as np
from . import _lib

ERROR_MAX = 0xFFFFFF88          # ZStdErrors.cs:95-98 : IsError(c) = c > (uint)-120


def _buf(b):
    """bytes-like -> (ctypes pointer/obj, length)"""
    if isinstance(b, np.ndarray):
        return b.ctypes.data_as(ctypes.c_void_p), b.nbytes
    if isinstance(b, (bytes, bytearray, memoryview)):
        mv = memoryview(b)
        if isinstance(b, bytes):
            return ctypes.c_char_p(b), len(b)
        return (ctypes.c_char * len(mv)).from_buffer(b), len(mv)
    raise TypeError(type(b))


class ZStdDecompress:
    """EPAM.Deltix.ZStd.ZStdDecompress (static). Sizes are 32-bit in the reference (size_t = UInt32, ZStdDecompress.cs:14)."""

    @staticmethod
    def Decompress(dst, src, dstCapacity=None, srcSize=None) -> int:
        L = _lib.lib()
        d, dn = _buf(dst)
        s, sn = _buf(src)
        r = L.zsmi_decompress(d, dn if dstCapacity is None else dstCapacity, s, sn if srcSize is None else srcSize)
        return r & 0xFFFFFFFF

    @staticmethod
    def GetDecompressedSize(src, srcSize=None) -> int:
        L = _lib.lib()
        s, sn = _buf(src)
        return int(L.zsmi_getDecompressedSize(s, sn if srcSize is None else srcSize))

    @staticmethod
    def IsError(code: int) -> bool:
        return (code & 0xFFFFFFFF) > ERROR_MAX


class ZstdDecompressor:
    """com.epam.deltix.zstd.ZstdDecompressor"""

    def decompress(self, input, inputOffset, inputLength, output, outputOffset, maxOutputLength) -> int:
        L = _lib.lib()
        src = bytes(memoryview(input)[inputOffset:inputOffset + inputLength])
        tmp = ctypes.create_string_buffer(max(maxOutputLength, 1))
        r = L.zsmi_decompress(tmp, maxOutputLength, src, len(src))
        if L.zsmi_isError(r):
            raise RuntimeError(f"{L.zsmi_getErrorName(r).decode()}: offset={inputOffset}")     # Util.java:32-40
        memoryview(output)[outputOffset:outputOffset + r] = tmp.raw[:r]
        return int(r)

    @staticmethod
    def getDecompressedSize(input, offset, length) -> int:
        L = _lib.lib()
        src = bytes(memoryview(input)[offset:offset + length])
        if len(src) < 5 or src[:4] != b"\x28\xb5\x2f\xfd":
            raise RuntimeError("Invalid magic prefix: offset=%d" % offset)                      # ZstdFrameDecompressor.java:928
        fhd = src[4]
        if (fhd >> 6) == 0 and not (fhd >> 5) & 1:
            return -1                                                                             # :922 returns -1 when absent
        return int(L.zsmi_getDecompressedSize(src, len(src)))


class ZstdCompressor:
    """One frame per call; level <= 2 fast parameters, level >= 3 default parameters."""

    def __init__(self, level=3):
        self.level = level

    @staticmethod
    def compressBound(n: int) -> int:
        return int(_lib.lib().zsmi_compressBound(n))

    def compress(self, src) -> bytes:
        L = _lib.lib()
        s, sn = _buf(src)
        cap = L.zsmi_compressBound(sn)
        out = ctypes.create_string_buffer(cap)
        r = L.zsmi_compress(out, cap, s, sn, self.level)
        if L.zsmi_isError(r):
            raise RuntimeError(L.zsmi_getErrorName(r).decode())
        return out.raw[:r]


class BatchCodec:
    """n independent chunks <-> n frames on one GPU.  Arrays of offsets/sizes live on the host (numpy);
    data lives on the device.  `stream` is a raw hipStream_t handle (e.g. torch.cuda.current_stream().cuda_stream)."""

    def __init__(self, device=-1, stream=None):
        self.L = _lib.lib()
        self.ctx = self.L.zsmi_createCtx(device, ctypes.c_void_p(stream) if stream else None)
        if not self.ctx:
            raise RuntimeError("zsmi_createCtx failed: no usable HIP device (this codec has no CPU path)")

    def close(self):
        if self.ctx:
            self.L.zsmi_freeCtx(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        rc = self.L.zsmi_sync(self.ctx)
        if rc:
            raise RuntimeError(f"device error {rc}")

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(ctypes.c_void_p)

    def compress_device(self, d_src_ptr, src_offsets, src_sizes, d_dst_ptr, dst_offsets, d_dst_sizes_ptr, level=3):
        so = np.ascontiguousarray(src_offsets, dtype=np.uint64); ss = np.ascontiguousarray(src_sizes, dtype=np.uint32)
        do = np.ascontiguousarray(dst_offsets, dtype=np.uint64)
        rc = self.L.zsmi_compressBatchDevice(self.ctx, ctypes.c_void_p(d_src_ptr), self._p(so), self._p(ss), len(ss),
                                             ctypes.c_void_p(d_dst_ptr), self._p(do), ctypes.c_void_p(d_dst_sizes_ptr), level)
        if rc:
            raise RuntimeError(f"zsmi_compressBatchDevice: error {rc}")

    def decompress_device(self, d_src_ptr, src_offsets, src_sizes, d_dst_ptr, dst_offsets, dst_caps, d_dst_sizes_ptr):
        so = np.ascontiguousarray(src_offsets, dtype=np.uint64); ss = np.ascontiguousarray(src_sizes, dtype=np.uint32)
        do = np.ascontiguousarray(dst_offsets, dtype=np.uint64); dc = np.ascontiguousarray(dst_caps, dtype=np.uint32)
        rc = self.L.zsmi_decompressBatchDevice(self.ctx, ctypes.c_void_p(d_src_ptr), self._p(so), self._p(ss), len(ss),
                                               ctypes.c_void_p(d_dst_ptr), self._p(do), self._p(dc), ctypes.c_void_p(d_dst_sizes_ptr))
        if rc:
            raise RuntimeError(f"zsmi_decompressBatchDevice: error {rc}")

    def pack_device(self, d_frames_ptr, dst_offsets, d_sizes_ptr, n, d_packed_ptr, d_packed_offsets_ptr):
        """frames sitting at dst_offsets (sizes on the device) -> one contiguous run at d_packed; d_packed_offsets[n + 1] (device, uint64)"""
        do = np.ascontiguousarray(dst_offsets, dtype=np.uint64)
        rc = self.L.zsmi_packFramesDevice(self.ctx, ctypes.c_void_p(d_frames_ptr), self._p(do), ctypes.c_void_p(d_sizes_ptr), n,
                                          ctypes.c_void_p(d_packed_ptr), ctypes.c_void_p(d_packed_offsets_ptr))
        if rc:
            raise RuntimeError(f"zsmi_packFramesDevice: error {rc}")

    def compress_host(self, src: np.ndarray, src_offsets, src_sizes, level=3):
        """returns (arena uint8, dst_offsets uint64, dst_sizes uint32)"""
        so = np.ascontiguousarray(src_offsets, dtype=np.uint64); ss = np.ascontiguousarray(src_sizes, dtype=np.uint32)
        n = len(ss)
        bounds = np.array([self.L.zsmi_compressBound(int(s)) for s in ss], dtype=np.uint64) if n < 4096 else \
            (ss.astype(np.uint64) + (ss.astype(np.uint64) >> 8) + 3 * (ss.astype(np.uint64) // 65536 + 1) + 18 + 64)
        do = np.zeros(n, dtype=np.uint64)
        if n > 1:
            do[1:] = np.cumsum(bounds)[:-1]
        arena = np.zeros(int(bounds.sum()), dtype=np.uint8)
        dsz = np.zeros(n, dtype=np.uint32)
        rc = self.L.zsmi_compressBatchHost(self.ctx, self._p(src), self._p(so), self._p(ss), n, self._p(arena), self._p(do), self._p(dsz), level)
        if rc:
            raise RuntimeError(f"zsmi_compressBatchHost: error {rc}")
        return arena, do, dsz

    def decompress_host(self, src: np.ndarray, src_offsets, src_sizes, dst_caps, dictionary: bytes = b""):
        """dictionary: every frame is decoded with it (raw content or a formatted dictionary; ZSTD_decompress_usingDict,
        ZStdDecompress.cs:2162)"""
        so = np.ascontiguousarray(src_offsets, dtype=np.uint64); ss = np.ascontiguousarray(src_sizes, dtype=np.uint32)
        dc = np.ascontiguousarray(dst_caps, dtype=np.uint32)
        n = len(ss)
        do = np.zeros(n, dtype=np.uint64)
        if n > 1:
            do[1:] = np.cumsum(dc.astype(np.uint64))[:-1]
        arena = np.zeros(max(int(dc.astype(np.uint64).sum()), 1), dtype=np.uint8)
        dsz = np.zeros(n, dtype=np.uint32)
        if dictionary:
            dbuf = np.frombuffer(dictionary, dtype=np.uint8)
            rc = self.L.zsmi_decompressBatchHost_usingDict(self.ctx, self._p(src), self._p(so), self._p(ss), n, self._p(arena), self._p(do), self._p(dc), self._p(dsz),
                                                           self._p(dbuf), len(dbuf))
        else:
            rc = self.L.zsmi_decompressBatchHost(self.ctx, self._p(src), self._p(so), self._p(ss), n, self._p(arena), self._p(do), self._p(dc), self._p(dsz))
        if rc:
            raise RuntimeError(f"zsmi_decompressBatchHost: error {rc}")
        return arena, do, dsz

    def kernel_times(self):
        out = (_lib.KernelTime * 16)()
        k = self.L.zsmi_getKernelTimes(self.ctx, out, 16)
        return {out[i].name.decode(): (out[i].seconds, out[i].launches) for i in range(k)}

    def enable_timing(self, on=True):
        """True / 1: events around every launch; 2: only around the dominant kernel (k_lz_walk*, k_dec_execute); False: off"""
        self.L.zsmi_enableKernelTiming(self.ctx, 2 if on == 2 else (1 if on else 0))
