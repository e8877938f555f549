"""Loads libzsmi.so (HIP kernels + C ABI).  The library is built in-tree by __graft_entry__.build()
(hipcc --offload-arch=gfx950).  There is no fallback: if the library is missing the import fails."""
import ctypes, os, subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# development tools (tools/gpu_debug.py, tools/time_kernels.py) set ZSMI_DEBUG_LIB=1: a second library built with
# -DZSMI_DEBUG_HOOKS (scratch read-back, stage-stop timing aids); the product library has neither
DEBUG = os.environ.get("ZSMI_DEBUG_LIB", "") == "1"
LIB_PATH = os.environ.get("ZSMI_LIB_FILE") or os.path.join(_HERE, "lib", "libzsmi_debug.so" if DEBUG else "libzsmi.so")   # ZSMI_LIB_FILE: kernel-shape experiments
CSRC = os.path.join(_HERE, "csrc")


def extra_flags():
    """compile flags beyond the fixed ones: kernel-shape experiments (ZSMI_HIPCC_FLAGS) and the debug-hook switch - part of the fingerprint, so
    that a variant build is never taken for the tree's product library"""
    return os.environ.get("ZSMI_HIPCC_FLAGS", "").split() + (["-DZSMI_DEBUG_HOOKS"] if DEBUG else [])


def source_fingerprint():
    """sha256 over the kernel sources (*.hip, *.h; comments and white space do not count) and the extra compile flags: the library carries
    the one it was built from (zsmi_versionString), profiles/*_traffic.json the one it was measured at"""
    import hashlib, re
    h = hashlib.sha256()
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")) and os.path.isfile(os.path.join(CSRC, f)))
    for f in names + [os.path.join("..", "..", "include", "zsmi.h")]:
        text = open(os.path.join(CSRC, f), "r", errors="replace").read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        text = re.sub(r"\s+", "", text)
        h.update(f.encode()); h.update(text.encode())
    flags = extra_flags()
    if flags:
        h.update(" ".join(flags).encode())
    return h.hexdigest()[:16]


def built_fingerprint(path=None):
    """the fingerprint inside a built library (None: no library, or one from before the fingerprint existed)"""
    path = path or LIB_PATH
    if not os.path.exists(path):
        return None
    import re
    m = re.search(rb"sources ([0-9a-f]{16}|unknown)\)", open(path, "rb").read())
    return m.group(1).decode() if m else None


def build(force=False):
    """hipcc --offload-arch=gfx950 -> the in-tree library.  Rebuilds when the library is missing or was built from other sources than the
    tree holds (the fingerprint inside it differs: a library that travelled with the snapshot is checked, not trusted; file times say
    nothing after a copy)."""
    fp = source_fingerprint()
    if not force and not os.environ.get("ZSMI_LIB_FILE") and os.path.exists(LIB_PATH) and built_fingerprint() == fp:
        return LIB_PATH
    if os.environ.get("ZSMI_LIB_FILE") and os.path.exists(LIB_PATH) and not force:
        return LIB_PATH                                                # a variant build named by hand: left alone
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", '-DZSMI_SOURCE_FP="%s"' % fp, "-o", LIB_PATH, os.path.join(CSRC, "zsmi_api.hip")]
    cmd += extra_flags()                                           # kernel-shape experiments (-DZS_CAND_G=4 ...), the debug hooks
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


class KernelTime(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("seconds", ctypes.c_double), ("launches", ctypes.c_uint32)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, u32, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int
    L.zsmi_isError.restype = ctypes.c_uint; L.zsmi_isError.argtypes = [sz]
    L.zsmi_getErrorCode.restype = ctypes.c_uint; L.zsmi_getErrorCode.argtypes = [sz]
    L.zsmi_getErrorName.restype = ctypes.c_char_p; L.zsmi_getErrorName.argtypes = [sz]
    L.zsmi_versionString.restype = ctypes.c_char_p
    L.zsmi_compressBound.restype = sz; L.zsmi_compressBound.argtypes = [sz]
    L.zsmi_getDecompressedSize.restype = ctypes.c_ulonglong; L.zsmi_getDecompressedSize.argtypes = [vp, sz]
    L.zsmi_compress.restype = sz; L.zsmi_compress.argtypes = [vp, sz, vp, sz, i32]
    L.zsmi_decompress.restype = sz; L.zsmi_decompress.argtypes = [vp, sz, vp, sz]
    L.zsmi_decompress_usingDict.restype = sz; L.zsmi_decompress_usingDict.argtypes = [vp, sz, vp, sz, vp, sz]
    L.zsmi_createCtx.restype = vp; L.zsmi_createCtx.argtypes = [i32, vp]
    L.zsmi_freeCtx.restype = None; L.zsmi_freeCtx.argtypes = [vp]
    L.zsmi_sync.restype = i32; L.zsmi_sync.argtypes = [vp]
    L.zsmi_compressBatchDevice.restype = i32; L.zsmi_compressBatchDevice.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, i32]
    L.zsmi_decompressBatchDevice.restype = i32; L.zsmi_decompressBatchDevice.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, vp]
    L.zsmi_compressBatchHost.restype = i32; L.zsmi_compressBatchHost.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, i32]
    L.zsmi_decompressBatchHost.restype = i32; L.zsmi_decompressBatchHost.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, vp]
    L.zsmi_decompressBatchHost_usingDict.restype = i32; L.zsmi_decompressBatchHost_usingDict.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, vp, vp, sz]
    L.zsmi_decompressBatchDevice_usingDict.restype = i32; L.zsmi_decompressBatchDevice_usingDict.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, vp, vp, sz]
    L.zsmi_packFramesDevice.restype = i32; L.zsmi_packFramesDevice.argtypes = [vp, vp, vp, vp, u32, vp, vp]
    L.zsmi_enableKernelTiming.restype = i32; L.zsmi_enableKernelTiming.argtypes = [vp, i32]
    L.zsmi_getKernelTimes.restype = i32; L.zsmi_getKernelTimes.argtypes = [vp, ctypes.POINTER(KernelTime), i32]
    L.zsmi_decodeScratchBytes.restype = sz; L.zsmi_decodeScratchBytes.argtypes = [vp]
    L.zsmi_shutdown.restype = None; L.zsmi_shutdown.argtypes = []
    if DEBUG or hasattr(L, "zsmi_dbg_copyScratch"):            # (a variant build named by ZSMI_LIB_FILE may carry the hooks too)
        L.zsmi_dbg_copyScratch.restype = i32; L.zsmi_dbg_copyScratch.argtypes = [vp, i32, vp, sz]
    _lib = L
    return L


EXPORTS = ["zsmi_isError", "zsmi_getErrorName", "zsmi_getErrorCode", "zsmi_decompress", "zsmi_getDecompressedSize",
           "zsmi_compress", "zsmi_compressBound", "zsmi_createCtx", "zsmi_freeCtx", "zsmi_sync",
           "zsmi_compressBatchDevice", "zsmi_decompressBatchDevice", "zsmi_compressBatchHost", "zsmi_decompressBatchHost",
           "zsmi_decompress_usingDict", "zsmi_decompressBatchDevice_usingDict", "zsmi_decompressBatchHost_usingDict",
           "zsmi_packFramesDevice", "zsmi_enableKernelTiming", "zsmi_getKernelTimes", "zsmi_versionString", "zsmi_decodeScratchBytes", "zsmi_shutdown"]
