"""Deterministic test inputs (test infrastructure)."""
import ctypes, os, subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
_DG = os.path.join(ROOT, "tools", "_build", "libdatagen.so")


def _datagen():
    src = os.path.join(ROOT, "tools", "datagen.c")
    if not os.path.exists(_DG) or os.path.getmtime(src) > os.path.getmtime(_DG):
        os.makedirs(os.path.dirname(_DG), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-pthread", "-o", _DG, src])
    return ctypes.CDLL(_DG)


def zipf_log(n, seed_lo=0x5EED, seed_hi=0xC0FFEE, threads=4, single=False) -> np.ndarray:
    D = _datagen()
    a = np.empty(n, dtype=np.uint8)
    if single:
        D.datagen_zipf_log_single(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), ctypes.c_uint64(seed_lo), ctypes.c_uint64(seed_hi))
    else:
        D.datagen_zipf_log(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), ctypes.c_uint64(seed_lo), ctypes.c_uint64(seed_hi), threads)
    return a


def alphabet_data() -> bytes:
    """the reference test's input (csharp/test/TestDecompress.cs:28-51), kept as a golden file"""
    return open(os.path.join(GOLDEN, "csharp_alphabet.bin"), "rb").read()


def fixtures():
    """libzstd 1.4.8 frames (tests/golden/gen_fixtures.py, gen_fixtures2.py): name -> (frame, content)"""
    out = {}
    for fn in ("libzstd_fixtures.npz", "libzstd_fixtures2.npz"):
        z = np.load(os.path.join(GOLDEN, fn))
        out.update({k[6:]: (z[k].tobytes(), z["data_" + k[6:]].tobytes()) for k in z.files if k.startswith("frame_")})
    return out


def mixed_inputs(seed=7):
    """name -> bytes : small and awkward inputs for round-trip tests"""
    rng = np.random.default_rng(seed)
    z = zipf_log(300000, single=True).tobytes()
    out = {
        "empty": b"", "one": b"A", "two": b"AB", "fifteen": b"0123456789abcde", "sixteen": b"0123456789abcdef",
        "aaaa_17": b"a" * 17, "abab_1000": b"ab" * 500, "zeros_65536": bytes(65536), "zeros_65537": bytes(65537),
        "zeros_131072": bytes(131072), "rand_255": rng.integers(0, 256, 255, dtype=np.uint8).tobytes(),
        "rand_256": rng.integers(0, 256, 256, dtype=np.uint8).tobytes(),
        "rand_70000": rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(),
        "log_255": z[:255], "log_256": z[:256], "log_1000": z[:1000], "log_8191": z[:8191], "log_8193": z[:8193],
        "log_65535": z[:65535], "log_65536": z[:65536], "log_65537": z[:65537], "log_65791": z[:65791], "log_65792": z[:65792],
        "log_131072": z[:131072], "log_200001": z[:200001],
        "alphabet": alphabet_data(),
        "sixbit_2000": (rng.integers(0, 64, 2000, dtype=np.uint8) + 32).astype(np.uint8).tobytes(),   # Huffman only, no match
        "nibbles": rng.integers(0, 4, 30000, dtype=np.uint8).tobytes(),
        "skewed": np.minimum(rng.geometric(0.08, 50000), 255).astype(np.uint8).tobytes(),
        "bin255": (np.arange(70000) % 251).astype(np.uint8).tobytes(),
        "allbytes": bytes(range(256)) * 300,
        "highbytes": (rng.integers(0, 40, 40000, dtype=np.uint8) + 200).astype(np.uint8).tobytes(),
        "run_mix": b"".join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 400)) for _ in range(600)),
        "longmatch": (lambda b: b + b + b)(rng.integers(0, 256, 20000, dtype=np.uint8).tobytes()),
        # matches farther back than 64 KiB (second block of an LZ unit copying from the first), incl. distance exactly 65536
        "far_70000x2": (lambda b: b + b)(rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()),
        "far_65536x2": (lambda b: b + b)(rng.integers(0, 256, 65536, dtype=np.uint8).tobytes()),
        "far_text": z[:60000] + rng.integers(0, 256, 30000, dtype=np.uint8).tobytes() + z[:41000],
    }
    rec = bytearray()
    for i in range(3000):
        rec += b"0123456789ABCDEF" + bytes([65 + i % 26])
    out["records"] = bytes(rec)
    return out
