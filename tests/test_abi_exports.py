"""CPU-side checks of the boundary: libzsmi.so builds for gfx950, loads, and exports every symbol include/zsmi.h
declares (no compute calls: there is no GPU here)."""
import ctypes, os, re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_header_symbols():
    from zstandard_amd import _lib
    path = _lib.build()
    so = ctypes.CDLL(path)
    hdr = open(os.path.join(ROOT, "include", "zsmi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(zsmi_[A-Za-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 18
    for n in names:
        assert hasattr(so, n), n
    assert sorted(names) == sorted(_lib.EXPORTS)


def test_host_only_entry_points():
    """the calls that need no device: error ABI, bound, header parse (ZStdDecompress.cs:590-622)"""
    from zstandard_amd import _lib, ZStdDecompress
    L = _lib.lib()
    assert L.zsmi_isError((1 << 64) - 20) and not L.zsmi_isError(1 << 40)
    assert L.zsmi_getErrorCode((1 << 64) - 72) == 72
    assert b"Corrupted" in L.zsmi_getErrorName((1 << 64) - 20)
    assert L.zsmi_compressBound(0) >= 9
    import _data as D
    for n, size in (("csharp_alphabet", 3409), ("java_a2z", 100000)):
        frame = open(os.path.join(D.GOLDEN, n + ".zst"), "rb").read()
        assert ZStdDecompress.GetDecompressedSize(frame) == size
    assert ZStdDecompress.GetDecompressedSize(b"\x28\xb5\x2f") == 0
    assert ZStdDecompress.GetDecompressedSize(b"garbage!") == 0
    fx = D.fixtures()
    import _oracle as O
    for name, (frame, want) in fx.items():
        assert ZStdDecompress.GetDecompressedSize(frame) == O.lib().zso_getDecompressedSize(frame, len(frame)), name


def test_no_cpu_fallback_in_product():
    """the product must not reach for the oracle"""
    for root, _, files in os.walk(os.path.join(ROOT, "zstandard_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                code = re.sub(r"//[^\n]*|#[^\n]*|/\*.*?\*/", "", txt, flags=re.S) if not f.endswith(".py") else re.sub(r"#[^\n]*", "", txt)
                assert not re.search(r"import\s+_oracle|from\s+_oracle|libzso|zso_[a-zA-Z]+\s*\(|oracle/", code), f
