"""The drop-in boundary from plain C: tests/c/abi_consumer.c is built with gcc against include/zsmi.h + libzsmi.so
(CPU: compile and link only; GPU: run it)."""
import os, subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp):
    from zstandard_amd import _lib
    lib = _lib.build()
    exe = os.path.join(tmp, "abi_consumer")
    subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-Wall", "-Werror", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_consumer.c"),
                           "-o", exe, "-L", os.path.dirname(lib), "-lzsmi", "-Wl,-rpath," + os.path.dirname(lib)])
    return exe


def test_c_consumer_compiles_and_links(tmp_path):
    assert os.path.exists(_build(str(tmp_path)))


@pytest.mark.gpu
def test_c_consumer_runs(tmp_path):
    exe = _build(str(tmp_path))
    g = os.path.join(ROOT, "tests", "golden")
    out = subprocess.run([exe, os.path.join(g, "csharp_alphabet.zst"), os.path.join(g, "csharp_alphabet.bin")], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi_consumer ok" in out.stdout
