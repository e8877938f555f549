"""The oracle (restated decoder incl. both Huffman decoders, scalar encoder) under AddressSanitizer + UBSan on the CPU
(GPU sanitizers are not available on this pool): mixed inputs and corpus classes at two levels, every frame decoded back,
plus six randomly damaged copies of each frame through the decoder's error paths."""
import os, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan on this box")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_asan_run.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "clean over" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
