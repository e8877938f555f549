"""Randomised parity (GPU): batches of ragged chunks cut from mixed content, HIP encoder == oracle E byte for byte, HIP decoder
restores every chunk (tools/gpu_fuzz.py holds the generator; three rounds here, more by hand: python tools/gpu_fuzz.py 20)."""
import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_ragged_mixed_batches_match_oracle(monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gpu_fuzz
    monkeypatch.setattr(sys, "argv", ["gpu_fuzz.py", "3"])
    assert gpu_fuzz.main() == 0


@pytest.mark.gpu
def test_libzstd_frames_of_many_shapes_decode(monkeypatch):
    """frames made by upstream libzstd at levels 1 .. 19 (shapes this codec's encoder never emits) and damaged copies of them:
    the HIP decoder restores the input / fails where oracle D fails (tools/gpu_fuzz_decode.py; skipped without libzstd)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gpu_fuzz_decode
    monkeypatch.setattr(sys, "argv", ["gpu_fuzz_decode.py", "2"])
    assert gpu_fuzz_decode.main() == 0
