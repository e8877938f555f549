#!/usr/bin/env python3
"""sha256 over the kernel sources (zstandard_amd/csrc) with comments and white space removed: the library carries the fingerprint it was built
from (zsmi_versionString), profiles/*_traffic.json the one it was measured at, bench.py reports the traffic only when they agree."""
import os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fingerprint(root=ROOT):
    from zstandard_amd import _lib
    return _lib.source_fingerprint()


if __name__ == "__main__":
    print(fingerprint())
