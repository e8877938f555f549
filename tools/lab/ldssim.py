#!/usr/bin/env python3
"""driver for tools/lab/lds_sim.c: LDS cycles per wave step and access site of k_lz_walk, per layout / read-width variant.
usage: ldssim.py [--bytes N] [--shape L:UNIT] [--pad P] [--class NAME]"""
import argparse, ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
import _data as D, _corpus as C
so = os.path.join(HERE, "..", "_build", "libldssim.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-pthread", "-Wno-unused-function", "-o", so, os.path.join(HERE, "lds_sim.c")])
L = ctypes.CDLL(so)
ap = argparse.ArgumentParser()
ap.add_argument("--bytes", type=int, default=4 << 20)
ap.add_argument("--shape", default="3:65536")
ap.add_argument("--pad", type=int, default=24)
ap.add_argument("--cls", default="zipf")
a = ap.parse_args()
level, unit = (int(x) for x in a.shape.split(":"))
data = D.zipf_log(a.bytes).tobytes() if a.cls == "zipf" else C.corpus(a.bytes)[a.cls]
L.sim_run(data, ctypes.c_size_t(len(data)), unit, level, a.pad)
