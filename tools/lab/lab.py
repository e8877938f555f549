#!/usr/bin/env python3
"""driver for tools/lab/mf_lab.c: ratio per corpus class vs libzstd for parameter sets.
usage: lab.py [--bytes N] [--shape L:CS] 'name:key=val,key=val' ..."""
import argparse, ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
import _oracle as O, _corpus as C
so = os.path.join(HERE, "..", "_build", "libmflab.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-pthread", "-Wno-unused-function", "-o", so, os.path.join(HERE, "mf_lab.c")])
L = ctypes.CDLL(so)
L.lab_compress.restype = ctypes.c_size_t
L.lab_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
L.lab_set.argtypes = [ctypes.c_char_p, ctypes.c_int]
ap = argparse.ArgumentParser()
ap.add_argument("--bytes", type=int, default=1 << 20)
ap.add_argument("--shape", default="3:65536")
ap.add_argument("--classes", default="")
ap.add_argument("--verify", action="store_true")
ap.add_argument("sets", nargs="*")
a = ap.parse_args()
level, cs = (int(x) for x in a.shape.split(":"))
cor = C.corpus(a.bytes)
if a.classes: cor = {k: v for k, v in cor.items() if k in a.classes.split(",")}
zs = {}
for name, data in cor.items():
    zs[name] = sum(len(O.zstd_compress(data[i:i + cs], level)) for i in range(0, len(data), cs))
print("%-28s" % "set" + "".join("%9s" % k for k in cor) + "    worst")
DEFAULTS = dict(shortLen=4, longLen=0, walkLog=10, look=8, fcap=8, bcap=8, rep=0, cross=0, minmatch=5, ideal=0, window=64, stepModel=1, repcost=0, longMin=8, crossCap=0, perPos=0, skipMul=4, nreps=1, repWin=64, sLog=13, lLog=13, tagBits=15, verify=1, repMin=4)
out = ctypes.create_string_buffer(cs + 4096)
for spec in a.sets or ["base:"]:
    nm, _, kv = spec.partition(":")
    prm = dict(DEFAULTS)
    for t in kv.split(","):
        if t: k, v = t.split("="); prm[k] = int(v)
    for k, v in prm.items(): L.lab_set(k.encode(), v)
    row, worst = "%-28s" % nm, 0
    for name, data in cor.items():
        e = 0
        for i in range(0, len(data), cs):
            c = data[i:i + cs]
            r = L.lab_compress(out, len(out), c, len(c), level)
            if a.verify: assert O.decompress(out.raw[:r], len(c)) == c, name
            e += r
        row += "%9.4f" % (e / zs[name]); worst = max(worst, e / zs[name])
    print(row + "%9.4f" % worst, flush=True)
