#!/usr/bin/env python3
"""driver for tools/lab/walk_lab.c: ratio per corpus class vs libzstd and walk work for parameter sets.
usage: wlab.py [--bytes N] [--shape L:CS] [--verify] 'name:key=val,key=val' ..."""
import argparse, ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
import _oracle as O, _corpus as C
so = os.path.join(HERE, "..", "_build", "libwalklab.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-pthread", "-Wno-unused-function", "-o", so, os.path.join(HERE, "walk_lab.c")])
L = ctypes.CDLL(so)
L.wl_compress.restype = ctypes.c_size_t
L.wl_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
L.wl_set.argtypes = [ctypes.c_char_p, ctypes.c_int]
ap = argparse.ArgumentParser()
ap.add_argument("--bytes", type=int, default=1 << 20)
ap.add_argument("--shape", default="3:65536")
ap.add_argument("--classes", default="")
ap.add_argument("--verify", action="store_true")
ap.add_argument("--extra", action="store_true")
ap.add_argument("--same", action="store_true", help="assert byte equality with oracle E (base parameters)")
ap.add_argument("sets", nargs="*")
a = ap.parse_args()
level, cs = (int(x) for x in a.shape.split(":"))
cor = C.corpus(a.bytes)
if a.extra:
    import numpy as np
    rng = np.random.default_rng(5)
    per = rng.integers(0, 256, 1000, dtype=np.uint8).tobytes()
    cor["period1k"] = (per * (a.bytes // 1000 + 1))[:a.bytes]
    z = bytearray(a.bytes)
    for i in rng.integers(0, a.bytes, a.bytes // 3000): z[int(i)] = 1 + int(rng.integers(0, 255))
    cor["zerosN"] = bytes(z)
if a.classes: cor = {k: v for k, v in cor.items() if k in a.classes.split(",")}
zs = {}
for name, data in cor.items():
    zs[name] = sum(len(O.zstd_compress(data[i:i + cs], level)) for i in range(0, len(data), cs))
print("%-24s" % "set" + "".join("%8s" % k[:7] for k in cor) + "   worst | steps/KiB scored/KiB seq/KiB kept/KiB merged/KiB")
DEFAULTS = dict(walkLog=10, crossMax=16384, look=0, merge=0, window=64, repwin=8, longEven=0, carryRep=0, skipFirst=0, approx=0, estLong=8, estShort=5, estRep=4, estSkip=5, useBack=1, estOff=1, estRun=0, initRep=0, bcap=8, fcap=8, walign=0, repMode=0)
out = ctypes.create_string_buffer(cs + 4096)
for spec in a.sets or ["base:"]:
    nm, _, kv = spec.partition(":")
    prm = dict(DEFAULTS)
    for t in kv.split(","):
        if t: k, v = t.split("="); prm[k] = int(v)
    for k, v in prm.items(): L.wl_set(k.encode(), v)
    row, worst, tot = "%-24s" % nm, 0, 0
    st = (ctypes.c_ulonglong * 12)(); L.wl_stats(st)
    for name, data in cor.items():
        e = 0
        for i in range(0, len(data), cs):
            c = data[i:i + cs]
            r = L.wl_compress(out, len(out), c, len(c), level)
            if a.verify: assert O.decompress(out.raw[:r], len(c)) == c, name
            if a.same: assert out.raw[:r] == O.compress(c, level), name
            e += r
        tot += len(data)
        row += "%8.4f" % (e / zs[name]); worst = max(worst, e / zs[name])
    L.wl_stats(st)
    k = tot / 1024
    print(row + "%8.4f | %7.1f %7.1f %7.1f %7.1f %7.2f" % (worst, st[0] / k, st[2] / k, st[3] / k, st[4] / k, st[7] / k) + "  | wave efficiency (steps / lanes x slowest) 16: %.3f 32: %.3f 64: %.3f, empty steps %.3f" % (st[0] / max(1, 16 * st[9]) if False else st[0] / max(1, 16 * st[9]), st[0] / max(1, 32 * st[8]), st[0] / max(1, 64 * st[10]), st[1] / max(1, st[0])), flush=True)
