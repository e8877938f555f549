#!/usr/bin/env python3
"""Ratio of oracle E (= the HIP encoder, byte-identical) vs upstream libzstd per corpus class and shape.
usage: tools/ratio_table.py [--bytes N] [--classes a,b,...] [--lib PATH-to-alternative-libzso]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import _oracle as O
import _corpus as C

ap = argparse.ArgumentParser()
ap.add_argument("--bytes", type=int, default=2 << 20)
ap.add_argument("--classes", default="")
ap.add_argument("--shapes", default="3:65536,1:131072,3:131072,1:65536,3:32768")
ap.add_argument("--verify", action="store_true")
a = ap.parse_args()
shapes = [tuple(int(x) for x in s.split(":")) for s in a.shapes.split(",")]
cor = C.corpus(a.bytes)
if a.classes:
    cor = {k: v for k, v in cor.items() if k in a.classes.split(",")}
print("%-10s" % "class" + "".join("  L%d/%3dK ours/zstd (ratio)" % (l, cs >> 10) for l, cs in shapes))
worst = {}
for name, data in cor.items():
    row = "%-10s" % name
    for level, cs in shapes:
        e = z = 0
        for i in range(0, len(data), cs):
            c = data[i:i + cs]
            f = O.compress(c, level)
            if a.verify:
                assert O.decompress(f, len(c)) == c
            e += len(f); z += len(O.zstd_compress(c, level))
        row += "  %7.4f (%6.3f)          " % (e / z, len(data) / e)
        worst[(level, cs)] = max(worst.get((level, cs), 0), e / z)
    print(row, flush=True)
print("%-10s" % "worst" + "".join("  %7.4f                   " % worst[s] for s in shapes))
