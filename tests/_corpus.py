"""Mixed corpus for ratio checks (test infrastructure): a Silesia surrogate built from seeded record
generators (JSON / CSV / XML-like) and from byte classes every box of this image carries (C headers,
Python sources, ELF objects).  Silesia itself is not in the image; bench.py takes it as --corpus PATH.

Every class is deterministic: file lists are sorted, generators are seeded, sizes are cut to `nbytes`."""
import glob, json, os
import numpy as np

import _data as D

DEFAULT_BYTES = 4 << 20


def _concat(paths, nbytes):
    out = bytearray()
    for p in paths:
        if len(out) >= nbytes:
            break
        try:
            if os.path.isfile(p) and not os.path.islink(p):
                with open(p, "rb") as f:
                    out += f.read(nbytes - len(out))
        except OSError:
            pass
    return bytes(out[:nbytes])


def pysrc(nbytes=DEFAULT_BYTES):
    return _concat(sorted(glob.glob("/usr/lib/python3*/*.py")), nbytes)


def cheaders(nbytes=DEFAULT_BYTES):
    return _concat(sorted(glob.glob("/usr/include/*.h")) + sorted(glob.glob("/usr/include/*/*.h")) + sorted(glob.glob("/usr/include/*/*/*.h")), nbytes)


def elf(nbytes=DEFAULT_BYTES):
    out = bytearray()
    for p in ("/lib/x86_64-linux-gnu/libc.so.6", "/usr/bin/python3.10", "/usr/bin/python3"):
        if os.path.exists(p):
            out += open(os.path.realpath(p), "rb").read(nbytes // 2)
        if len(out) >= nbytes:
            break
    return bytes(out[:nbytes])


def hipso(nbytes=DEFAULT_BYTES):
    for p in ("/opt/rocm/lib/libamdhip64.so",):
        if os.path.exists(p):
            with open(os.path.realpath(p), "rb") as f:
                f.seek(2 << 20)                      # skip the dynamic-symbol head: code + rodata + embedded kernels
                return f.read(nbytes)
    return b""


_WORDS = None


def _words(rng):
    global _WORDS
    if _WORDS is None:
        r = np.random.default_rng(12345)
        _WORDS = ["".join(chr(97 + int(c)) for c in r.integers(0, 26, int(r.integers(3, 11)))) for _ in range(4000)]
    return _WORDS


def json_records(nbytes=DEFAULT_BYTES, seed=11):
    rng = np.random.default_rng(seed)
    W = _words(rng)
    out, size, i = [], 0, 0
    cities = ["Berlin", "Paris", "Lisbon", "Tallinn", "Oslo", "Vienna", "Prague", "Zagreb", "Dublin", "Madrid"]
    while size < nbytes:
        zi = np.minimum(rng.zipf(1.3, 8), len(W)) - 1
        rec = {
            "id": 100000 + i, "user": W[int(zi[0])] + "_" + W[int(zi[1])], "active": bool(rng.integers(0, 2)),
            "score": round(float(rng.random()) * 100, 3), "city": cities[int(rng.integers(0, len(cities)))],
            "tags": [W[int(z)] for z in zi[2:2 + int(rng.integers(0, 5))]],
            "ts": "2024-%02d-%02dT%02d:%02d:%02dZ" % (1 + i // 40000 % 12, 1 + i // 1500 % 28, i // 60 % 24, i % 60, int(rng.integers(0, 60))),
            "geo": {"lat": round(float(rng.normal(48, 5)), 5), "lon": round(float(rng.normal(11, 8)), 5)},
            "msg": " ".join(W[int(z)] for z in np.minimum(rng.zipf(1.2, int(rng.integers(3, 14))), len(W)) - 1),
        }
        s = json.dumps(rec) + "\n"
        out.append(s); size += len(s); i += 1
    return "".join(out).encode()[:nbytes]


def json_flat(nbytes=DEFAULT_BYTES, seed=21):
    """flat numeric-heavy objects with a fixed key order (telemetry-like): long matches at record distance"""
    rng = np.random.default_rng(seed)
    out, size, i = [], 0, 0
    hosts = ["node-%03d" % k for k in range(40)]
    while size < nbytes:
        rec = {"seq": i, "host": hosts[int(min(rng.zipf(1.4), 40)) - 1], "cpu": int(rng.integers(0, 100)), "mem": int(rng.integers(20, 60)) * 1024,
               "ok": True, "latency_ms": round(float(rng.gamma(2.0, 3.0)), 2), "status": int((200, 200, 200, 200, 204, 404, 500)[int(rng.integers(0, 7))]),
               "region": ("eu-west-1", "us-east-1", "ap-south-1")[int(rng.integers(0, 3))], "version": "2.%d.%d" % (i // 50000, i // 5000 % 10)}
        s = json.dumps(rec) + "\n"
        out.append(s); size += len(s); i += 1
    return "".join(out).encode()[:nbytes]


def repetitive(nbytes=DEFAULT_BYTES, seed=22):
    """very compressible: long runs, long repeats of earlier material with sparse edits"""
    rng = np.random.default_rng(seed)
    base = rng.integers(32, 127, 3000, dtype=np.uint8).tobytes()
    out = bytearray()
    while len(out) < nbytes:
        k = int(rng.integers(0, 4))
        if k == 0:
            out += bytes([int(rng.integers(0, 256))]) * int(rng.integers(100, 6000))
        elif k == 1:
            a = int(rng.integers(0, 2000)); out += base[a:a + int(rng.integers(200, 1000))] * int(rng.integers(1, 12))
        elif k == 2 and len(out) > 10000:
            a = int(rng.integers(0, len(out) - 9000)); piece = bytearray(out[a:a + int(rng.integers(1000, 9000))])
            for _ in range(int(rng.integers(0, 6))):
                piece[int(rng.integers(0, len(piece)))] ^= 1 + int(rng.integers(0, 255))
            out += piece
        else:
            out += rng.integers(0, 256, int(rng.integers(10, 200)), dtype=np.uint8).tobytes()
    return bytes(out[:nbytes])


def csv_records(nbytes=DEFAULT_BYTES, seed=12):
    rng = np.random.default_rng(seed)
    W = _words(rng)
    out, size, i = ["id,timestamp,symbol,side,price,qty,venue,account,flag\n"], 0, 0
    syms = ["EURUSD", "USDJPY", "GBPUSD", "AUDUSD", "USDCHF", "BTCUSD", "XAUUSD", "ETHUSD"]
    px = {s: 1.0 + k for k, s in enumerate(syms)}
    while size < nbytes:
        s = syms[int(min(rng.zipf(1.5), len(syms))) - 1]
        px[s] *= 1 + float(rng.normal(0, 1e-4))
        line = "%d,%d.%03d,%s,%s,%.5f,%d,%s,ACC%04d,%s\n" % (
            i, 1700000000 + i // 7, int(rng.integers(0, 1000)), s, "BS"[int(rng.integers(0, 2))], px[s],
            int(rng.integers(1, 50)) * 100, ("XNAS", "XLON", "XEUR", "XTKS")[int(rng.integers(0, 4))], int(min(rng.zipf(1.4), 3000)),
            W[int(min(rng.zipf(1.3), len(W))) - 1])
        out.append(line); size += len(line); i += 1
    return "".join(out).encode()[:nbytes]


def xml_records(nbytes=DEFAULT_BYTES, seed=13):
    rng = np.random.default_rng(seed)
    W = _words(rng)
    out, size, i = ['<?xml version="1.0" encoding="UTF-8"?>\n<catalog>\n'], 0, 0
    while size < nbytes:
        zi = np.minimum(rng.zipf(1.25, 24), len(W)) - 1
        s = ('  <item id="%d" lang="%s">\n    <title>%s</title>\n    <author>%s %s</author>\n    <price currency="%s">%d.%02d</price>\n'
             '    <description>%s</description>\n    <stock>%d</stock>\n  </item>\n') % (
            i, ("en", "de", "fr", "pt")[int(rng.integers(0, 4))], " ".join(W[int(z)] for z in zi[:int(rng.integers(2, 6))]).title(),
            W[int(zi[6])].title(), W[int(zi[7])].title(), ("EUR", "USD", "GBP")[int(rng.integers(0, 3))], int(rng.integers(1, 300)), int(rng.integers(0, 100)),
            " ".join(W[int(z)] for z in zi[8:8 + int(rng.integers(5, 16))]), int(rng.integers(0, 1000)))
        out.append(s); size += len(s); i += 1
    return "".join(out).encode()[:nbytes]


def zipf(nbytes=DEFAULT_BYTES):
    return D.zipf_log(nbytes).tobytes()


def binary_table(nbytes=DEFAULT_BYTES, seed=14):
    """fixed-width little-endian records (db / sensor-table like): slowly varying fields next to noisy ones"""
    rng = np.random.default_rng(seed)
    n = nbytes // 32 + 1
    rec = np.zeros((n, 8), dtype=np.uint32)
    rec[:, 0] = np.arange(n) + 5000000
    rec[:, 1] = 1700000000 + np.arange(n) // 3
    rec[:, 2] = np.minimum(rng.zipf(1.3, n), 500)
    rec[:, 3] = (np.cumsum(rng.integers(-3, 4, n)) + 100000).astype(np.uint32)
    rec[:, 4] = rng.integers(0, 16, n)
    rec[:, 5] = 0
    rec[:, 6] = rng.integers(0, 1 << 12, n)
    rec[:, 7] = np.where(rng.random(n) < 0.1, rng.integers(0, 1 << 30, n), 0xDEADBEEF)
    return rec.tobytes()[:nbytes]


CLASSES = {
    "zipf": zipf, "pysrc": pysrc, "cheaders": cheaders, "elf": elf, "hipso": hipso,
    "json": json_records, "jsonflat": json_flat, "repetitive": repetitive, "csv": csv_records, "xml": xml_records, "bintable": binary_table,
}


def corpus(nbytes=DEFAULT_BYTES):
    """name -> bytes for every class available on this box"""
    out = {}
    for k, f in CLASSES.items():
        b = f(nbytes)
        if len(b) >= 65536:
            out[k] = b
    return out


_SEEDED = {"json": (json_records, 11), "jsonflat": (json_flat, 21), "repetitive": (repetitive, 22), "csv": (csv_records, 12), "xml": (xml_records, 13)}


def _segment(job):
    name, k, seg = job
    f, seed = _SEEDED[name]
    return f(seg, seed=seed + 1000 * k)


def corpus_distinct(nbytes=64 << 20, workers=8, seg=1 << 20):
    """name -> bytes with (up to) `nbytes` of DISTINCT data per class, for throughput numbers whose source must not sit in a cache: the record
    generators run as independent segments of `seg` bytes (seed + 1000 k) on a pool of processes (started by fork: call this BEFORE the GPU is
    touched), the Zipf log and the binary table at full size, the file classes (Python sources, C headers, ELF, libamdhip64.so) as much as the box holds."""
    import multiprocessing as mp
    out = {}
    nseg = (nbytes + seg - 1) // seg
    jobs = [(name, k, seg) for name in _SEEDED for k in range(nseg)]
    with mp.get_context("fork").Pool(max(1, workers)) as pool:
        parts = pool.map(_segment, jobs, chunksize=1)
    for i, name in enumerate(_SEEDED):
        out[name] = b"".join(parts[i * nseg:(i + 1) * nseg])[:nbytes]
    out["zipf"] = D.zipf_log(nbytes, seed_lo=0xC1A55).tobytes()
    out["bintable"] = binary_table(nbytes)
    for name, f in (("pysrc", pysrc), ("cheaders", cheaders), ("elf", elf), ("hipso", hipso)):
        b = f(nbytes)
        if len(b) >= 65536:
            out[name] = b
    return {k: out[k] for k in CLASSES if k in out}
