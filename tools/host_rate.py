#!/usr/bin/env python3
"""Development aid (GPU box): the PCIe-inclusive rate of the host-buffer batch calls (zsmi_compressBatchHost / zsmi_decompressBatchHost:
stage through device memory, run the device form, copy back, synchronise) on pageable and on pinned host buffers.  Never the headline."""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _data as D
from zstandard_amd import BatchCodec
n, cs = 4096, 65536
bc = BatchCodec(0); L = bc.L; vp = ctypes.c_void_p
offs = np.arange(n, dtype=np.uint64) * cs; sizes = np.full(n, cs, dtype=np.uint32)
stride = 66048; doffs = np.arange(n, dtype=np.uint64) * stride
P = lambda a: a.ctypes.data_as(vp)
for pinned in (False, True):
    mk = (lambda k: torch.empty(k, dtype=torch.uint8).pin_memory().numpy()) if pinned else (lambda k: np.empty(k, dtype=np.uint8))
    src = mk(n * cs); src[:] = D.zipf_log(n * cs)
    arena = mk(n * stride); dsz = np.zeros(n, dtype=np.uint32)
    def comp(): assert L.zsmi_compressBatchHost(bc.ctx, P(src), P(offs), P(sizes), n, P(arena), P(doffs), P(dsz), 3) == 0
    comp(); comp()
    t = time.perf_counter(); [comp() for _ in range(3)]; dt = (time.perf_counter() - t) / 3
    out = mk(n * cs); osz = np.zeros(n, dtype=np.uint32)
    def dec(): assert L.zsmi_decompressBatchHost(bc.ctx, P(arena), P(doffs), P(dsz), n, P(out), P(offs), P(sizes), P(osz)) == 0
    dec(); dec()
    t = time.perf_counter(); [dec() for _ in range(3)]; dd = (time.perf_counter() - t) / 3
    assert (out == src).all()
    print("%s host buffers: compress %.1f GiB/s, decompress %.1f GiB/s (256 MiB per call, copies included)" % ("pinned" if pinned else "pageable", n * cs / dt / 2**30, n * cs / dd / 2**30))
