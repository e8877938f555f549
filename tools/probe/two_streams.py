#!/usr/bin/env python3
"""Development aid (GPU box): do the compress kernels of two half batches, issued on two streams, run side by side to any profit?
One context over 4096 chunks against two contexts (a stream each) over 2048 chunks each, and 4 x 1024; the host issues the calls alternately
and waits for all of them at the end of the timed region."""
import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data as D
from zstandard_amd import BatchCodec, _lib
n, cs, steps = 4096, 65536, 20
data = D.zipf_log(n * cs)
dsrc = torch.from_numpy(data).cuda()
Z = _lib.lib(); bound = int(Z.zsmi_compressBound(cs))
ddst = torch.empty(n * bound, dtype=torch.uint8, device="cuda"); dsz = torch.empty(n, dtype=torch.int32, device="cuda")
for parts in (1, 2, 4):
    m = n // parts
    streams = [torch.cuda.Stream() for _ in range(parts)]
    ctxs = [BatchCodec(0, s.cuda_stream) for s in streams]
    off = [np.arange(p * m, (p + 1) * m, dtype=np.uint64) * cs for p in range(parts)]
    doff = [np.arange(p * m, (p + 1) * m, dtype=np.uint64) * bound for p in range(parts)]
    sz = np.full(m, cs, dtype=np.uint32)
    def step():
        for p in range(parts):
            ctxs[p].compress_device(dsrc.data_ptr(), off[p], sz, ddst.data_ptr(), doff[p], dsz.data_ptr() + 4 * p * m, 3)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("%d stream(s) x %d chunks: %.3f ms a step = %.1f GiB/s" % (parts, m, dt * 1e3, n * cs / dt / 2**30), flush=True)
    for c in ctxs: c.close()
