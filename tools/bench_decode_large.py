#!/usr/bin/env python3
"""Development aid (GPU box): decode rate of a call of LARGE frames (default 2048 frames of 1 MiB = 16 blocks each, this codec's frames)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from zstandard_amd import BatchCodec
import _data as D
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=2048); ap.add_argument("--chunk", type=int, default=1 << 20); ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--libzstd", action="store_true", help="frames built by upstream libzstd (level 3) on the host"); a = ap.parse_args()
n, cs = a.frames, a.chunk
host = D.zipf_log(min(n * cs, 1 << 30), threads=32)
host = np.tile(host, (n * cs + len(host) - 1) // len(host))[:n * cs]
bc = BatchCodec(0, torch.cuda.current_stream().cuda_stream)
d_src = torch.from_numpy(host).cuda()
bound = int(bc.L.zsmi_compressBound(cs)); stride = (bound + 255) // 256 * 256
d_frames = torch.empty(n * stride, dtype=torch.uint8, device="cuda"); d_fsz = torch.zeros(n, dtype=torch.int32, device="cuda")
offs = np.arange(n, dtype=np.uint64) * cs; sizes = np.full(n, cs, dtype=np.uint32); foffs = np.arange(n, dtype=np.uint64) * stride
if a.libzstd:
    import ctypes, _oracle as O
    assert O.libzstd(), "no libzstd here"
    zb = np.empty(n * stride, dtype=np.uint8); fsz = np.zeros(n, dtype=np.uint32); vp = ctypes.c_void_p
    rc = O.lib().zso_libzstdCompressBatch(zb.ctypes.data_as(vp), foffs.ctypes.data_as(vp), fsz.ctypes.data_as(vp), host.ctypes.data_as(vp), offs.ctypes.data_as(vp), sizes.ctypes.data_as(vp), n, 3, 16)
    assert rc == 0
    d_frames = torch.from_numpy(zb).cuda()
else:
    bc.compress_device(d_src.data_ptr(), offs, sizes, d_frames.data_ptr(), foffs, d_fsz.data_ptr(), 3); torch.cuda.synchronize()
    fsz = d_fsz.cpu().numpy().astype(np.uint32)
d_out = torch.empty(n * cs, dtype=torch.uint8, device="cuda"); d_osz = torch.zeros(n, dtype=torch.int32, device="cuda")
step = lambda: bc.decompress_device(d_frames.data_ptr(), foffs, fsz, d_out.data_ptr(), offs, sizes, d_osz.data_ptr())
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
bc.enable_timing(True); step(); torch.cuda.synchronize(); kt = bc.kernel_times()
assert (d_osz.cpu().numpy() == cs).all() and torch.equal(d_out, d_src)
print("frames %d x %d B: %.1f GiB/s" % (n, cs, n * cs * a.steps / dt / 2**30), {k: round(v[0] * 1e3, 3) for k, v in kt.items()})
