#!/usr/bin/env python3
"""Decode-path measurement (BASELINE config 4 shape: many ~32 KiB frames): frames built on-device by this codec's
encoder, decoded by k_decode_frames; a sample is cross-checked against the input.  Prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from zstandard_amd import BatchCodec
import _data as D, _oracle as O

ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=57344); ap.add_argument("--chunk", type=int, default=32768)
ap.add_argument("--steps", type=int, default=5); ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--libzstd", action="store_true", help="frames built by upstream libzstd (level 3) on the host instead of this codec's encoder")
ap.add_argument("--no-kernel-timing", action="store_true", help="with --times-only: no events around the launches, the step time alone")
ap.add_argument("--times-only", action="store_true", help="per-kernel times, no output check, no CPU leg (timing-aid builds: ZSMI_LIB_FILE)"); a = ap.parse_args()
n, cs = a.frames, a.chunk
cls = os.environ.get("CLS", "zipf")          # a class of tests/_corpus.py (32 MiB of it, tiled) instead of the Zipf log
if cls == "zipf":
    host = D.zipf_log(n * cs, threads=32)
else:
    import _corpus as C
    one = np.frombuffer(C.CLASSES[cls](32 << 20), dtype=np.uint8)
    host = np.tile(one, (n * cs + len(one) - 1) // len(one))[:n * cs].copy()
bc = BatchCodec(0, torch.cuda.current_stream().cuda_stream)
d_src = torch.from_numpy(host).cuda()
bound = int(bc.L.zsmi_compressBound(cs)); stride = (bound + 255) // 256 * 256
d_frames = torch.empty(n * stride, dtype=torch.uint8, device="cuda"); d_fsz = torch.zeros(n, dtype=torch.int32, device="cuda")
offs = np.arange(n, dtype=np.uint64) * cs; sizes = np.full(n, cs, dtype=np.uint32); foffs = np.arange(n, dtype=np.uint64) * stride
if a.libzstd:
    import ctypes
    Zl = O.libzstd(); assert Zl, "no libzstd here"
    zb = np.empty(n * stride, dtype=np.uint8); fsz = np.zeros(n, dtype=np.uint32); vp_ = ctypes.c_void_p
    rc = O.lib().zso_libzstdCompressBatch(zb.ctypes.data_as(vp_), foffs.ctypes.data_as(vp_), fsz.ctypes.data_as(vp_), host.ctypes.data_as(vp_), offs.ctypes.data_as(vp_), sizes.ctypes.data_as(vp_), n, 3, 16)
    assert rc == 0
    d_frames = torch.from_numpy(zb).cuda()
else:
    bc.compress_device(d_src.data_ptr(), offs, sizes, d_frames.data_ptr(), foffs, d_fsz.data_ptr(), 3)
    torch.cuda.synchronize()
    fsz = d_fsz.cpu().numpy().astype(np.uint32)
d_out = torch.empty(n * cs, dtype=torch.uint8, device="cuda"); d_osz = torch.zeros(n, dtype=torch.int32, device="cuda")
def step(): bc.decompress_device(d_frames.data_ptr(), foffs, fsz, d_out.data_ptr(), offs, sizes, d_osz.data_ptr())
for _ in range(a.warmup): step()
torch.cuda.synchronize(); bc.enable_timing(not a.no_kernel_timing); t0 = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
kt = bc.kernel_times()
if a.times_only:
    print(os.environ.get("ZSMI_LIB_FILE", "library"), {k: round(v[0] / a.steps * 1e3, 3) for k, v in kt.items()}, "step %.3f ms" % (dt / a.steps * 1e3))
    sys.exit(0)
assert (d_osz.cpu().numpy() == cs).all()
assert torch.equal(d_out, d_src), "decoded bytes differ from the input"
comp = int(fsz.astype(np.uint64).sum())
# CPU baseline: oracle D on all host cores over a bounded sample
def usable_cores():
    k = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max": k = min(k, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return min(k, 64)
m = min(n, 4096); cores = usable_cores()
fr = d_frames.cpu().numpy()
L = O.lib(); import ctypes
dst = np.empty(m * cs, dtype=np.uint8); dsz = np.zeros(m, dtype=np.uint32); vp = ctypes.c_void_p
t1 = time.perf_counter()
L.zso_decompressBatch(dst.ctypes.data_as(vp), offs[:m].ctypes.data_as(vp), sizes[:m].ctypes.data_as(vp), dsz.ctypes.data_as(vp), fr.ctypes.data_as(vp), foffs[:m].ctypes.data_as(vp), fsz[:m].ctypes.data_as(vp), m, cores)
cdt = time.perf_counter() - t1
assert (dst == host[:m * cs]).all()
secs = sum(v[0] for v in kt.values()); launches = kt["k_decode_frames"][1]
print({k: round(v[0] / a.steps * 1e3, 3) for k, v in kt.items()}, file=sys.stderr)
print(json.dumps({"metric": "GiB/s decompress (output bytes), frames of %d B" % cs, "value": round(n * cs * a.steps / dt / 2**30, 3), "frames": n, "ms_per_step": round(dt / a.steps * 1e3, 3), "decode_scratch_bytes": int(bc.L.zsmi_decodeScratchBytes(bc.ctx)),
                  "roofline": {"bound": "hbm", "kernel": "decode kernels together", "achieved": round((n * cs + comp) * a.steps / secs / 1e9, 2), "peak": 8000.0, "unit": "GB/s"},
                  "cpu_baseline": {"value": round(m * cs / cdt / 2**30, 3), "unit": "GiB/s", "cores": cores, "kind": "port", "sample": "%d frames, oracle D (restated reference decoder)" % m}}))
