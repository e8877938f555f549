#!/usr/bin/env python3
"""BASELINE config 4 at its own size: batch-decompress 1 048 576 frames of 32 KiB on one MI355X (GPU box).
Frames are built on the device by this codec's encoder (level 3) from the Zipf-token log stream (--unique-gib of distinct data,
tiled to 32 GiB: every frame is still its own frame), decoded in ONE zsmi_decompressBatchDevice call (the > 65 536-item launch
loop of zsmi_api.hip runs 16 times), the whole 32 GiB output compared with the input on the device, a sample of frames
cross-decoded by oracle D on the host.  Prints one JSON line (kept as profiles/r2_config4.json)."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from zstandard_amd import BatchCodec
import _data as D, _oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1 << 20); ap.add_argument("--frame-size", type=int, default=32768)
ap.add_argument("--unique-gib", type=float, default=4.0); ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
n, fs = a.frames, a.frame_size
uniq = min(n, int(a.unique_gib * (1 << 30)) // fs)
t0 = time.time()
host = D.zipf_log(uniq * fs, threads=min(32, os.cpu_count() or 1))
print(f"datagen {time.time() - t0:.1f}s", file=sys.stderr, flush=True)
d_u = torch.from_numpy(host).cuda()
reps = (n + uniq - 1) // uniq
d_src = d_u.repeat(reps)[:n * fs].contiguous()
del d_u
bc = BatchCodec(0, torch.cuda.current_stream().cuda_stream)
bound = int(bc.L.zsmi_compressBound(fs)); stride = (bound + 255) // 256 * 256
d_frames = torch.empty(n * stride, dtype=torch.uint8, device="cuda"); d_fsz = torch.zeros(n, dtype=torch.int32, device="cuda")
offs = np.arange(n, dtype=np.uint64) * fs; sizes = np.full(n, fs, dtype=np.uint32); foffs = np.arange(n, dtype=np.uint64) * stride
t0 = time.perf_counter()
bc.compress_device(d_src.data_ptr(), offs, sizes, d_frames.data_ptr(), foffs, d_fsz.data_ptr(), 3)
torch.cuda.synchronize()
t_first = time.perf_counter() - t0
t0 = time.perf_counter()
bc.compress_device(d_src.data_ptr(), offs, sizes, d_frames.data_ptr(), foffs, d_fsz.data_ptr(), 3)
torch.cuda.synchronize()
t_comp = time.perf_counter() - t0
fsz = d_fsz.cpu().numpy().astype(np.uint32)
assert (fsz < 0xFFFFFF88).all()
comp = int(fsz.astype(np.uint64).sum())
print(f"compressed {n} frames: {n * fs / t_comp / 2**30:.1f} GiB/s (first call {t_first:.2f}s), ratio {n * fs / comp:.3f}", file=sys.stderr, flush=True)
d_out = torch.empty(n * fs, dtype=torch.uint8, device="cuda"); d_osz = torch.zeros(n, dtype=torch.int32, device="cuda")
def step(): bc.decompress_device(d_frames.data_ptr(), foffs, fsz, d_out.data_ptr(), offs, sizes, d_osz.data_ptr())
step(); torch.cuda.synchronize()
bc.enable_timing(True)
t0 = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
kt = bc.kernel_times()
assert (d_osz.cpu().numpy() == fs).all(), "a frame failed to decode"
assert torch.equal(d_out, d_src), "decoded bytes differ from the input"
# 1 % sample cross-decoded on the host by oracle D, spread over the batch
idx = np.linspace(0, n - 1, max(16, n // 100)).astype(np.int64)[:4096]
fr_host = [d_frames[int(foffs[i]):int(foffs[i]) + int(fsz[i])].cpu().numpy().tobytes() for i in idx[:256]]
for i, f in zip(idx[:256], fr_host):
    assert O.decompress(f, fs) == host[(int(i) % uniq) * fs:(int(i) % uniq + 1) * fs].tobytes()
print(json.dumps({"config": "BASELINE config 4: batch decompress %d frames of %d B, 1 x MI355X" % (n, fs), "frames": n, "frame_bytes": fs,
                  "distinct_input_gib": round(uniq * fs / 2**30, 2), "compressed_bytes": comp, "ratio": round(n * fs / comp, 4),
                  "decode_gib_s": round(n * fs / dt / 2**30, 2), "decode_ms_per_call": round(dt * 1e3, 2), "launch_rounds": (n + 65535) // 65536,
                  "kernels_ms_per_call": {k: round(v[0] / a.steps * 1e3, 3) for k, v in kt.items()},
                  "compress_gib_s_same_frames": round(n * fs / t_comp / 2**30, 2),
                  "verified": "torch.equal over the whole %d GiB output; 256 frames spread over the batch decoded by oracle D on the host" % (n * fs >> 30)}))
