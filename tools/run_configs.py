#!/usr/bin/env python3
"""BASELINE configs 3 and 5 at their own per-GPU sizes on one MI355X (GPU box), whole output verified.

config 3: 1 GiB of text-like data as 8192 chunks of 128 KiB, level 1.  enwik9 is not in the image: the data is the text classes of
          tests/_corpus.py (XML / JSON / CSV records, Python sources, C headers; --unique-mib of distinct bytes, tiled - every
          chunk is still its own frame).
config 5: one rank's share of the 16 GiB Zipf-token log stream (2 GiB = 16384 chunks of 128 KiB), level 3.

Each: compress in ONE zsmi_compressBatchDevice call (the sub-batch loop over > ZSMI_BLOCKS_IN_FLIGHT blocks runs), decompress the
frames in one call, torch.equal over the whole output, a sample of frames decoded by oracle D and compared in size with libzstd at
the same level (labelled yardstick).  Prints one JSON line per config (kept as profiles/r2_configs.json)."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from zstandard_amd import BatchCodec
import _data as D, _oracle as O, _corpus as C

ap = argparse.ArgumentParser()
ap.add_argument("--unique-mib", type=int, default=64); ap.add_argument("--steps", type=int, default=3); ap.add_argument("--only", default="")
a = ap.parse_args()
bc = BatchCodec(0, torch.cuda.current_stream().cuda_stream)


def run(name, host, total, cs, level):
    n = total // cs
    d_u = torch.from_numpy(host).cuda()
    d_src = d_u.repeat((total + len(host) - 1) // len(host))[:total].contiguous()
    del d_u
    bound = int(bc.L.zsmi_compressBound(cs)); stride = (bound + 255) // 256 * 256
    d_frames = torch.empty(n * stride, dtype=torch.uint8, device="cuda"); d_fsz = torch.zeros(n, dtype=torch.int32, device="cuda")
    offs = np.arange(n, dtype=np.uint64) * cs; sizes = np.full(n, cs, dtype=np.uint32); foffs = np.arange(n, dtype=np.uint64) * stride
    def comp(): bc.compress_device(d_src.data_ptr(), offs, sizes, d_frames.data_ptr(), foffs, d_fsz.data_ptr(), level)
    comp(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps): comp()
    torch.cuda.synchronize()
    tc = (time.perf_counter() - t0) / a.steps
    fsz = d_fsz.cpu().numpy().astype(np.uint32)
    assert (fsz < 0xFFFFFF88).all()
    csum = int(fsz.astype(np.uint64).sum())
    d_out = torch.empty(total, dtype=torch.uint8, device="cuda"); d_osz = torch.zeros(n, dtype=torch.int32, device="cuda")
    def dec(): bc.decompress_device(d_frames.data_ptr(), foffs, fsz, d_out.data_ptr(), offs, sizes, d_osz.data_ptr())
    dec(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps): dec()
    torch.cuda.synchronize()
    td = (time.perf_counter() - t0) / a.steps
    assert (d_osz.cpu().numpy() == cs).all(), "a frame failed to decode"
    assert torch.equal(d_out, d_src), "decoded bytes differ from the input"
    # a sample spread over the distinct part: oracle D decodes it, libzstd compresses the same chunks
    uniqChunks = min(n, len(host) // cs)
    idx = np.linspace(0, uniqChunks - 1, min(256, uniqChunks)).astype(np.int64)
    ours = zs = 0
    for i in idx:
        f = d_frames[int(foffs[i]):int(foffs[i]) + int(fsz[i])].cpu().numpy().tobytes()
        chunk = host[int(i) * cs:(int(i) + 1) * cs].tobytes()
        assert O.decompress(f, cs) == chunk
        ours += len(f); zs += len(O.zstd_compress(chunk, level))
    print(json.dumps({"config": name, "chunks": n, "chunk_bytes": cs, "level": level, "bytes": total, "distinct_input_mib": len(host) >> 20,
                      "compress_gib_s": round(total / tc / 2**30, 2), "ratio": round(total / csum, 4), "size_vs_libzstd_same_level_sample": round(ours / zs, 4),
                      "decode_gib_s": round(total / td / 2**30, 2), "sub_batches": (n * ((cs + 65535) // 65536) + 16383) // 16384,
                      "verified": "torch.equal over the whole output; %d frames decoded by oracle D on the host" % len(idx)}), flush=True)
    del d_src, d_frames, d_out
    torch.cuda.empty_cache()


if a.only in ("", "3"):
    per = (a.unique_mib << 20) // 5
    parts = [C.xml_records(per), C.json_records(per), C.csv_records(per), C.pysrc(per), C.cheaders(per)]
    text = np.frombuffer(b"".join(parts), dtype=np.uint8)
    text = text[:len(text) // 131072 * 131072].copy()
    run("BASELINE config 3 shape: 1 GiB of text (corpus text classes, enwik9 absent) as 128 KiB chunks, level 1, 1 x MI355X", text, 1 << 30, 131072, 1)
if a.only in ("", "5"):
    z = D.zipf_log(2 << 30, threads=min(32, os.cpu_count() or 1))
    run("BASELINE config 5, one rank's share: 2 GiB of the Zipf-token log stream as 128 KiB chunks, level 3, 1 x MI355X", z, 2 << 30, 131072, 3)
