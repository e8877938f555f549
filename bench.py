#!/usr/bin/env python3
"""Headline benchmark: GiB/s compress @ level 3, 64 KiB chunks (BASELINE.json), on N GPUs of one node.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the compress path (candidates -> walk -> entropy -> frame assembly) over one batch of
--chunks chunks per GPU, inputs already resident in HBM.  Workload = BASELINE config[1] shape (independent
64 KiB chunks, level 3); the Silesia corpus is not available offline, so the chunks come from the synthetic
Zipf-token log stream of SURVEY.md 8(d) unless --corpus PATH is given.  Chunks shard across ranks with no
data-path collective (weak scaling: every GPU gets its own --chunks chunks).
Rank 0 prints one JSON line.  The same line carries "decode" (BASELINE config 4 shape: many ~32 KiB frames, the one path the
reference implements itself), "ratio_by_class" (HIP encoder vs upstream libzstd on the mixed corpus of tests/_corpus.py) and
"libzstd_yardstick" (upstream libzstd on the host cores: NOT the reference, which has no encoder and cannot run here).

--gpus N without a torch.distributed environment (no WORLD_SIZE) starts the N ranks itself, before anything touches the GPU.
"""
import argparse, ctypes, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def usable_cores():
    """threads for the CPU baseline: the cgroup CPU quota if there is one, else the affinity mask (the GPU boxes expose all
    host CPUs but give a job a share of about 16); ZSMI_CPU_THREADS overrides"""
    if os.environ.get("ZSMI_CPU_THREADS"):
        return max(1, int(os.environ["ZSMI_CPU_THREADS"]))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return min(n, 64)


def source_fingerprint():
    """fingerprint of the kernel sources (comments and white space do not count): profiles/*_traffic.json carries the one it was measured at"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from src_fingerprint import fingerprint
    return fingerprint(ROOT)


def self_launch(args):
    """--gpus N given, no rank environment: run the N ranks under torch.distributed.run (one process per GPU) and pass its
    output through.  Nothing in this process has touched the GPU yet."""
    import subprocess
    port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def threaded(fn, items, threads):
    """ctypes calls release the GIL: a thread pool keeps `threads` host cores busy"""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(threads) as ex:
        return list(ex.map(fn, items))


def load_corpus(path, nbytes, rank):
    data = np.fromfile(path, dtype=np.uint8)
    if len(data) == 0:
        raise SystemExit("empty corpus")
    reps = (nbytes + len(data) - 1) // len(data)
    return np.tile(data, reps)[:nbytes]



def class_corpus(args):
    """the data classes of throughput_by_class at >= 64 MiB of DISTINCT bytes each (round 3 tiled 1 MiB of a class 256 x: 16 chunks, permanently cache resident).
    Built on a pool of forked processes, so it runs before anything touches the GPU."""
    import _corpus as C
    t0 = time.time()
    classes = dict(C.corpus_distinct(args.class_bytes, workers=usable_cores()))
    rng = np.random.default_rng(7)
    classes["zeros"] = bytes(1 << 20)
    classes["random"] = rng.integers(0, 256, args.class_bytes, dtype=np.uint8).tobytes()
    classes["period1000"] = rng.integers(0, 256, 1000, dtype=np.uint8).tobytes()
    return classes, round(time.time() - t0, 1)


def throughput_by_class(bc, args, torch, classes, gen_s):
    """GiB/s and ratio of the compress path at this level / chunk size per data class: every class of tests/_corpus.py plus all-zero,
    uniform-random and period-1000 inputs, each at `distinct_bytes` of distinct data tiled (on the device) to the batch.  Compress: the headline's batch
    shape and timing (device-resident input, events excluded), fewer steps.  Decode: the class cut in frames of the decode leg's shape and count
    (--decode-frames x --decode-frame-size), built by this codec, decoded in ONE call as the decode leg does - so the classes compare with the decode
    leg and with each other.  Not part of `value`."""
    cs = args.chunk_size
    n = args.chunks                                  # the headline's batch size
    nbytes = n * cs
    nf, fs = (args.decode_frames or 57344), args.decode_frame_size
    offs = np.arange(n, dtype=np.uint64) * cs; sizes = np.full(n, cs, dtype=np.uint32)
    bound = int(bc.L.zsmi_compressBound(cs)); stride = (bound + 255) // 256 * 256
    doffs = np.arange(n, dtype=np.uint64) * stride
    d_dst = torch.empty(n * stride, dtype=torch.uint8, device="cuda"); d_sizes = torch.zeros(n, dtype=torch.int32, device="cuda")
    foffs_in = np.arange(nf, dtype=np.uint64) * fs; fsizes = np.full(nf, fs, dtype=np.uint32)
    fbound = int(bc.L.zsmi_compressBound(fs)); fstride = (fbound + 255) // 256 * 256
    ffo = np.arange(nf, dtype=np.uint64) * fstride
    d_frames = torch.empty(nf * fstride, dtype=torch.uint8, device="cuda"); d_fsz = torch.zeros(nf, dtype=torch.int32, device="cuda")
    d_out = torch.empty(nf * fs, dtype=torch.uint8, device="cuda"); d_osz = torch.zeros(nf, dtype=torch.int32, device="cuda")
    out = {}
    def tiled(d_one, total):
        reps = (total + d_one.numel() - 1) // d_one.numel()
        return d_one.repeat(reps)[:total].contiguous()
    for name, data in classes.items():
        d_one = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
        d_src = tiled(d_one, nbytes)
        step = lambda: bc.compress_device(d_src.data_ptr(), offs, sizes, d_dst.data_ptr(), doffs, d_sizes.data_ptr(), args.level)
        step(); torch.cuda.synchronize()
        k = 5
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        csz = d_sizes.cpu().numpy().astype(np.uint32)
        assert (csz < 0xFFFFFF88).all(), name
        out[name] = {"GiB/s": round(nbytes * k / dt / (1 << 30), 1), "ratio": round(nbytes / float(csz.astype(np.uint64).sum()), 3), "distinct_bytes": len(data)}
        del d_src
        # decode: nf frames of fs bytes of the class, built here (untimed), decoded in one call
        d_in = tiled(d_one, nf * fs)
        bc.compress_device(d_in.data_ptr(), foffs_in, fsizes, d_frames.data_ptr(), ffo, d_fsz.data_ptr(), args.level); torch.cuda.synchronize()
        fsz = d_fsz.cpu().numpy().astype(np.uint32)
        assert (fsz < 0xFFFFFF88).all(), name
        dstep = lambda: bc.decompress_device(d_frames.data_ptr(), ffo, fsz, d_out.data_ptr(), foffs_in, fsizes, d_osz.data_ptr())
        dstep(); torch.cuda.synchronize()
        kd = 3
        t0 = time.perf_counter()
        for _ in range(kd):
            dstep()
        torch.cuda.synchronize()
        ddt = time.perf_counter() - t0
        assert (d_osz.cpu().numpy() == fs).all() and torch.equal(d_out, d_in), "decode of class %s" % name
        out[name]["decode GiB/s"] = round(nf * fs * kd / ddt / (1 << 30), 1)
        del d_in, d_one
    return {"per_class": out, "class_generation_s": gen_s,
            "note": "compress: %d x %d B chunks per class, level %d, 5 steps after one warm-up; decode: %d frames of %d B of the class built by this codec, one call a step, 3 steps after "
                    "one warm-up, output compared with the input; each class is distinct_bytes of distinct data (generators of tests/_corpus.py in 1 MiB segments, the file classes as much as "
                    "the box holds) tiled on the device to the batch; zeros: all-zero; period1000: one random 1000-byte string repeated" % (n, cs, args.level, nf, fs)}


def libzstd_frames_decode(bc, args, torch, host, nf, fs):
    """BASELINE config 4's "pre-built zstd frames": the same slices compressed by UPSTREAM libzstd (level 3, one frame a slice, built on the
    host cores), decoded by the HIP decoder; whole output verified.  None when libzstd cannot be loaded."""
    import _oracle as O
    Z = O.libzstd()
    if not Z:
        return None
    L = O.lib(); vp = ctypes.c_void_p
    bound = int(Z.ZSTD_compressBound(fs)); stride = (bound + 255) // 256 * 256
    zb = np.empty(nf * stride, dtype=np.uint8); zs = np.zeros(nf, dtype=np.uint32)
    offs = np.arange(nf, dtype=np.uint64) * fs; sizes = np.full(nf, fs, dtype=np.uint32); foffs = np.arange(nf, dtype=np.uint64) * stride
    rc = L.zso_libzstdCompressBatch(zb.ctypes.data_as(vp), foffs.ctypes.data_as(vp), zs.ctypes.data_as(vp), host.ctypes.data_as(vp),
                                    offs.ctypes.data_as(vp), sizes.ctypes.data_as(vp), nf, 3, usable_cores())
    assert rc == 0
    d_frames = torch.from_numpy(zb).cuda(); d_src = torch.from_numpy(host[:nf * fs]).cuda()
    d_out = torch.empty(nf * fs, dtype=torch.uint8, device="cuda"); d_osz = torch.zeros(nf, dtype=torch.int32, device="cuda")
    step = lambda: bc.decompress_device(d_frames.data_ptr(), foffs, zs, d_out.data_ptr(), offs, sizes, d_osz.data_ptr())
    step(); torch.cuda.synchronize()
    k = max(2, args.steps // 4)
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bc.enable_timing(True)                              # where the time goes (outside the timed steps): the general kernel's share shows the frames the fast path leaves
    step(); torch.cuda.synchronize()
    kt = bc.kernel_times(); bc.enable_timing(False)
    assert (d_osz.cpu().numpy() == fs).all(), "a libzstd frame failed to decode"
    assert torch.equal(d_out, d_src), "decoded bytes differ from the input"
    return {"value": round(nf * fs * k / dt / (1 << 30), 3), "unit": "GiB/s", "frames": nf, "steps": k, "compressed_bytes": int(zs.astype(np.uint64).sum()),
            "kernels_ms_per_step": {kk: round(v[0] * 1e3, 4) for kk, v in kt.items()},
            "label": "frames built by upstream libzstd %d ZSTD_compress level 3 from the same %d B slices (BASELINE config 4: pre-built zstd frames); whole output verified" % (Z.ZSTD_versionNumber(), fs)}



def large_frames_decode(bc, args, torch, host, nf=2048, fs=1 << 20):
    """Frames of 16 blocks (1 MiB chunks of the same stream, this codec's encoder at level 3): a call of large frames reserves up to 16 block slots an
    item and stays on the decoder's fast path.  Whole output verified; not part of `value`."""
    n = min(nf, len(host) // fs)
    d_src = torch.from_numpy(host[:n * fs]).cuda()
    bound = int(bc.L.zsmi_compressBound(fs)); stride = (bound + 255) // 256 * 256
    d_frames = torch.empty(n * stride, dtype=torch.uint8, device="cuda"); d_fsz = torch.zeros(n, dtype=torch.int32, device="cuda")
    offs = np.arange(n, dtype=np.uint64) * fs; sizes = np.full(n, fs, dtype=np.uint32); foffs = np.arange(n, dtype=np.uint64) * stride
    bc.compress_device(d_src.data_ptr(), offs, sizes, d_frames.data_ptr(), foffs, d_fsz.data_ptr(), 3); torch.cuda.synchronize()
    fsz = d_fsz.cpu().numpy().astype(np.uint32)
    d_out = torch.empty(n * fs, dtype=torch.uint8, device="cuda"); d_osz = torch.zeros(n, dtype=torch.int32, device="cuda")
    step = lambda: bc.decompress_device(d_frames.data_ptr(), foffs, fsz, d_out.data_ptr(), offs, sizes, d_osz.data_ptr())
    step(); torch.cuda.synchronize()
    k = 3
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bc.enable_timing(True); step(); torch.cuda.synchronize(); kt = bc.kernel_times(); bc.enable_timing(False)
    assert (d_osz.cpu().numpy() == fs).all() and torch.equal(d_out, d_src), "a large frame failed to decode"
    return {"value": round(n * fs * k / dt / (1 << 30), 3), "unit": "GiB/s", "frames": n, "frame_bytes": fs, "blocks_per_frame": fs // 65536,
            "kernels_ms_per_step": {kk: round(v[0] * 1e3, 4) for kk, v in kt.items()},
            "label": "frames of 16 compressed blocks built by this codec from the same stream, one call; the general kernel's share shows frames that left the fast path"}


def io_inclusive_leg(torch, dist, rank, world, all_sizes, root_src, compress_shard, device, barrier, reps=3):
    """SURVEY 8e end to end, the job that starts and ends on rank 0: scatter the shards (one grouped batch of sends), every rank compresses
    and packs its shard, all-gather of the frame sizes, gather of the frames to rank 0 (one grouped batch of receives).
    compress_shard(shard tensor, chunk sizes of the shard) -> (frames packed back to back: uint8 tensor on `device`, frame sizes: np.uint32).
    Returns on every rank (best wall time of reps - 1 repetitions after one warm-up, seconds of this rank's compress + pack, frames on rank 0, offsets, sizes)."""
    from zstandard_amd.sharding import scatter_chunks, gather_frames
    best = None; mine = None; out_all = goffs = gsz = None
    for it in range(reps):
        barrier(); t1 = time.perf_counter()
        shard, (b0, e0) = scatter_chunks(root_src, all_sizes, root=0, device=device)
        t2 = time.perf_counter()
        packed, fsz = compress_shard(shard, all_sizes[b0:e0])
        t3 = time.perf_counter()
        out_all, goffs, gsz = gather_frames(packed, fsz, all_sizes, root=0)
        barrier(); dt = time.perf_counter() - t1
        if it:
            if best is None or dt < best:
                best, mine = dt, t3 - t2
    secs = [None] * world
    dist.all_gather_object(secs, (rank, float(mine), int(np.asarray(all_sizes[b0:e0], dtype=np.uint64).sum())))
    return best, secs, out_all, goffs, gsz


def decode_leg(bc, args, rank, world, distributed, barrier, torch, dist):
    """BASELINE config 4 shape: nf frames of ~32 KiB (level-3 output of this codec's encoder, built on the device), decoded
    per step; timed like the compress leg (barrier + synchronize on both sides, max over ranks); the whole output is compared
    with the input after the timed region.  Returns the "decode" object (rank 0) or None."""
    import _data as D, _oracle as O
    nf, fs = args.decode_frames, args.decode_frame_size
    host = D.zipf_log(nf * fs, seed_lo=0xDEC0DE + 7919 * rank, threads=min(32, os.cpu_count() or 1))
    d_src = torch.from_numpy(host).cuda()
    bound = int(bc.L.zsmi_compressBound(fs)); stride = (bound + 255) // 256 * 256
    d_frames = torch.empty(nf * stride, dtype=torch.uint8, device="cuda"); d_fsz = torch.zeros(nf, dtype=torch.int32, device="cuda")
    offs = np.arange(nf, dtype=np.uint64) * fs; sizes = np.full(nf, fs, dtype=np.uint32); foffs = np.arange(nf, dtype=np.uint64) * stride
    bc.compress_device(d_src.data_ptr(), offs, sizes, d_frames.data_ptr(), foffs, d_fsz.data_ptr(), 3)
    torch.cuda.synchronize()
    fsz = d_fsz.cpu().numpy().astype(np.uint32)
    assert (fsz < 0xFFFFFF88).all()
    d_out = torch.empty(nf * fs, dtype=torch.uint8, device="cuda"); d_osz = torch.zeros(nf, dtype=torch.int32, device="cuda")

    def step():
        bc.decompress_device(d_frames.data_ptr(), foffs, fsz, d_out.data_ptr(), offs, sizes, d_osz.data_ptr())
    for _ in range(args.warmup):
        step()
    barrier()
    bc.enable_timing(2)                     # events around the dominant kernel only (k_dec_execute)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kt = bc.kernel_times()
    bc.enable_timing(True)                  # every kernel: the same steps once more, outside the timed region
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    kt_all = bc.kernel_times()
    bc.enable_timing(False)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert (d_osz.cpu().numpy() == fs).all(), "a frame failed to decode"
    assert torch.equal(d_out, d_src), "decoded bytes differ from the input"
    if rank != 0:
        return None
    comp = int(fsz.astype(np.uint64).sum())
    name, (secs, launches) = max(kt.items(), key=lambda kv: kv[1][0])
    per_launch = (nf * fs + comp) * args.steps / launches
    roof = {"bound": "hbm", "kernel": name, "achieved": round(per_launch / (secs / launches) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(per_launch / (secs / launches) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None, "algorithmic_bytes_per_launch": int(per_launch),
            "avg_launch_ms": round(secs / launches * 1e3, 4), "kernels_ms_per_step": {k: round(v[0] / args.steps * 1e3, 4) for k, v in kt_all.items()}}
    try:                                            # HBM bytes per launch of the dominant kernel from the committed PMC passes, scaled to this launch size
        tjf = json.load(open(os.path.join(ROOT, "profiles", "r4_decode_traffic.json")))
        if tjf.get("kernel_source_sha256") == source_fingerprint() and name in tjf["kernels"]:
            roof["traffic"] = int(tjf["kernels"][name]["hbm_bytes"] * nf / tjf["kernels"][name]["frames_per_launch"])
        else:
            roof["traffic_note"] = "profiles/r4_decode_traffic.json was measured at other kernel sources (%s): not reported" % tjf.get("kernel_source_sha256")
    except Exception as e:
        roof["traffic_note"] = "no traffic file: %s" % type(e).__name__
    out = {"metric": f"GiB/s decompress (output bytes), frames of {fs} B", "value": round(nf * fs * world * args.steps / elapsed / (1 << 30), 3), "unit": "GiB/s",
           "frames_per_gpu_per_step": nf, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "verified": "whole output equals the input (torch.equal) after the timed region",
           "roofline": roof}
    if not args.no_extras:
        zl = libzstd_frames_decode(bc, args, torch, host, nf, fs)
        if zl:
            out["libzstd_frames"] = zl
        try:
            out["large_frames"] = large_frames_decode(bc, args, torch, host)
        except Exception as e:                                  # an extra: say what happened and go on
            out["large_frames"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    if not args.no_cpu_baseline:
        cores = usable_cores(); m = min(nf, 4096)
        fr = d_frames[:m * stride].cpu().numpy(); L = O.lib(); vp = ctypes.c_void_p
        dst = np.empty(m * fs, dtype=np.uint8); dsz = np.zeros(m, dtype=np.uint32)
        best = None
        for _ in range(2):
            t1 = time.perf_counter()
            L.zso_decompressBatch(dst.ctypes.data_as(vp), offs[:m].ctypes.data_as(vp), sizes[:m].ctypes.data_as(vp), dsz.ctypes.data_as(vp), fr.ctypes.data_as(vp),
                                  foffs[:m].ctypes.data_as(vp), fsz[:m].ctypes.data_as(vp), m, cores)
            d = time.perf_counter() - t1; best = d if best is None else min(best, d)
        assert (dst == host[:m * fs]).all()
        out["cpu_baseline"] = {"value": round(m * fs / best / (1 << 30), 3), "unit": "GiB/s", "cores": cores, "kind": "port",
                               "sample": f"{m} of the same frames, oracle D (C restatement of the reference's C# decoder), one thread per core"}
        Z = O.libzstd()
        if Z and not args.no_extras:
            t1 = time.perf_counter()
            rc = L.zso_libzstdDecompressBatch(dst.ctypes.data_as(vp), offs[:m].ctypes.data_as(vp), sizes[:m].ctypes.data_as(vp), dsz.ctypes.data_as(vp), fr.ctypes.data_as(vp),
                                              foffs[:m].ctypes.data_as(vp), fsz[:m].ctypes.data_as(vp), m, cores)
            d = time.perf_counter() - t1
            assert rc == 0 and (dst == host[:m * fs]).all()
            out["libzstd_yardstick"] = {"value": round(m * fs / d / (1 << 30), 3), "unit": "GiB/s", "cores": cores,
                                        "label": "upstream libzstd %d ZSTD_decompress on the same frames, NOT the reference" % Z.ZSTD_versionNumber()}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--chunks", type=int, default=4096, help="64 KiB chunks per GPU per step")
    ap.add_argument("--chunk-size", type=int, default=65536)
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--corpus", type=str, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-io", action="store_true", help="with --force-dist on one GPU: also time the job that starts and ends on rank 0: scatter shards, compress, pack, "
                    "gather frames (SURVEY 8e steps 1-4); with more than one rank that leg always runs; reported as io_inclusive, never as value")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even for one rank: exercises the N > 1 code path on a 1-GPU box")
    ap.add_argument("--decode-frames", type=int, default=57344, help="frames of --decode-frame-size bytes decoded per step of the decode leg (0: no decode leg)")
    ap.add_argument("--decode-frame-size", type=int, default=32768)
    ap.add_argument("--no-extras", action="store_true", help="skip ratio_by_class and libzstd_yardstick")
    ap.add_argument("--class-bytes", type=int, default=64 << 20, help="distinct bytes per data class of throughput_by_class")
    ap.add_argument("--dry-run-gloo", action="store_true", help="CPU rehearsal of the launch path: gloo process group, no GPU work; prints the ranks that ran")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    class_data, class_gen_s = (None, 0.0)
    if rank == 0 and not args.no_extras and not args.dry_run_gloo:
        class_data, class_gen_s = class_corpus(args)          # (a pool of forked processes: before torch / HIP are initialised)
    import torch
    assert world == args.gpus or (args.gpus == 1 and args.force_dist), f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU"
    distributed = world > 1 or args.force_dist
    if args.dry_run_gloo:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank)); os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group(backend="gloo")
        ranks = [None] * world
        dist.all_gather_object(ranks, {"rank": rank, "pid": os.getpid()})
        if rank == 0:
            print(json.dumps({"dry_run": "gloo", "n_gpus": world, "ranks": ranks}))
        dist.destroy_process_group()
        return
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank)); os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    from zstandard_amd import BatchCodec
    import _data as D

    cs, n = args.chunk_size, args.chunks
    nbytes = cs * n
    t0 = time.time()
    if args.corpus:
        host = load_corpus(args.corpus, nbytes, rank); data_label = f"corpus:{os.path.basename(args.corpus)}"
    else:
        host = D.zipf_log(nbytes, seed_lo=0x5EED + 1000003 * rank, threads=min(32, os.cpu_count() or 1)); data_label = "synthetic"
    gen_s = time.time() - t0

    stream = torch.cuda.current_stream()
    bc = BatchCodec(local_rank, stream.cuda_stream)
    d_src = torch.from_numpy(host).cuda()
    offs = np.arange(n, dtype=np.uint64) * cs
    sizes = np.full(n, cs, dtype=np.uint32)
    bound = int(bc.L.zsmi_compressBound(cs)); stride = (bound + 255) // 256 * 256
    d_dst = torch.empty(n * stride, dtype=torch.uint8, device="cuda")
    d_sizes = torch.zeros(n, dtype=torch.int32, device="cuda")
    doffs = np.arange(n, dtype=np.uint64) * stride

    def step():
        bc.compress_device(d_src.data_ptr(), offs, sizes, d_dst.data_ptr(), doffs, d_sizes.data_ptr(), args.level)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    bc.enable_timing(2)                     # HIP events around the dominant kernel only (k_lz_walk): the roofline's launch duration comes from the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    issued = time.perf_counter() - t0           # the host has handed over every step (the calls do not wait for the device)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    ktimes = bc.kernel_times()
    # every kernel's time: the same steps once more with events around every launch (ten event records a step cost ~2 % of it: not in the timed region)
    bc.enable_timing(True)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ktimes_all = bc.kernel_times()
    bc.enable_timing(False)
    # sustained clocks: the same step back to back for >= 2 s, outside the headline's timed region
    sustained = None
    if not args.no_extras:
        t1 = time.perf_counter(); k = 0
        while time.perf_counter() - t1 < 2.0:
            for _ in range(50):
                step()
            torch.cuda.synchronize(); k += 50
        dt = time.perf_counter() - t1
        sustained = {"value": round(nbytes * k / dt / (1 << 30), 3), "unit": "GiB/s per GPU", "steps": k, "seconds": round(dt, 2),
                     "note": "the headline's step looped for >= 2 s (synchronised every 50 steps), this rank only"}

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    io_inclusive = None
    if distributed and (world > 1 or args.with_io):
        # the whole job from rank 0's input to rank 0's frames; inputs and outputs stay in HBM (no PCIe leg).  Default whenever there is
        # more than one rank: this is the path north_star names (RCCL scatter of input shards, gather of frames)
        all_sizes = np.full(n * world, cs, dtype=np.uint32)
        g_src = torch.cat([d_src] * world) if rank == 0 else None          # root holds world x the per-GPU batch
        d_packed = torch.empty(n * stride, dtype=torch.uint8, device="cuda")
        d_poffs = torch.zeros(n + 1, dtype=torch.int64, device="cuda")

        def compress_shard(shard, shard_sizes):
            bc.compress_device(shard.data_ptr(), offs, sizes, d_dst.data_ptr(), doffs, d_sizes.data_ptr(), args.level)
            bc.pack_device(d_dst.data_ptr(), doffs, d_sizes.data_ptr(), n, d_packed.data_ptr(), d_poffs.data_ptr())
            torch.cuda.synchronize()
            return d_packed, d_sizes.cpu().numpy().astype(np.uint32)
        try:
            best, secs, out_all, goffs, gsz = io_inclusive_leg(torch, dist, rank, world, all_sizes, g_src, compress_shard, "cuda", barrier)
        except Exception as e:                                             # the headline does not depend on this leg: say what happened and go on
            best = None
            io_inclusive = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        if rank == 0 and best is not None:
            import _oracle as O
            hostf = out_all.cpu().numpy(); ok = True; checked = 0
            for r in range(world):                                             # frames of every rank's shard decode to the chunks they were made from
                for i in (r * n, r * n + n // 2, r * n + n - 1):
                    f = hostf[int(goffs[i]):int(goffs[i]) + int(gsz[i])].tobytes()
                    ok = ok and O.decompress(f, cs) == host[(i % n) * cs:(i % n + 1) * cs].tobytes(); checked += 1
            io_inclusive = {"value": round(n * world * cs / best / (1 << 30), 3), "unit": "GiB/s", "ms": round(best * 1e3, 3), "ranks": world,
                            "backend": dist.get_backend(), "frames_decode": ok, "frames_checked": checked,
                            "per_rank_compress_pack_GiB/s": [round(b / t / (1 << 30), 1) for (_, t, b) in sorted(secs)],
                            "note": "rank 0 input -> scatter (grouped send/recv) -> compress -> pack -> all-gather sizes -> gather frames to rank 0; device memory only; "
                                    "a sample of every rank's frames decoded on rank 0 with oracle D after the timed region"}

    decode = decode_leg(bc, args, rank, world, distributed, barrier, torch, dist if distributed else None) if args.decode_frames else None

    csz = d_sizes.cpu().numpy().astype(np.uint32)
    assert (csz < 0xFFFFFF88).all(), "a chunk failed to compress"
    comp_bytes = int(csz.astype(np.uint64).sum())
    tot = torch.tensor([float(nbytes), float(comp_bytes)], dtype=torch.float64, device="cuda")
    if distributed:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    total_in, total_out = float(tot[0].item()), float(tot[1].item())

    if rank == 0:
        value = total_in * args.steps / elapsed / (1 << 30)
        # roofline of the dominant kernel: algorithmic bytes per launch (S + C of the chunks one launch handles,
        # SURVEY.md 8d) over its average launch duration, HIP events on the launch stream
        dom = max(ktimes.items(), key=lambda kv: kv[1][0]) if ktimes else None
        roofline = None
        if dom:
            name, (secs, launches) = dom
            per_launch_bytes = (nbytes + comp_bytes) * args.steps / launches
            avg = secs / launches
            achieved = per_launch_bytes / avg / 1e9
            # HBM bytes per launch of that kernel from the committed PMC passes (FETCH_SIZE + WRITE_SIZE), scaled to this launch size.
            # The file names the kernel sources it was measured at: other sources -> null.
            traffic = None; traffic_note = None
            try:
                tjf = json.load(open(os.path.join(ROOT, "profiles", "r4_traffic.json")))
                if tjf.get("kernel_source_sha256") != source_fingerprint():
                    traffic_note = "profiles/r4_traffic.json was measured at other kernel sources (%s): not reported" % tjf.get("kernel_source_sha256")
                else:
                    tj = tjf["kernels"].get(name)
                    if tj:
                        traffic = int(tj["hbm_bytes"] * (n * args.steps / launches) / tj["blocks_per_launch"])
            except Exception as e:
                traffic_note = "no traffic file: %s" % type(e).__name__
            roofline = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "algorithmic_bytes_per_launch": int(per_launch_bytes),
                        "avg_launch_ms": round(avg * 1e3, 4),
                        "kernels_ms_per_step": {k: round(v[0] / args.steps * 1e3, 4) for k, v in ktimes_all.items()},
                        "kernels_ms_per_step_note": "second pass of the same steps with events around every launch; avg_launch_ms is from the timed region"}
            if traffic_note:
                roofline["traffic_note"] = traffic_note
        # exactness spot check of the timed output + ratio yardstick
        import _oracle as O
        sample = list(range(0, n, max(1, n // 16)))[:16]
        dst_host = d_dst.cpu().numpy()
        for i in sample:
            f = dst_host[int(doffs[i]):int(doffs[i]) + int(csz[i])].tobytes()
            c = host[i * cs:(i + 1) * cs].tobytes()
            assert O.decompress(f, len(c)) == c, "timed output must decode under the restated reference decoder"
        ratio = total_in / total_out
        ratio_vs_zstd = None
        if O.libzstd():
            z = sum(len(O.zstd_compress(host[i * cs:(i + 1) * cs].tobytes(), args.level)) for i in sample)
            e = int(csz[sample].astype(np.uint64).sum())
            ratio_vs_zstd = round(z / e, 4)          # > 1: smaller than libzstd ; 0.99 = 1 % larger
        cpu = None
        if not args.no_cpu_baseline:
            cores = usable_cores()
            m = min(n, max(256, 32 * cores))                      # bounded sample of the same workload (whole default batch)
            dt = None
            for _ in range(2):                                    # best of two: the first run pays the threads' workspace page faults
                t1 = time.perf_counter()
                O.compress_batch(host, offs[:m], sizes[:m], args.level, cores)
                d = time.perf_counter() - t1
                dt = d if dt is None else min(dt, d)
            cpu = {"value": round(m * cs / dt / (1 << 30), 4), "unit": "GiB/s", "cores": cores, "kind": "port",
                   "sample": f"{m} x {cs} B chunks of the same batch, oracle E (scalar statement of the HIP encoder), one thread per core"}
        # BASELINE.json's metric at its own configuration; other levels / chunk sizes (development runs) are labelled as what they are
        metric = "GiB/s compress @ level 3, 64 KiB chunks" if (args.level == 3 and cs == 65536) else f"GiB/s compress @ level {args.level}, {cs // 1024} KiB chunks"
        out = {"metric": metric, "value": round(value, 3), "unit": "GiB/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "host_issue_ms_per_step": round(issued / args.steps * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": data_label, "ranks": world,
               "config": {"workload": f"{n} independent {cs} B chunks per GPU per step, level {args.level}, BASELINE config[1] shape "
                                      f"(Silesia unavailable offline -> Zipf-token log stream, SURVEY 8d)", "chunks_per_gpu": n,
                          "chunk_bytes": cs, "level": args.level, "parallelism": f"chunks sharded over {world} GPU(s), no collective in the data path"},
               "ratio": round(ratio, 4), "ratio_vs_libzstd_same_level": ratio_vs_zstd,
               "hbm_read_roofline_frac": round(total_in * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 5),
               "roofline": roofline, "cpu_baseline": cpu, "datagen_s": round(gen_s, 2),
               "library": bc.L.zsmi_versionString().decode()}
        if sustained:
            out["sustained"] = sustained
        if decode:
            out["decode"] = decode
        if not args.no_extras:
            out["throughput_by_class"] = throughput_by_class(bc, args, torch, class_data, class_gen_s)
        Z = O.libzstd()
        if Z and not args.no_extras:
            # the ratio contract per data class (tests/_corpus.py), HIP encoder vs upstream libzstd at this level and chunk size
            import _corpus as C
            rb = {}
            for cname, cdata in C.corpus(1 << 20).items():
                buf = np.frombuffer(cdata, dtype=np.uint8)
                co = np.arange(0, len(cdata), cs, dtype=np.uint64); csizes = np.minimum(len(cdata) - co, cs).astype(np.uint32)
                _, _, cz = bc.compress_host(buf, co, csizes, args.level)
                zz = sum(len(O.zstd_compress(cdata[int(o):int(o) + int(k)], args.level)) for o, k in zip(co, csizes))
                rb[cname] = round(int(cz.astype(np.uint64).sum()) / zz, 4)
            out["ratio_by_class"] = {"ours_over_libzstd_compressed_size": rb, "worst": max(rb.values()), "tolerance": 1.01,
                                     "note": "1 MiB per class, %d B chunks, level %d; < 1: smaller than libzstd" % (cs, args.level)}
            if not args.no_cpu_baseline:
                cores = usable_cores(); m = min(n, max(256, 32 * cores))
                zb = np.empty(m * stride, dtype=np.uint8); zs = np.zeros(m, dtype=np.uint32); vp = ctypes.c_void_p
                t1 = time.perf_counter()
                rc = O.lib().zso_libzstdCompressBatch(zb.ctypes.data_as(vp), doffs[:m].ctypes.data_as(vp), zs.ctypes.data_as(vp), host.ctypes.data_as(vp),
                                                      offs[:m].ctypes.data_as(vp), sizes[:m].ctypes.data_as(vp), m, args.level, cores)
                d = time.perf_counter() - t1
                assert rc == 0
                out["libzstd_yardstick"] = {"value": round(m * cs / d / (1 << 30), 3), "unit": "GiB/s", "cores": cores,
                                            "label": "upstream libzstd %d ZSTD_compress level %d on %d chunks of the same batch, NOT the reference (which has no encoder)" % (Z.ZSTD_versionNumber(), args.level, m)}
        if io_inclusive:
            out["io_inclusive"] = io_inclusive
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    bc.close()


if __name__ == "__main__":
    main()
