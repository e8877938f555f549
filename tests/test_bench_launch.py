"""bench.py --gpus N without a rank environment starts N ranks itself (CPU rehearsal: gloo, no GPU work)."""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_spawns_that_many_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-gloo"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and sorted(x["rank"] for x in out["ranks"]) == [0, 1]
    assert len({x["pid"] for x in out["ranks"]}) == 2


def test_world_size_must_match_gpus_flag():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run-gloo"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def _io_worker(rank, world, port, q):
    """bench.py's io_inclusive leg (scatter -> compress -> pack -> all-gather sizes -> gather) under gloo, the oracle standing in for the GPU codec"""
    import time
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import bench, _oracle as O, _data as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, cs = 6, 65536                                         # per rank, as bench.py lays the job out: world x n chunks on the root
    all_sizes = np.full(n * world, cs, dtype=np.uint32)
    host = D.zipf_log(n * cs, seed_lo=77)
    root_src = torch.from_numpy(np.tile(host, world).copy()) if rank == 0 else None

    def compress_shard(shard, shard_sizes):
        offs = np.arange(len(shard_sizes), dtype=np.uint64) * cs
        arena, do, dsz = O.compress_batch(shard.numpy(), offs, np.asarray(shard_sizes, dtype=np.uint32), 3, 1)
        packed = np.concatenate([arena[int(do[i]):int(do[i]) + int(dsz[i])] for i in range(len(dsz))])
        return torch.from_numpy(packed.copy()), dsz

    best, secs, out_all, goffs, gsz = bench.io_inclusive_leg(torch, dist, rank, world, all_sizes, root_src, compress_shard, None, dist.barrier, reps=2)
    if rank == 0:
        ok = best > 0 and sorted(r for r, _, _ in secs) == list(range(world)) and all(b == n * cs for _, _, b in secs) and len(gsz) == n * world
        for i in range(n * world):
            f = out_all[int(goffs[i]):int(goffs[i]) + int(gsz[i])].numpy().tobytes()
            ok = ok and O.decompress(f, cs) == host[(i % n) * cs:(i % n + 1) * cs].tobytes()
        q.put(ok)
    dist.destroy_process_group()


def test_io_inclusive_leg_under_gloo():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_io_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = q.get(timeout=180)
    for p in ps:
        p.join(timeout=60)
    assert res is True and all(p.exitcode == 0 for p in ps)
