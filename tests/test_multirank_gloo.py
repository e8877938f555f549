"""N > 1 path on CPU: two gloo ranks partition a batch (zstandard_amd.sharding), each handles its shard, sizes are
all-gathered, rank 0 rebuilds the global frame order.  The per-shard codec call is the oracle here (no GPU)."""
import os, socket, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch, torch.distributed as dist
    import _oracle as O, _data as D
    from zstandard_amd.sharding import partition_chunks, global_frame_offsets
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = D.zipf_log(1 << 20)
    rng = np.random.default_rng(3)
    sizes = rng.integers(1000, 70000, 24).astype(np.uint32)
    offs = np.zeros(len(sizes), dtype=np.uint64); offs[1:] = np.cumsum(sizes.astype(np.uint64))[:-1]
    b, e = partition_chunks(sizes, world)[rank]
    arena, do, dsz = O.compress_batch(data, offs[b:e], sizes[b:e], 3, 1)
    mine = torch.zeros(len(sizes), dtype=torch.int64); mine[b:e] = torch.from_numpy(dsz.astype(np.int64))
    dist.all_reduce(mine)                                   # every chunk belongs to exactly one rank
    gathered = [None] * world
    dist.all_gather_object(gathered, (b, e, [arena[int(do[i]):int(do[i]) + int(dsz[i])].tobytes() for i in range(e - b)]))
    dist.barrier()
    if rank == 0:
        frames = [None] * len(sizes)
        for (bb, ee, fl) in gathered:
            frames[bb:ee] = fl
        ok = all(O.decompress(frames[i], int(sizes[i])) == data[int(offs[i]):int(offs[i]) + int(sizes[i])].tobytes() for i in range(len(sizes)))
        offs_g, total = global_frame_offsets([np.array([len(f) for f in fl]) for (_, _, fl) in gathered])
        q.put((ok, [int(x) for x in mine.tolist()] == [len(f) for f in frames], total == sum(len(f) for f in frames)))
    dist.destroy_process_group()


def test_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = q.get(timeout=120)
    for p in ps:
        p.join(timeout=60)
    assert res == (True, True, True)


def _io_sizes(kind):
    rng = np.random.default_rng(9)
    if kind == "skewed8":
        # what 8 ranks make of it: a chunk that is a whole shard and more (the ranks behind it get EMPTY shards), a rank whose shard is ONE chunk,
        # runs of empty and tiny chunks, and ordinary shards - all seven peers of the root transfer at once, or not at all
        return np.concatenate([[131072 * 3], [0, 0, 7], rng.integers(20000, 50000, 6), [131072], [1, 2, 3], rng.integers(0, 30000, 10)]).astype(np.uint32)
    return np.concatenate([rng.integers(0, 70000, 21), [0, 131072, 5]]).astype(np.uint32)


def _worker_io(rank, world, port, q, kind="plain"):
    """scatter from rank 0 -> per-shard codec call (oracle, CPU) -> gather to rank 0: the §8e data path end to end"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch, torch.distributed as dist
    import _oracle as O, _data as D
    from zstandard_amd.sharding import scatter_chunks, gather_frames
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = _io_sizes(kind)                                                                      # known to every rank
    total = int(sizes.astype(np.uint64).sum())
    data = torch.from_numpy(D.zipf_log(total + 16)[:total].copy()) if rank == 0 else None        # only the root has the input
    shard, (b, e) = scatter_chunks(data, sizes, root=0)
    loc = sizes[b:e]
    offs = np.zeros(len(loc), dtype=np.uint64)
    if len(loc) > 1:
        offs[1:] = np.cumsum(loc.astype(np.uint64))[:-1]
    src = shard.numpy() if shard.numel() else np.zeros(1, dtype=np.uint8)
    if len(loc):
        arena, do, dsz = O.compress_batch(src, offs, loc, 3, 1)
        packed = np.concatenate([arena[int(do[i]):int(do[i]) + int(dsz[i])] for i in range(len(loc))])
    else:
        dsz = np.zeros(0, dtype=np.uint32); packed = np.zeros(0, dtype=np.uint8)
    out, goffs, gsizes = gather_frames(torch.from_numpy(packed.copy()), dsz, sizes, root=0)
    dist.barrier()
    if rank == 0:
        full = data.numpy(); pos = 0; ok = len(gsizes) == len(sizes)
        for i in range(len(sizes)):
            f = out[int(goffs[i]):int(goffs[i]) + int(gsizes[i])].numpy().tobytes()
            ok = ok and O.decompress(f, max(int(sizes[i]), 1)) == full[pos:pos + int(sizes[i])].tobytes()
            pos += int(sizes[i])
        q.put(ok)
    else:
        assert out is None and len(gsizes) == len(sizes)
    dist.destroy_process_group()


def test_scatter_compress_gather_gloo_world8():
    """the configuration the driver's scaling run uses: 8 ranks, the root sending to / receiving from 7 peers in one grouped batch;
    shards that are empty, a shard of one chunk"""
    import torch.multiprocessing as mp
    from zstandard_amd.sharding import partition_chunks
    parts = partition_chunks(_io_sizes("skewed8"), 8)
    counts = [e - b for b, e in parts]
    assert 0 in counts and 1 in counts and max(counts) > 3, counts          # the shapes this test is for
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_io, args=(r, 8, port, q, "skewed8")) for r in range(8)]
    for p in ps:
        p.start()
    res = q.get(timeout=300)
    for p in ps:
        p.join(timeout=120)
    assert res is True and all(p.exitcode == 0 for p in ps)


@pytest.mark.parametrize("world", [2, 3])
def test_scatter_compress_gather_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_io, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = q.get(timeout=180)
    for p in ps:
        p.join(timeout=60)
    assert res is True and all(p.exitcode == 0 for p in ps)


def test_partition_properties():
    from zstandard_amd.sharding import partition_chunks
    rng = np.random.default_rng(1)
    for world in (1, 2, 4, 8):
        for n in (0, 1, 7, 100):
            sizes = rng.integers(0, 131072, n)
            parts = partition_chunks(sizes, world)
            assert len(parts) == world and parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
    sizes = np.full(16384, 131072)
    parts = partition_chunks(sizes, 8)
    assert all(e - b == 2048 for b, e in parts)        # 16 384 chunks / GPU ... here 2048 each of 16 384
