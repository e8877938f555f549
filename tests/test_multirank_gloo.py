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
