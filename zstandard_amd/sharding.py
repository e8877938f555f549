"""Chunk partitioning for multi-GPU batches: chunks are independent frames, so a batch is cut
in contiguous chunk-index ranges balanced by uncompressed bytes, one range per rank; the codec itself needs no
collective.  When a job starts and ends on one rank, the input shards are scattered and the frames gathered with
grouped point-to-point transfers (RCCL send/recv over xGMI on GPUs, gloo on CPUs) and one all-gather of sizes."""
import numpy as np


def partition_chunks(sizes, world: int):
    """-> list of (begin, end) chunk-index ranges, one per rank, contiguous, balanced by bytes"""
    sizes = np.asarray(sizes, dtype=np.uint64)
    n = len(sizes)
    if world <= 1 or n == 0:
        return [(0, n)] + [(n, n)] * (max(world, 1) - 1)
    cum = np.concatenate([[0], np.cumsum(sizes)])
    total = int(cum[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r // world
        cuts.append(int(np.searchsorted(cum, target, side="left")))
    cuts.append(n)
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def global_frame_offsets(all_sizes):
    """all_sizes: per-rank arrays of frame sizes in rank order -> (offsets of every frame in the gathered stream, total)"""
    flat = np.concatenate([np.asarray(s, dtype=np.uint64) for s in all_sizes]) if len(all_sizes) else np.zeros(0, np.uint64)
    offs = np.zeros(len(flat), dtype=np.uint64)
    if len(flat) > 1:
        offs[1:] = np.cumsum(flat)[:-1]
    return offs, int(flat.sum())


# ---- collectives around the codec (torch.distributed; backend "nccl" = RCCL on GPUs, "gloo" in the CPU tests) ----
def _byte_ranges(sizes, world):
    parts = partition_chunks(sizes, world)
    cum = np.concatenate([[0], np.cumsum(np.asarray(sizes, dtype=np.uint64))]).astype(np.uint64)
    return parts, [(int(cum[b]), int(cum[e])) for b, e in parts]


def scatter_chunks(data, sizes, root=0, device=None, group=None):
    """The root rank holds `data` (uint8 tensor: the chunks back to back, in order); every rank receives the bytes of
    its shard (one grouped batch of sends on the root, one receive elsewhere).
    `sizes` (all chunk sizes) is known to every rank.  -> (shard tensor, (first chunk, end chunk))"""
    import torch, torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    parts, ranges = _byte_ranges(sizes, world)
    lo, hi = ranges[rank]
    if rank == root:
        ops = [dist.P2POp(dist.isend, data[ranges[r][0]:ranges[r][1]], r, group) for r in range(world) if r != root and ranges[r][1] > ranges[r][0]]
        shard = data[lo:hi]
    else:
        shard = torch.empty(hi - lo, dtype=torch.uint8, device=device)
        ops = [dist.P2POp(dist.irecv, shard, root, group)] if hi > lo else []
    for req in (dist.batch_isend_irecv(ops) if ops else []):
        req.wait()
    return shard, parts[rank]


def gather_frames(packed, frame_sizes, sizes, root=0, group=None):
    """Every rank holds its frames packed back to back (`packed`, uint8 tensor) and their sizes (`frame_sizes`, this
    rank's chunks in order).  Sizes are all-gathered (step 3), then the frames travel to the root in one grouped batch
    of receives (step 4).  -> on the root: (all frames back to back in chunk order, offsets uint64, sizes uint32);
    elsewhere (None, offsets, sizes) -- every rank learns the global layout."""
    import torch, torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    parts = partition_chunks(sizes, world)
    counts = [e - b for b, e in parts]
    width = max(max(counts), 1)
    mine = torch.zeros(width, dtype=torch.int64, device=packed.device)
    if counts[rank]:
        mine[:counts[rank]] = torch.as_tensor(np.asarray(frame_sizes, dtype=np.int64), device=packed.device)
    table = [torch.zeros(width, dtype=torch.int64, device=packed.device) for _ in range(world)]
    dist.all_gather(table, mine, group=group)
    per_rank = [t[:c].cpu().numpy().astype(np.uint64) for t, c in zip(table, counts)]
    offsets, total = global_frame_offsets(per_rank)
    all_sizes = np.concatenate(per_rank).astype(np.uint32) if per_rank else np.zeros(0, np.uint32)
    bytes_of = [int(s.sum()) for s in per_rank]
    starts = np.concatenate([[0], np.cumsum(bytes_of)]).astype(np.uint64)
    out = None
    if rank == root:
        out = torch.empty(total, dtype=torch.uint8, device=packed.device)
        out[int(starts[root]):int(starts[root]) + bytes_of[root]] = packed[:bytes_of[root]]
        ops = [dist.P2POp(dist.irecv, out[int(starts[r]):int(starts[r]) + bytes_of[r]], r, group) for r in range(world) if r != root and bytes_of[r]]
    else:
        ops = [dist.P2POp(dist.isend, packed[:bytes_of[rank]], root, group)] if bytes_of[rank] else []
    for req in (dist.batch_isend_irecv(ops) if ops else []):
        req.wait()
    return out, offsets, all_sizes
