"""Chunk partitioning for multi-GPU batches: chunks are independent frames, so a batch is cut
in contiguous chunk-index ranges balanced by uncompressed bytes, one range per rank; no data-path collective.
Only sizes are exchanged (all-gather) to place frames in a global order."""
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
