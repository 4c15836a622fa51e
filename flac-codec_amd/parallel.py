"""Multi-GPU sharding of the encode hot path.

A FLAC frame depends only on its own samples, the options and its frame number
(/root/reference/src/encode.rs:2284-2294), so a stream shards by CONTIGUOUS FRAME RANGES, one
process per GPU, with no data-path collective.  What crosses shards is bookkeeping only:
SEEKTABLE byte offsets (a prefix sum of frame sizes, encode.rs:1999-2003), STREAMINFO min/max
frame size (encode.rs:2414-2436) and the frame/sample totals.  One all-gather of four integers
per rank (RCCL over xGMI when the backend is "nccl") carries all of it.  The stream MD5 is a
serial chain over the PCM and stays with whoever owns the input (encode.rs:571, 2100).
"""
import ctypes as C

from . import _lib

COUNTER_FIELDS = ("frames", "bytes", "min_frame", "max_frame")


def shard_range(total_frames, world, rank):
    """Contiguous frame range [lo, hi) of `rank`: [k*F/G, (k+1)*F/G)."""
    return total_frames * rank // world, total_frames * (rank + 1) // world


def local_counters(analyzer, n_frames):
    """{frames, bytes, min_frame, max_frame} of the frames this rank packed last."""
    off = (C.c_uint64 * (n_frames + 1))()
    total = C.c_uint64(0)
    rc = _lib.lib().flacgpu_fetch_frames(analyzer._h, None, 0, off, C.byref(total))
    if rc not in (0, -5):
        raise RuntimeError(f"flacgpu_fetch_frames: {rc}")
    sizes = [off[i + 1] - off[i] for i in range(n_frames)]
    return [n_frames, int(total.value), min(sizes), max(sizes)]


def all_gather_counters(local, dist=None, device=None):
    """All-gather the 4-integer counter record of every rank; returns a list per rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [list(local)]
    import torch

    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    mine = torch.tensor(local, dtype=torch.int64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [[int(v) for v in t.tolist()] for t in out]


def merge_counters(per_rank):
    """Stream-level bookkeeping from the per-shard records: byte offset of every shard's first
    frame (exclusive prefix sum), totals and STREAMINFO min/max frame size."""
    offsets, acc = [], 0
    for c in per_rank:
        offsets.append(acc)
        acc += c[1]
    return {
        "shard_byte_offsets": offsets,
        "total_frames": sum(c[0] for c in per_rank),
        "total_bytes": acc,
        "min_frame": min(c[2] for c in per_rank),
        "max_frame": max(c[3] for c in per_rank),
    }


def gather_shard_counters(analyzer, n_frames, dist=None):
    per_rank = all_gather_counters(local_counters(analyzer, n_frames), dist)
    return merge_counters(per_rank)
