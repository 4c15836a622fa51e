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
import ctypes as C_

from . import _lib

COUNTER_FIELDS = ("frames", "bytes", "min_frame", "max_frame")


def shard_range(total_frames, world, rank):
    """Contiguous frame range [lo, hi) of `rank`: [k*F/G, (k+1)*F/G)."""
    return total_frames * rank // world, total_frames * (rank + 1) // world


def local_counters(analyzer, n_frames=None):
    """{frames, bytes, min_frame, max_frame} of the frames this rank packed last (the offsets array is
    sized by the batch the context holds, whatever the caller believes)."""
    held = getattr(analyzer, "last_frames", 0) or n_frames
    if n_frames is not None and held != n_frames:
        raise ValueError(f"the context holds a batch of {held} frames, not {n_frames}")
    n_frames = held
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


def merge_counters(per_rank, dist=None):
    """Stream-level bookkeeping from the per-shard records: byte offset of every shard's first
    frame (exclusive prefix sum), totals and STREAMINFO min/max frame size (ranks without frames -- more ranks than
    frames -- take no part in min / max).  `ranks_seen` is the length of the all-gather's result and `backend` the
    collective backend that produced it: a scaling record carrying them proves that N ranks met over RCCL."""
    offsets, acc = [], 0
    for c in per_rank:
        offsets.append(acc)
        acc += c[1]
    busy = [c for c in per_rank if c[0]]
    backend = dist.get_backend() if dist is not None and dist.is_initialized() else None
    return {
        "shard_byte_offsets": offsets,
        "total_frames": sum(c[0] for c in per_rank),
        "total_bytes": acc,
        "min_frame": min(c[2] for c in busy) if busy else 0,
        "max_frame": max(c[3] for c in busy) if busy else 0,
        "frames_per_rank": [c[0] for c in per_rank],
        "ranks_seen": len(per_rank),
        "backend": backend,
    }


def gather_shard_counters(analyzer, n_frames, dist=None):
    per_rank = all_gather_counters(local_counters(analyzer, n_frames), dist)
    return merge_counters(per_rank, dist)


def gather_exact(payload, lengths, dist, dtype):
    """Rank 0 receives every rank's `payload` (a 1-D tensor of lengths[rank] elements) at its EXACT length -- point to
    point, rank by rank: no padding to the longest shard (a gather of max-sized buffers moved world x max bytes and
    allocated as much on the owner).  Returns the list of tensors on rank 0, None elsewhere."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    dev = payload.device
    if rank != 0:
        if lengths[rank]:
            dist.send(payload, dst=0)
        return None
    out = [payload]
    for r in range(1, world):
        buf = torch.empty(lengths[r], dtype=dtype, device=dev)
        if lengths[r]:
            dist.recv(buf, src=r)
        out.append(buf)
    return out


def encode_stream_sharded(pcm, options, sample_rate, bits_per_sample, channels, dist=None, device=-1,
                          batch_frames=1024):
    """ONE stream encoded by every rank of `dist` together.

    The stream's blocks are cut into contiguous frame ranges (`shard_range`), each rank encodes its
    range on its GPU with the frame numbers the whole stream gives them; what crosses ranks is
      1. one all-gather of {frames, bytes, min_frame, max_frame} per rank (`merge_counters`:
         shard byte offsets, totals, STREAMINFO min/max frame size; encode.rs:2414-2436),
      2. one gather of the per-frame sizes (4 bytes a frame) to rank 0, from which the owner rebuilds
         the seek points exactly as the single writer would (encode.rs:1999-2003),
      3. the gather of the finished frame bytes to rank 0 (the only data movement; compressed).
    Rank 0 owns the input: it computes the stream MD5 (a serial chain over the whole PCM that cannot be
    merged from partial states, encode.rs:571, 2100), builds the metadata with flacenc_stream_header
    and returns the .flac bytes -- identical to what one FlacSampleWriter produces.  Other ranks
    return None.  `pcm`: interleaved int32 samples of the WHOLE stream (every rank reads its own
    range of it); `options`: flac_codec_amd.encode.Options.
    """
    import hashlib

    import numpy as np

    from . import encode as E
    from .gpu import GpuAnalyzer

    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    co = options._c_options()
    B, C = co.block_size, channels
    total_pcm = pcm.size // C
    if pcm.size % C or total_pcm == 0:
        raise ValueError("samples not divisible by channels")
    n_frames = (total_pcm + B - 1) // B
    last_len = total_pcm - (n_frames - 1) * B
    lo, hi = shard_range(n_frames, world, rank)
    an = None
    chunks, sizes = [], []
    if hi > lo:
        an = GpuAnalyzer(B, co.max_partition_order, co.max_lpc_order, co.mid_side, co.exhaustive_channel_correlation,
                         co.window_kind, co.window_param, bits_per_sample, C, max_frames=min(batch_frames, hi - lo),
                         device=device)
        f = lo
        while f < hi:
            take = min(batch_frames, hi - f)
            ll = last_len if f + take == n_frames else B
            a, b = f * B * C, ((f + take - 1) * B + ll) * C
            data, off = an.encode_frames(pcm[a:b], take, ll, f, sample_rate)
            chunks.append(data)
            sizes.extend(off[i + 1] - off[i] for i in range(take))
            f += take
        an.close()
    mine = b"".join(chunks)
    return finish_sharded_stream(mine, sizes, pcm, options, sample_rate, bits_per_sample, channels, dist)


def encode_stream_multi_device(pcm, options, sample_rate, bits_per_sample, channels, devices=None, batch_frames=1024,
                               depth=2):
    """ONE stream, ONE process, several GPUs (flacgpu_multi_*, include/flacenc_gpu.h): the C ABI cuts the stream's
    blocks into batches of contiguous frames dealt to the listed devices in turn (`devices=None`: all visible; an ordinal
    may repeat), merges the four-integer records and hands the frames back in stream order; the owner adds the MD5 and the metadata
    (flacenc_stream_header).  Returns (.flac bytes, per-shard records, merged record) -- the bytes a single
    FlacSampleWriter produces."""
    import numpy as np

    from .gpu import MultiDevice

    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    co = options._c_options()
    B, C = co.block_size, channels
    total_pcm = pcm.size // C
    if pcm.size % C or total_pcm == 0:
        raise ValueError("samples not divisible by channels")
    n_frames = (total_pcm + B - 1) // B
    last_len = total_pcm - (n_frames - 1) * B
    md = MultiDevice(B, co.max_partition_order, co.max_lpc_order, co.mid_side, co.exhaustive_channel_correlation,
                     co.window_kind, co.window_param, bits_per_sample, C, max_frames=min(batch_frames, n_frames),
                     devices=devices, depth=depth)
    try:
        body, off, per_shard, merged = md.encode(pcm, n_frames, last_len, 0, sample_rate)
    finally:
        md.close()
    sizes = [off[i + 1] - off[i] for i in range(n_frames)]
    assert merged == [n_frames, len(body), min(sizes), max(sizes)]
    return _with_header(body, sizes, pcm, options, sample_rate, bits_per_sample, channels), per_shard, merged


def _with_header(body, all_sizes, pcm, options, sample_rate, bits_per_sample, channels):
    """Owner's half: the stream MD5 and everything in front of the first frame (flacenc_stream_header)."""
    import hashlib

    import numpy as np

    from . import encode as E

    co = options._c_options()
    B, C = co.block_size, channels
    total_pcm = pcm.size // C
    n_frames = (total_pcm + B - 1) // B
    last_len = total_pcm - (n_frames - 1) * B
    width = (bits_per_sample + 7) // 8
    le = np.ascontiguousarray(pcm.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :width])
    md5 = hashlib.md5(le.tobytes()).digest()
    L = E._stream_lib()
    L.flacenc_stream_header.argtypes = [C_.POINTER(E._COptions), C_.c_uint32, C_.c_uint32, C_.c_uint32, C_.c_uint64,
                                        C_.c_char_p, C_.c_uint64, C_.POINTER(C_.c_uint32), C_.c_uint32, C_.c_void_p,
                                        C_.c_size_t, C_.POINTER(C_.c_size_t)]
    fs = (C_.c_uint32 * n_frames)(*all_sizes)
    ln = C_.c_size_t(0)
    cap = 1 << 20
    buf = (C_.c_uint8 * cap)()
    rc = L.flacenc_stream_header(C_.byref(co), sample_rate, bits_per_sample, C, total_pcm, md5, n_frames, fs, last_len,
                                 buf, cap, C_.byref(ln))
    if rc and ln.value > cap:
        cap = ln.value
        buf = (C_.c_uint8 * cap)()
        rc = L.flacenc_stream_header(C_.byref(co), sample_rate, bits_per_sample, C, total_pcm, md5, n_frames, fs,
                                     last_len, buf, cap, C_.byref(ln))
    E._check(rc)
    return bytes(buf[: ln.value]) + body


def finish_sharded_stream(mine, sizes, pcm, options, sample_rate, bits_per_sample, channels, dist=None):
    """The exchange + assembly half of `encode_stream_sharded`: every rank passes the finished frame bytes of ITS
    contiguous frame range (`mine`, with the per-frame `sizes`), rank 0 -- which owns the whole input `pcm` --
    gets the .flac back, the others None.  Used by bench.py --scaling strong (frames encoded by the timed loop)."""
    import hashlib

    import numpy as np

    from . import encode as E

    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    co = options._c_options()
    B, C = co.block_size, channels
    local = [len(sizes), len(mine), min(sizes) if sizes else 0, max(sizes) if sizes else 0]
    per_rank = all_gather_counters(local, dist)
    merged = merge_counters(per_rank, dist)
    if world > 1:
        import torch

        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        tsz = torch.tensor(sizes, dtype=torch.int32, device=dev) if sizes else torch.empty(0, dtype=torch.int32, device=dev)
        tby = (torch.frombuffer(bytearray(mine), dtype=torch.uint8).to(dev) if mine
               else torch.empty(0, dtype=torch.uint8, device=dev))
        gsz = gather_exact(tsz, [c[0] for c in per_rank], dist, torch.int32)
        gby = gather_exact(tby, [c[1] for c in per_rank], dist, torch.uint8)
        if rank != 0:
            return None
        all_sizes, body = [], []
        for r in range(world):
            all_sizes.extend(int(v) for v in gsz[r].tolist())
            body.append(gby[r].cpu().numpy().tobytes())
        body = b"".join(body)
    else:
        all_sizes, body = list(sizes), mine
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    total_pcm = pcm.size // C
    n_frames = (total_pcm + B - 1) // B
    last_len = total_pcm - (n_frames - 1) * B
    assert len(all_sizes) == n_frames and sum(all_sizes) == len(body) == merged["total_bytes"]
    return _with_header(body, all_sizes, pcm, options, sample_rate, bits_per_sample, channels)
