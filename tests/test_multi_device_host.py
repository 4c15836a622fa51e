"""Host half of the several-GPUs boundary (include/flacenc_gpu.h): flacgpu_merge_counters / flacgpu_shard_range are the C
restatement of flac_codec_amd/parallel.py's merge_counters / shard_range (the bookkeeping the reference's single process
keeps in `Encoder`: /root/reference/src/encode.rs:1999-2003 seek-point offsets, :2414-2436 min / max frame size).  No GPU
needed: pure host functions of the product library."""
import random

from flac_codec_amd import parallel
from flac_codec_amd.gpu import merge_counters_c, shard_range_c


def test_shard_ranges_match_the_python_cut_and_tile_the_stream():
    rng = random.Random(5)
    for _ in range(300):
        total = rng.choice([0, 1, 5, 37, 8192, 29127, (1 << 36) - 1, rng.randrange(1, 1 << 40)])
        shards = rng.randrange(1, 17)
        got = [shard_range_c(total, shards, k) for k in range(shards)]
        assert got == [parallel.shard_range(total, shards, k) for k in range(shards)]
        assert got[0][0] == 0 and got[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(got, got[1:]))          # contiguous, no gap, no overlap
        sizes = [hi - lo for lo, hi in got]
        assert max(sizes) - min(sizes) <= 1


def test_merge_matches_the_python_merge_with_idle_shards():
    rng = random.Random(6)
    for _ in range(300):
        n = rng.randrange(1, 12)
        per = []
        for _k in range(n):
            frames = rng.choice([0, 0, 1, rng.randrange(1, 5000)])
            if frames == 0:
                per.append([0, 0, 0, 0])
                continue
            lo = rng.randrange(14, 20000)
            hi = rng.randrange(lo, lo + 9000)
            per.append([frames, rng.randrange(frames * lo, frames * hi + 1), lo, hi])
        merged, offs = merge_counters_c(per)
        want = parallel.merge_counters(per)
        assert merged == [want["total_frames"], want["total_bytes"], want["min_frame"], want["max_frame"]]
        assert offs == want["shard_byte_offsets"]


def test_every_multi_device_symbol_is_exported():
    from flac_codec_amd import _lib

    have = _lib.exported_symbols()
    for name in ("flacgpu_multi_create", "flacgpu_multi_encode", "flacgpu_multi_encode_device", "flacgpu_multi_wait",
                 "flacgpu_multi_counters", "flacgpu_multi_destroy", "flacgpu_rccl_allgather_counters",
                 "flacgpu_device_count", "flacenc_encode_many_devices"):
        assert name in have, name
