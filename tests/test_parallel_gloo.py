"""world_size-2 gloo test of the N>1 control path (runs on CPU): contiguous frame sharding and
the all-gather of per-shard counters that rebuilds stream-level bookkeeping."""
import os
import socket

import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist

    from flac_codec_amd.parallel import all_gather_counters, merge_counters, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(1001, world, rank)
    frames = hi - lo
    # deterministic fake frame sizes: frame f is 100 + (f % 7) bytes
    sizes = [100 + (f % 7) for f in range(lo, hi)]
    local = [frames, sum(sizes), min(sizes), max(sizes)]
    per_rank = all_gather_counters(local, dist, device="cpu")
    merged = merge_counters(per_rank)
    q.put((rank, lo, hi, merged))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_stream():
    from flac_codec_amd.parallel import shard_range

    for total in (1, 7, 8192, 1001):
        for world in (1, 2, 4, 8):
            r = [shard_range(total, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))


def test_gloo_all_gather_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_sizes = [100 + (f % 7) for f in range(1001)]
    for rank, lo, hi, merged in res:
        assert merged["total_frames"] == 1001
        assert merged["total_bytes"] == sum(all_sizes)
        assert merged["shard_byte_offsets"] == [0, sum(all_sizes[:res[1][1]])]
        assert merged["min_frame"] == 100 and merged["max_frame"] == 106


def _finish_worker(rank, world, port, q, n_frames, case):
    """finish_sharded_stream over gloo with synthetic frame bytes: the exchange (counters all-gathered, sizes and
    bytes received at their exact lengths) and the metadata rebuilt on rank 0; no GPU involved."""
    import numpy as np
    import torch.distributed as dist

    from flac_codec_amd.encode import Options
    from flac_codec_amd.parallel import finish_sharded_stream, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    B, C = 4096, 2
    rng = np.random.Generator(np.random.PCG64(case))
    pcm = rng.integers(-30000, 30000, size=(n_frames * B - 777) * C, dtype=np.int64).astype(np.int32)
    sizes_all = [int(v) for v in rng.integers(40, 9000, size=n_frames)]
    body_all = rng.integers(0, 256, size=sum(sizes_all), dtype=np.uint8).tobytes()
    lo, hi = shard_range(n_frames, world, rank)
    a, b = sum(sizes_all[:lo]), sum(sizes_all[:hi])
    out = finish_sharded_stream(body_all[a:b], sizes_all[lo:hi], pcm, Options.best(), 48000, 16, C,
                                dist if world > 1 else None)
    q.put((rank, out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run_finish(world, n_frames, case):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_finish_worker, args=(r, world, port, q, n_frames, case)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res[r] is None for r in range(1, world))
    return res[0]


def test_sharded_finish_matches_single_rank_for_1_2_3_8_ranks():
    """The .flac rank 0 assembles is the same whatever the rank count -- including 8 ranks for 5 frames (ranks without
    frames) and uneven shards."""
    for n_frames, case in ((37, 1), (5, 2)):
        ref = _run_finish(1, n_frames, case)
        assert ref[:4] == b"fLaC"
        for world in (2, 3, 8):
            assert _run_finish(world, n_frames, case) == ref, (world, n_frames)


def test_merge_counters_with_idle_ranks():
    from flac_codec_amd.parallel import merge_counters

    m = merge_counters([[2, 300, 100, 200], [0, 0, 0, 0], [1, 50, 50, 50], [0, 0, 0, 0]])
    assert (m["total_frames"], m["total_bytes"], m["min_frame"], m["max_frame"]) == (3, 350, 50, 200)
    assert m["shard_byte_offsets"] == [0, 300, 300, 350] and m["ranks_seen"] == 4 and m["frames_per_rank"] == [2, 0, 1, 0]
    assert merge_counters([[0, 0, 0, 0]])["min_frame"] == 0
