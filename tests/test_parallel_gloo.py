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
