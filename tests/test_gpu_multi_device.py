"""Several GPUs behind the C ABI, one process (/root/reference/src/encode.rs:1997-2022, 2414-2436,
3964-4010): on a 1-GPU box device 0 is listed two or three times -- every listing is a shard with contexts, a pipeline
and a host thread of its own, so the whole multi-shard path runs -- and the stream must be the single writer's (and the
oracle's) byte for byte.  Plus the one collective of the one-process-per-GPU shape over a REAL RCCL communicator (one
rank: what a 1-GPU box can host), through the C entry and through torch.distributed's nccl backend."""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as orc
from _pcm import synth_fast

pytestmark = pytest.mark.gpu

CONFIGS = {  # BASELINE configs 3 and 4 (bench.CONFIGS), shrunk in frames only
    3: dict(ch=2, bps=24, rate=48000, lpc=12, po=6),
    4: dict(ch=8, bps=24, rate=192000, lpc=12, po=6),
}


def _single_writer(pcm, o, rate, bps, ch):
    from flac_codec_amd.encode import FlacSampleWriter

    w = FlacSampleWriter(None, o, rate, bps, ch, pcm.size)
    w.write(pcm)
    w.finalize()
    data = w.getvalue()
    w.close()
    return data


@pytest.mark.parametrize("config,listing,frames,tail", [(3, [0, 0], 37, 333), (3, [0, 0, 0], 26, 0), (4, [0, 0], 13, 7),
                                                        (4, [0, 0, 0], 12, 0), (3, [0, 0, 0], 2, 0)])
def test_device_listed_several_times_gives_the_single_writers_stream(config, listing, frames, tail):
    from flac_codec_amd.encode import Options
    from flac_codec_amd.parallel import encode_stream_multi_device

    cfg = CONFIGS[config]
    ch, bps, rate = cfg["ch"], cfg["bps"], cfg["rate"]
    pcm = synth_fast(7000 + config + frames, ch, bps, 4096 * frames + tail)
    o = Options.best().max_lpc_order(cfg["lpc"]).max_partition_order(cfg["po"])
    data, per_shard, merged = encode_stream_multi_device(pcm, o, rate, bps, ch, devices=listing, batch_frames=5)
    single = _single_writer(pcm, o, rate, bps, ch)
    assert data == single, "the multi-device stream differs from the single writer's"
    oo = orc.options("best").copy(max_lpc_order=cfg["lpc"], max_partition_order=cfg["po"])
    rc, ref, _ = orc.encode_stream(oo, rate, bps, ch, pcm, total_known=True)
    assert rc == 0 and data == ref
    n_frames = frames + (1 if tail else 0)
    # batches of 5 frames dealt to the shards in turn (r06): shard k encodes batches k, k + G, ...
    G, dealt = len(listing), [0] * len(listing)
    for j, first in enumerate(range(0, n_frames, 5)):
        dealt[j % G] += min(5, n_frames - first)
    assert [c[0] for c in per_shard] == dealt
    assert merged[0] == n_frames and merged[1] == sum(c[1] for c in per_shard)
    busy = [c for c in per_shard if c[0]]
    assert merged[2] == min(c[2] for c in busy) and merged[3] == max(c[3] for c in busy)


def test_packed_samples_and_more_shards_than_frames():
    """3-byte little-endian samples (what FlacByteWriter receives) through the multi-device call; five shards for three
    frames: two idle shards that take no part in min / max."""
    from flac_codec_amd.gpu import GpuAnalyzer, MultiDevice

    ch, bps, B = 2, 24, 4096
    pcm = synth_fast(7100, ch, bps, B * 3)
    le = np.ascontiguousarray(pcm.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3]).reshape(-1)
    md = MultiDevice(B, 6, 12, True, True, 2, 0.5, bps, ch, max_frames=2, devices=[0] * 5)
    body, off, per, merged = md.encode(le, 3, B, 40, 48000, bytes_per_sample=3)
    md.close()
    an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, bps, ch, max_frames=3)
    want, want_off = an.encode_frames(pcm, 3, B, 40, 48000)
    an.close()
    assert body == want and off == want_off
    # batches of two frames dealt in turn: {0, 1} to shard 0, {2} to shard 1, three idle shards
    assert [c[0] for c in per] == [2, 1, 0, 0, 0] and merged[0] == 3 and merged[2] == min(c[2] for c in per if c[0])


def test_one_host_copy_per_output_byte_and_threads_near_their_gpu():
    """every retired batch goes from its pinned slot straight to its place in `out` -- the host copies
    exactly as many bytes as it hands out (r05: three times as many) --, a too-small `out` reports the size and copies only
    what fits, and the shards' parked threads are bound to their GPU's local CPUs where sysfs names them."""
    from flac_codec_amd import _lib
    from flac_codec_amd.gpu import GpuAnalyzer, MultiDevice

    ch, bps, B, F = 2, 24, 4096, 41
    pcm = synth_fast(7150, ch, bps, B * F)
    md = MultiDevice(B, 6, 12, True, True, 2, 0.5, bps, ch, max_frames=4, devices=[0, 0, 0], depth=3)
    an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, bps, ch, max_frames=F)
    want, want_off = an.encode_frames(pcm, F, B, 9, 48000)
    an.close()
    for _ in range(3):
        body, off, per, merged = md.encode(pcm, F, B, 9, 48000)
        assert body == want and off == want_off
    out_b, copied, near = md.host_copy_stats()
    assert out_b == 3 * len(want) and copied == out_b, (out_b, copied)
    node, cpus = md.numa_info(0)
    assert near == (2 if cpus else 0), (near, node, cpus)     # (shard 0 runs on the calling thread)
    # a buffer that is too small: the total comes back, nothing past the buffer's end is written
    L = _lib.lib()
    small = np.zeros(len(want) // 2 + 8, dtype=np.uint8)
    small[len(want) // 2:] = 0xA5
    total = C.c_uint64(0)
    src = np.ascontiguousarray(pcm, dtype=np.int32)
    rc = L.flacgpu_multi_encode(md._h, C.c_void_p(src.ctypes.data), 4, F, B, 9, 48000, C.c_void_p(small.ctypes.data),
                                len(want) // 2, None, C.byref(total), None, None)
    assert rc == -5 and total.value == len(want) and bytes(small[len(want) // 2:]) == b"\xa5" * 8
    md.close()


def test_resident_batches_rotate_through_every_shards_contexts():
    """flacgpu_multi_encode_device / _wait / _counters: what bench.py --in-process times (PCM resident in HBM)."""
    import torch

    from flac_codec_amd.gpu import GpuAnalyzer, MultiDevice

    ch, bps, B, F = 2, 24, 4096, 6
    md = MultiDevice(B, 6, 12, True, True, 2, 0.5, bps, ch, max_frames=F, devices=[0, 0, 0], depth=2)
    pcms = [synth_fast(7200 + k, ch, bps, B * F) for k in range(3)]
    bufs = [torch.from_numpy(p).to("cuda:0") for p in pcms]
    torch.cuda.synchronize()
    for _round in range(3):                       # every context of every shard gets a batch
        for k in range(3):
            md.encode_device(k, bufs[k].data_ptr(), F, B, 100 * k, 48000)
    md.wait()
    per, merged = md.counters()
    an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, bps, ch, max_frames=F)
    for k in range(3):
        want, want_off = an.encode_frames(pcms[k], F, B, 100 * k, 48000)
        got, off = md.fetch_last(k, F)
        assert got == want and off == want_off
        sizes = [want_off[i + 1] - want_off[i] for i in range(F)]
        assert per[k] == [F, len(want), min(sizes), max(sizes)]
    an.close()
    md.close()
    assert merged[0] == 3 * F and merged[1] == sum(c[1] for c in per)


def test_many_streams_dealt_over_a_device_list():
    from flac_codec_amd.encode import BatchEncoder, Options

    streams = [synth_fast(7300 + i, 2, 16, 4096 * (3 + i % 4) + 17 * i) for i in range(9)]
    o = Options.default()
    one = BatchEncoder(o, threads=4).encode(streams, 44100, 16, 2)
    many = BatchEncoder(o, threads=4, devices=[0, 0, 0]).encode(streams, 44100, 16, 2)
    every = BatchEncoder(o, threads=4, devices="all").encode(streams, 44100, 16, 2)
    assert one == many == every
    rc, ref, _ = orc.encode_stream(orc.options("default"), 44100, 16, 2, streams[4], total_known=True)
    assert rc == 0 and many[4] == ref


def test_bad_device_lists_are_refused():
    from flac_codec_amd.gpu import GpuError, MultiDevice

    with pytest.raises(GpuError):
        MultiDevice(4096, 6, 12, True, True, 2, 0.5, 24, 2, max_frames=4, devices=[0, 99])
    with pytest.raises(GpuError):
        MultiDevice(4096, 6, 12, True, True, 2, 0.5, 24, 2, max_frames=4, devices=[-1])


def test_counters_all_gathered_over_a_real_rccl_communicator():
    """The C entry for the one-process-per-GPU shape on a communicator the CALLER made (ncclCommInitAll through ctypes,
    as a Rust / C host would through its own RCCL binding): one rank is what a 1-GPU box can host."""
    from flac_codec_amd import _lib
    from flac_codec_amd.gpu import rccl_allgather_counters

    assert _lib.lib().flacgpu_rccl_available() == 1
    rccl = C.CDLL("librccl.so.1")
    comm = C.c_void_p(None)
    devs = (C.c_int * 1)(0)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, devs) == 0
    try:
        local = [8192, 133935808, 16317, 16376]
        got, me = rccl_allgather_counters(comm.value, local)
        assert got == [local] and me == 0
        # too small a receive array is an error, not a truncation
        out = (_lib.ShardCounters * 1)()
        n = C.c_uint32(0)
        mine = _lib.ShardCounters(*local)
        rc = _lib.lib().flacgpu_rccl_allgather_counters(comm, None, C.byref(mine), out, 0, C.byref(n), None)
        assert rc == -5 and n.value == 1
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_python_all_gather_runs_on_the_nccl_backend():
    """parallel.all_gather_counters with backend "nccl" (= RCCL) -- until r05 it had only ever run on gloo."""
    import torch.distributed as dist

    from flac_codec_amd.parallel import all_gather_counters, merge_counters

    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        import torch

        torch.cuda.set_device(0)
        local = [37, 600000, 15000, 16400]
        # world_size 1 short-cuts inside all_gather_counters: call the collective itself
        mine = torch.tensor(local, dtype=torch.int64, device="cuda")
        out = [torch.zeros_like(mine)]
        dist.all_gather(out, mine)
        torch.cuda.synchronize()
        per = [[int(v) for v in out[0].tolist()]]
        assert per == [local] == all_gather_counters(local, dist)
        m = merge_counters(per, dist)
        assert m["backend"] == "nccl" and m["ranks_seen"] == 1 and m["total_frames"] == 37
    finally:
        dist.destroy_process_group()
