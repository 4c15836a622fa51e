"""ONE stream encoded by two ranks (two processes sharing GPU 0, gloo for the collectives, so that it
runs inside a 1-GPU lease): contiguous frame ranges per rank, counters all-gathered, frame sizes and
bytes gathered to the owner, metadata rebuilt by flacenc_stream_header -- the .flac must be the one
a single FlacSampleWriter produces (and the oracle's).  /root/reference/src/encode.rs:1999-2003,
2024-2110, 2414-2436."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CASES = [  # (channels, bps, samples per channel, preset, seconds of seek interval)
    (2, 24, 4096 * 21 + 333, "best"),
    (2, 16, 44100 * 12 + 5, "default"),
    (1, 16, 1152 * 9, "fast"),
]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import torch.distributed as dist

    from _pcm import synth_fast
    from flac_codec_amd.encode import Options
    from flac_codec_amd.parallel import encode_stream_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    outs = []
    for i, (ch, bps, n, preset) in enumerate(CASES):
        pcm = synth_fast(6100 + i, ch, bps, n)
        data = encode_stream_sharded(pcm, getattr(Options, preset)(), 44100, bps, ch, dist, device=0, batch_frames=5)
        outs.append(data)
    q.put((rank, outs))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_stream_byte_identical():
    import _oracle as orc
    from _pcm import synth_fast
    from flac_codec_amd.encode import FlacSampleWriter, Options

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(o is None for o in res[1])
    for i, (ch, bps, n, preset) in enumerate(CASES):
        pcm = synth_fast(6100 + i, ch, bps, n)
        w = FlacSampleWriter(None, getattr(Options, preset)(), 44100, bps, ch, pcm.size)
        w.write(pcm)
        w.finalize()
        single = w.getvalue()
        w.close()
        rc, ref, _ = orc.encode_stream(orc.options(preset), 44100, bps, ch, pcm, total_known=True)
        assert rc == 0 and single == ref
        assert res[0][i] == single, f"case {i}: the sharded stream differs from the single writer's"


def test_single_rank_is_the_same_path():
    from _pcm import synth_fast
    from flac_codec_amd.encode import FlacSampleWriter, Options
    from flac_codec_amd.parallel import encode_stream_sharded

    pcm = synth_fast(6200, 2, 16, 4096 * 7 + 9)
    data = encode_stream_sharded(pcm, Options.default(), 48000, 16, 2, None, batch_frames=3)
    w = FlacSampleWriter(None, Options.default(), 48000, 16, 2, pcm.size)
    w.write(pcm)
    w.finalize()
    assert data == w.getvalue()
    w.close()
