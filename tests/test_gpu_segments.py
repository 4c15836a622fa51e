"""Frames of SEVERAL streams in one analysis batch (flacgpu_encode_segments*; the reference runs one
Encoder per file, /root/reference/src/encode.rs:487-627): every segment's frames must be the bytes the stream's own batch
gives (and the oracle's), with the stream's own frame numbers, whatever the order and the sizes of the segments -- through
the host path (segments uploaded to their places) and the device path (direct stereo input read in place through the
per-frame address table; other shapes gathered)."""
import numpy as np
import pytest

import _oracle as orc
from _compare import orc_options_for, planar_frames
from _pcm import synth_fast, synth_hi

pytestmark = pytest.mark.gpu


def _streams(channels, bps, B, counts, seed):
    out = []
    for i, n in enumerate(counts):
        gen = synth_hi if i % 2 else synth_fast
        kw = dict(segment=B, orders=[2, 5, 9, 12]) if i % 2 else {}
        out.append(np.ascontiguousarray(gen(seed + i, channels, bps, B * n, **kw)))
    return out


def _expect(stream, channels, bps, B, first, rate, max_lpc=12):
    oopts = orc_options_for(B, 6, max_lpc, True, True)
    out = []
    for f, planar in enumerate(planar_frames(stream, channels, B)):
        rc, fb, _ = orc.encode_frame(oopts, rate, bps, planar, frame_number=first + f)
        assert rc == 0
        out.append(fb)
    return out


@pytest.mark.parametrize("channels,bps,B", [(2, 24, 4096), (2, 16, 1152), (1, 16, 4096), (3, 24, 4096), (8, 24, 4096)])
def test_host_segments_are_each_streams_own_frames(channels, bps, B):
    from flac_codec_amd.gpu import GpuAnalyzer

    counts = [5, 1, 9, 3, 7]
    firsts = [0, 1000, 7, 0x7FFFFFF0, 31]
    streams = _streams(channels, bps, B, counts, 8100 + channels)
    an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, bps, channels, max_frames=sum(counts))
    data, off = an.encode_segments(list(zip(streams, firsts)), 48000)
    res, _ = an.verify_device(48000, 0)
    assert (res.frames, res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (sum(counts), 0, 0, 0)
    f = 0
    for s, n, first in zip(streams, counts, firsts):
        want = _expect(s, channels, bps, B, first, 48000)
        for k in range(n):
            assert data[off[f]:off[f + 1]] == want[k], f"segment starting at frame {first}: frame {k} differs from the oracle"
            f += 1
        # ... and what the stream's own batch gives
        own, own_off = an.encode_frames(s, n, B, first, 48000)
        assert own == b"".join(want) and own_off[n] == len(own)
    an.close()


@pytest.mark.parametrize("channels,bps", [(2, 24), (2, 16), (4, 24), (8, 24), (3, 16), (6, 24), (5, 24), (1, 16)])
def test_device_segments_in_place_and_gathered(channels, bps):
    """Direct stereo input and interleaved independent channels (transposing loads for 3 / 4 / 6 / 8, split producers for 5)
    are read in place through the address table (segments in shuffled order); with one of them a view that does not start
    on 16 bytes the whole batch is gathered instead, as one-channel streams always are."""
    import torch

    from flac_codec_amd.gpu import GpuAnalyzer

    B = 4096
    counts = [6, 2, 11, 4]
    firsts = [500, 0, 77, 123456]
    streams = _streams(channels, bps, B, counts, 8200 + channels)
    bufs = [torch.from_numpy(s).cuda() for s in streams]
    odd = torch.empty(streams[1].size + 1, dtype=torch.int32, device="cuda")
    odd[1:] = bufs[1]
    torch.cuda.synchronize()
    an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, bps, channels, max_frames=sum(counts))
    want = [_expect(s, channels, bps, B, first, 44100) for s, first in zip(streams, firsts)]
    for order, use_odd in (([2, 0, 3, 1], False), ([1, 3, 0, 2], True)):
        segs = []
        for i in order:
            ptr = odd.data_ptr() + 4 if (use_odd and i == 1) else bufs[i].data_ptr()
            segs.append((ptr, counts[i], firsts[i]))
        an.encode_segments_device(segs, 44100)
        data, off = an.fetch_frames()
        f = 0
        for i in order:
            for k in range(counts[i]):
                assert data[off[f]:off[f + 1]] == want[i][k], f"order {order}: stream {i} frame {k}"
                f += 1
        res, _ = an.verify_device(44100, 0)
        assert (res.frames, res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (sum(counts), 0, 0, 0)
    # the context goes back to ordinary batches afterwards (frame numbers from the call again)
    own, own_off = an.encode_frames(streams[0], counts[0], B, 9, 44100)
    assert own == b"".join(_expect(streams[0], channels, bps, B, 9, 44100))
    an.close()


@pytest.mark.parametrize("channels,bps,B", [(2, 24, 4096), (2, 16, 4096), (2, 16, 1152), (1, 24, 4096), (8, 24, 4096), (2, 32, 4096)])
def test_packed_segments_async_host(channels, bps, B):
    """flacgpu_encode_segments_packed_async_host (r06): the segments' blocks back to back in ONE pinned buffer at stream width
    (or int32), frames stored straight into pinned host memory -- the bytes of flacgpu_encode_segments and of the oracle,
    every frame with its own stream's frame number; an ordinary batch on the same context afterwards numbers its frames
    from the call again (no stale segment table)."""
    from test_gpu_packed import le_bytes

    from flac_codec_amd.gpu import GpuAnalyzer

    counts = [4, 1, 6, 2]
    firsts = [0, 99, 0xFFFFFFF0, 12]
    streams = _streams(channels, bps, B, counts, 8700 + channels + bps)
    max_lpc = 12
    an = GpuAnalyzer(B, 6, max_lpc, True, True, 2, 0.5, bps, channels, max_frames=sum(counts))
    width = (bps + 7) // 8
    if not an.packed_input_supported(width):
        width = 4
    blob = np.concatenate([le_bytes(s, width) for s in streams])
    data, off = an.encode_segments_packed(blob, width, list(zip(counts, firsts)), 48000)
    ref, ref_off = an.encode_segments(list(zip(streams, firsts)), 48000)
    assert data == ref and list(off) == list(ref_off)
    f = 0
    for s, n, first in zip(streams, counts, firsts):
        want = _expect(s, channels, bps, B, first, 48000, max_lpc)
        for k in range(n):
            assert data[off[f]:off[f + 1]] == want[k], f"segment starting at frame {first}: frame {k} differs from the oracle"
            f += 1
    own, _ = an.encode_frames(streams[2], counts[2], B, 5, 48000)
    assert own == b"".join(_expect(streams[2], channels, bps, B, 5, 48000, max_lpc))
    an.close()


def test_plain_batches_after_a_segments_batch_forget_its_tables():
    """flacgpu_pack_plans and the two-ranges branch of flacgpu_encode_device start a batch without
    passing the analysis entry that used to clear the segment state; after a segments batch on the same context their frames
    must carry the call's frame numbers, not the stale table's."""
    import torch

    from flac_codec_amd.gpu import GpuAnalyzer

    B, ch, bps = 4096, 2, 24
    n = 256
    s = synth_fast(8800, ch, bps, B * n)
    an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, bps, ch, max_frames=n)
    an.encode_segments([(s[: B * ch * 100], 5000), (s[B * ch * 100:], 70000)], 48000)
    want, want_off = an.encode_frames(s, n, B, 3, 48000)
    # two ranges: eligible from 256 frames on
    an.set_two_ranges(True)
    an.encode_segments([(s[: B * ch * 100], 5000), (s[B * ch * 100:], 70000)], 48000)
    d = torch.from_numpy(s).cuda()
    an.encode_device(d.data_ptr(), n, B, 3, 48000)
    got, off = an.fetch_frames()
    assert got == want and list(off) == list(want_off)
    res, _ = an.verify_device(48000, 3)
    assert (res.frames, res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (n, 0, 0, 0)
    an.set_two_ranges(False)
    # pack_plans: the decisions of the plain batch, packed again after another segments batch
    an.encode_frames(s, n, B, 3, 48000)
    plans, subs, _ = an.fetch(n, want_residuals=False)
    an.encode_segments([(s[: B * ch * 100], 5000), (s[B * ch * 100:], 70000)], 48000)
    got2, off2 = an.pack_plans(s, n, B, plans, subs, 3, 48000)
    assert got2 == want and list(off2) == list(want_off)
    an.close()


def test_too_many_frames_and_empty_segments_are_refused():
    from flac_codec_amd.gpu import GpuAnalyzer, GpuError

    an = GpuAnalyzer(4096, 6, 12, True, True, 2, 0.5, 24, 2, max_frames=4)
    s = synth_fast(8300, 2, 24, 4096 * 3)
    with pytest.raises(GpuError):
        an.encode_segments([(s, 0), (s, 100)], 48000)
    an.close()


def test_many_small_streams_coalesced_give_each_streams_own_file():
    """flacenc_encode_many_coalesced: streams of several shapes, lengths with and without a short last block, a stream
    shorter than one block -- every .flac byte-identical to flacenc_encode_many's and the oracle's."""
    from flac_codec_amd.encode import BatchEncoder, Options

    o = Options.best()
    cases = []
    for i in range(20):
        n = 4096 * (1 + i % 7) + (0 if i % 3 == 0 else 37 * i + 5)
        cases.append(synth_fast(8400 + i, 2, 24, n))
    cases.append(synth_fast(8450, 2, 24, 100))           # shorter than a block: the tail call alone
    plain = BatchEncoder(o, threads=4).encode(cases, 48000, 24, 2)
    co = BatchEncoder(o, threads=4, coalesce=True).encode(cases, 48000, 24, 2)
    assert co == plain
    oo = orc.options("best")
    for i in (0, 1, 5, 20):
        rc, ref, _ = orc.encode_stream(oo, 48000, 24, 2, cases[i], total_known=True)
        assert rc == 0 and co[i] == ref, i
    # another shape, other options (seek table by frames, padding, tags): the metadata is rebuilt from the frame sizes
    o2 = Options.default().seektable_frames(3).padding(500).tag("TITLE", "x")
    mono = [synth_fast(8500 + i, 1, 16, 4096 * (2 + i) + 11 * i) for i in range(6)]
    assert BatchEncoder(o2, threads=3, coalesce=True).encode(mono, 44100, 16, 1) == BatchEncoder(o2, threads=3).encode(mono, 44100, 16, 1)


def test_coalesced_shapes_the_packed_upload_cannot_take():
    """The ring uploads at the stream's width where the device can widen it (whole 16-byte groups per block); otherwise int32
    goes up and the MD5 bytes are packed beside it -- block sizes that are no multiple of four, 32-bit samples (width 4: the
    samples ARE the MD5 bytes), 8-bit mono; long streams (cut into segments, chains on the engines) and short ones (one
    segment, hashed by the workers) in one call, with and without a short last block."""
    from flac_codec_amd.encode import BatchEncoder, Options

    for o, bps, ch, B in ((Options.default().block_size(1001), 16, 2, 1001), (Options.default(), 32, 2, 4096),
                          (Options.best(), 8, 1, 4096), (Options.default().block_size(1152), 24, 2, 1152)):
        cases = []
        for i in range(12):
            blocks = (1, 3, 40, 70)[i % 4]
            n = B * blocks + (0 if i % 3 == 0 else 11 * i + 1)
            cases.append(synth_fast(8900 + i + bps, ch, bps, n))
        plain = BatchEncoder(o, threads=4).encode(cases, 44100, bps, ch)
        co = BatchEncoder(o.batch_frames(128), threads=4, coalesce=True).encode(cases, 44100, bps, ch)
        assert co == plain, (bps, ch, B)


def test_a_short_stream_is_never_cut_by_a_small_batch():
    """Found by tools/soak/soak_many.py coalesce (r06): a batch closed early in front of a short stream left the plan's small
    remainder batch at the head of that stream and cut it in two segments -- two HASH tasks then ran ONE MD5 chain from the same
    state.  A short stream travels whole: seven streams of 8 blocks, one of 7, one of 4 with 64-frame batches (the plan is
    64 + 3: the four-block stream meets the three-frame batch), every .flac (digest included) the per-stream writers'."""
    from flac_codec_amd.encode import BatchEncoder, Options

    B = 4096
    cases = [synth_fast(8950 + i, 2, 24, B * n + 100 * (i % 3)) for i, n in enumerate([8] * 7 + [7, 4])]
    o = Options.best()
    plain = BatchEncoder(o, threads=4).encode(cases, 48000, 24, 2)
    co = BatchEncoder(Options.best().batch_frames(64), threads=6, coalesce=True).encode(cases, 48000, 24, 2)
    assert co == plain
    rc, ref, _ = orc.encode_stream(orc.options("best"), 48000, 24, 2, cases[8], total_known=True)
    assert rc == 0 and co[8] == ref


def test_sixty_four_streams_of_512_frames():
    """VERDICT r04 item 6's shape: 64 streams x 512 frames of 24-bit stereo; coalesced == one writer per stream, two of them
    against the oracle."""
    from flac_codec_amd.encode import BatchEncoder, Options

    o = Options.best()
    base = synth_fast(8600, 2, 24, 4096 * 512 + 4096 * 64)
    streams = [base[2 * 4096 * i: 2 * 4096 * (i + 512)] for i in range(64)]     # overlapping windows: distinct, cheap to make
    co = BatchEncoder(o, threads=8, coalesce=True).encode(streams, 48000, 24, 2, copy=False)
    co = [bytes(v) for v in co]
    plain = BatchEncoder(o, threads=8).encode(streams, 48000, 24, 2, copy=False)
    for i in range(64):
        assert co[i] == bytes(plain[i]), i
    oo = orc.options("best")
    for i in (0, 63):
        rc, ref, _ = orc.encode_stream(oo, 48000, 24, 2, streams[i], total_known=True, threads=8)
        assert rc == 0 and co[i] == ref
