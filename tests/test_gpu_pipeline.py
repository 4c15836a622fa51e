"""flacgpu_pipeline_* (include/flacenc_gpu.h): the full-duplex host -> host batch loop.  Through ctypes (every batch of a
rotation must be the synchronous call's bytes, in submit order, for int32 and stream-width uploads, short last batch
and short last frame included) and through examples/c_abi_pipeline.c built with gcc (the recipe INTEGRATION.md points a
binding's `encode_blocks` at), whose frames must hash to the oracle's."""
import os
import subprocess

import numpy as np
import pytest

import _oracle as orc
from _compare import orc_options_for, planar_frames
from _pcm import synth_fast, synth_hi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = 4096


@pytest.mark.parametrize("bps,width", [(24, 4), (24, 3), (16, 2), (16, 4)])
def test_rotation_gives_the_synchronous_bytes(bps, width):
    from flac_codec_amd.gpu import GpuAnalyzer, PinnedBuffer, Pipeline

    depth, fpb, batches = 3, 24, 8
    total_frames = fpb * (batches - 1) + 7                     # the last batch is short ...
    n = B * total_frames - 1234                                  # ... and so is its last frame (samples per channel)
    pcm = np.ascontiguousarray(np.concatenate([synth_fast(300 + bps, 2, bps, n // 2), synth_hi(301, 2, bps, n - n // 2)]))
    assert pcm.size == 2 * n
    an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, bps, 2, max_frames=fpb)
    pipe = Pipeline(B, 6, 12, True, True, 2, 0.5, bps, 2, max_frames=fpb, depth=depth)
    bufs = [PinnedBuffer(fpb * B * 2 * 4) for _ in range(depth)]
    got = []
    expect = []
    submitted = 0
    f0 = 0
    while f0 < total_frames or pipe.in_flight():
        if f0 < total_frames and pipe.in_flight() < depth:
            nf = min(fpb, total_frames - f0)
            chunk = pcm[f0 * B * 2: min(n, (f0 + nf) * B) * 2]
            last = chunk.size // 2 - (nf - 1) * B
            buf = bufs[submitted % depth]
            if width == 4:
                buf.array[: chunk.size * 4] = chunk.view(np.uint8)
            else:
                le = chunk.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :width].reshape(-1)
                buf.array[: le.size] = le
            assert pipe.submit(buf.address, width, nf, last, 100 + f0, 44100)
            expect.append(an.encode_frames(chunk, nf, last, 100 + f0, 44100))
            submitted += 1
            f0 += nf
        else:
            got.append(pipe.retire())
    assert len(got) == len(expect) == batches
    for g, e in zip(got, expect):
        assert g[0] == e[0] and g[1] == e[1]
    # and the frames are the oracle's
    oopts = orc_options_for(B, 6, 12, True, True)
    data = b"".join(g[0] for g in got)
    pos = 0
    for f, planar in enumerate(planar_frames(pcm, 2, B)):
        rc, fb, _ = orc.encode_frame(oopts, 44100, bps, planar, frame_number=100 + f)
        assert rc == 0 and data[pos:pos + len(fb)] == fb, f
        pos += len(fb)
    assert pos == len(data)
    pipe.close()
    an.close()
    for b in bufs:
        b.close()


def test_busy_and_empty():
    from flac_codec_amd.gpu import GpuError, PinnedBuffer, Pipeline

    pipe = Pipeline(B, 6, 12, True, True, 2, 0.5, 24, 2, max_frames=4, depth=2)
    buf = PinnedBuffer(4 * B * 2 * 4)
    buf.array[:] = 0
    with pytest.raises(GpuError):
        pipe.retire()                                   # nothing in flight
    assert pipe.submit(buf.address, 4, 4, B, 0, 48000)
    assert pipe.submit(buf.address, 4, 4, B, 4, 48000)
    assert pipe.submit(buf.address, 4, 4, B, 8, 48000) is False   # FLACGPU_ERR_BUSY: both slots hold a batch
    a = pipe.retire()
    assert pipe.submit(buf.address, 4, 4, B, 8, 48000)
    b = pipe.retire()
    c = pipe.retire()
    assert len(a[1]) == len(b[1]) == len(c[1]) == 5 and pipe.in_flight() == 0
    pipe.close()
    buf.close()


def make_signal(frames, seed):
    """examples/c_abi_pipeline.c make_signal, restated."""
    s = seed
    y1 = y2 = 0
    out = np.empty(frames * 2, dtype=np.int32)
    for i in range(frames):
        s = (s * 1103515245 + 12345) & 0xFFFFFFFF
        e = ((s >> 10) & 0x3FFFF) - 131072
        y = ((58000 * y1 - 29491 * y2) >> 15) + e
        y = max(-8000000, min(8000000, y))
        y2, y1 = y1, y
        s = (s * 1103515245 + 12345) & 0xFFFFFFFF
        e2 = ((s >> 14) & 0xFFF) - 2048
        out[2 * i] = y
        out[2 * i + 1] = ((3 * y) >> 2) + e2
    return out


@pytest.mark.parametrize("width", [4, 3])
def test_c_recipe_matches_oracle(tmp_path, width):
    exe = str(tmp_path / "c_abi_pipeline")
    libdir = os.path.join(ROOT, "flac-codec_amd")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_pipeline.c"), "-L" + libdir, "-lflacenc_amd",
                           "-Wl,-rpath," + libdir, "-o", exe])
    batches, fpb = 6, 8
    out = subprocess.check_output([exe, str(batches), str(fpb), str(width)], text=True).split()
    size, digest, n_frames = int(out[0]), int(out[1], 16), int(out[2])
    oopts = orc_options_for(B, 6, 12, True, True)
    h = 1469598103934665603
    total = 0
    sigs = {k: make_signal(fpb * B, 1000 + k) for k in range(4)}
    for b in range(batches):
        for f, planar in enumerate(planar_frames(sigs[b % 4], 2, B)):
            rc, fb, _ = orc.encode_frame(oopts, 48000, 24, planar, frame_number=b * fpb + f)
            assert rc == 0
            total += len(fb)
            for byte in fb:
                h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert (size, n_frames) == (total, batches * fpb)
    assert digest == h


def test_eight_channels_through_the_pipeline():
    """5..8 channels are assembled by k_sub64 (one workgroup per subframe), whose plain stores write the frames into the
    slot's pinned host buffer like k_frame64's do: the rotation must still give the synchronous call's bytes and the oracle's."""
    from flac_codec_amd.gpu import GpuAnalyzer, PinnedBuffer, Pipeline

    ch, bps, fpb, depth = 8, 24, 6, 2
    pcm = np.ascontiguousarray(synth_hi(77, ch, bps, B * fpb * 3, sections=6))
    an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, bps, ch, max_frames=fpb)
    pipe = Pipeline(B, 6, 12, True, True, 2, 0.5, bps, ch, max_frames=fpb, depth=depth)
    bufs = [PinnedBuffer(fpb * B * ch * 4) for _ in range(depth)]
    got, expect = [], []
    for b in range(3):
        if pipe.in_flight() == depth:
            got.append(pipe.retire())
        chunk = pcm[b * fpb * B * ch: (b + 1) * fpb * B * ch]
        bufs[b % depth].array[:] = chunk.view(np.uint8)
        assert pipe.submit(bufs[b % depth].address, 4, fpb, B, b * fpb, 192000)
        expect.append(an.encode_frames(chunk, fpb, B, b * fpb, 192000))
    while pipe.in_flight():
        got.append(pipe.retire())
    assert [g[0] for g in got] == [e[0] for e in expect] and [g[1] for g in got] == [e[1] for e in expect]
    oopts = orc_options_for(B, 6, 12, True, True)
    data = b"".join(g[0] for g in got)
    pos = 0
    for f, planar in enumerate(planar_frames(pcm, ch, B)):
        rc, fb, _ = orc.encode_frame(oopts, 192000, bps, planar, frame_number=f)
        assert rc == 0 and data[pos:pos + len(fb)] == fb, f
        pos += len(fb)
    pipe.close()
    an.close()
    for b in bufs:
        b.close()
