"""Full-size checks (BASELINE.json configs at their real sizes) through size-independent
properties: encode -> decode round trip with MD5, CRC-8/16 on every frame, byte identity on
sampled frames, idempotence across batch sizes."""
import os
import struct
import sys

import numpy as np
import pytest

import _oracle as orc
from _compare import orc_options_for
from _pcm import generate_sine_2, synth_fast

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tiled(seed, ch, bps, frames, block=4096, distinct=256):
    base = synth_fast(seed, ch, bps, block * min(distinct, frames))
    # vary each repetition a little so frames are not identical
    reps = (frames + distinct - 1) // distinct
    parts = []
    for r in range(reps):
        parts.append(((base.astype(np.int64) + (r & 3)) .clip(-(1 << (bps - 1)), (1 << (bps - 1)) - 1)).astype(np.int32))
    return np.concatenate(parts)[: frames * block * ch]


@pytest.mark.parametrize("ch,bps,rate,frames,lpc", [
    (2, 24, 48000, 8192, 12),    # config 3 (headline) at full size
    (2, 16, 48000, 8192, 0),     # config 2 (fixed predictors only)
    (8, 24, 192000, 2048, 12),   # config 4 shape (per-GPU share)
    (2, 24, 96000, 2048, 32),    # config 5 (order 32)
])
def test_full_size_roundtrip(ch, bps, rate, frames, lpc):
    from flac_codec_amd.encode import FlacSampleWriter, Options

    pcm = tiled(300 + ch + bps + lpc, ch, bps, frames)
    opts = Options.best().batch_frames(min(frames, 4096))
    opts.max_lpc_order(lpc if lpc else None)
    if lpc == 0:
        opts.max_partition_order(5)
    w = FlacSampleWriter(None, opts, rate, bps, ch, pcm.size)
    step = 4096 * ch * 1000 + 12345 * ch
    for s in range(0, pcm.size, step):
        w.write(pcm[s:s + step])
    w.finalize()
    data = w.getvalue()
    st = w.stats()
    w.close()
    rc, out, info = orc.decode_stream(data)   # verifies CRC-8 / CRC-16 of every frame + MD5
    assert rc == 0 and info.md5_ok == 1
    assert info.frames == frames == st.frames
    assert np.array_equal(out, pcm)
    assert (info.min_frame, info.max_frame) == (st.min_frame_size, st.max_frame_size)
    # byte identity with the oracle on a sample of frames spread over the stream
    oo = orc_options_for(4096, opts._c.max_partition_order, lpc, True, True)
    # locate frames by re-encoding sampled blocks with the oracle and searching the stream
    for f in (0, 1, frames // 2, frames - 1):
        blk = pcm[f * 4096 * ch:(f + 1) * 4096 * ch].reshape(4096, ch).T
        rc, fb, _ = orc.encode_frame(oo, rate, bps, np.ascontiguousarray(blk), frame_number=f)
        assert rc == 0 and fb in data, f"frame {f} not byte-identical to the oracle"


@pytest.mark.parametrize("block,bps,lpc,exhaustive,mid_side", [
    (1152, 16, 0, False, False),    # Options::fast() at the bench's size: 29 127 frames per batch
    (1152, 24, 12, True, True),
    (2304, 24, 12, True, True),
    (1024, 16, 8, False, True),
    (2048, 24, 12, True, False),
])
def test_full_size_short_blocks_read_in_place(block, bps, lpc, exhaustive, mid_side):
    """The shorter wave block lengths at the bench's batch size (67 M samples of interleaved stereo resident in HBM,
    analysed and assembled in place): every frame decodes back to its input on the device (structure, CRC-16, PCM),
    frames sampled over the batch equal the oracle's, and the copy-input path gives the same bytes."""
    import torch

    from flac_codec_amd.gpu import GpuAnalyzer

    frames = 8192 * 4096 // block
    pcm = tiled(500 + block + bps + lpc, 2, bps, frames, block=block, distinct=509)
    try:
        d = torch.from_numpy(pcm).cuda()
    except RuntimeError as e:   # torch initialised after the library in this process does not always find the GPU
        pytest.skip(f"torch cannot use the GPU here: {e}")
    an = GpuAnalyzer(block, 6, lpc, mid_side, exhaustive, 2, 0.5, bps, 2, max_frames=frames)
    an.set_timing(True)
    an.analyze_device(d.data_ptr(), frames, block)
    assert "k_deinterleave" not in an.kernel_ms()
    an.set_timing(False)
    an.encode_device(d.data_ptr(), frames, block, 3, 44100)
    data, off = an.fetch_frames(frames)
    res, _ = an.verify_device(44100, 3)
    assert (res.frames, res.bad_structure, res.bad_crc16, res.frames_pcm_differs, res.samples_differ) == (frames, 0, 0, 0, 0)
    an.set_tuning(an.TUNE_COPY_INPUT, 1)
    an.encode_device(d.data_ptr(), frames, block, 3, 44100)
    data2, off2 = an.fetch_frames(frames)
    an.close()
    assert off2 == off and data2 == data
    oo = orc_options_for(block, 6, lpc, mid_side, exhaustive)
    for f in (0, 1, 508, 509, frames // 2, frames - 2, frames - 1):
        blk = pcm[f * block * 2:(f + 1) * block * 2].reshape(block, 2).T
        rc, fb, _ = orc.encode_frame(oo, 44100, bps, np.ascontiguousarray(blk), frame_number=3 + f)
        assert rc == 0 and data[off[f]:off[f + 1]] == fb, f"frame {f} differs from the oracle"


def test_stereo_batches_beyond_96_mi_samples_are_cut_in_ranges():
    """Twice the bench's batch (16384 stereo frames, 134 M samples) in ONE flacgpu_encode_device call is run as two ranges of
    8192 frames by default (FLACGPU_TUNE_CHUNK_MSAMPLES): the bytes of the uncut batch, every frame decoding back to its input
    on the device, sampled frames equal to the oracle's."""
    import torch

    from flac_codec_amd.gpu import GpuAnalyzer

    frames = 16384
    pcm = tiled(771, 2, 24, frames, block=4096, distinct=509)
    try:
        d = torch.from_numpy(pcm).cuda()
    except RuntimeError as e:
        pytest.skip(f"torch cannot use the GPU here: {e}")
    outs = []
    for chunk in (None, 0):
        an = GpuAnalyzer(4096, 6, 12, True, True, 2, 0.5, 24, 2, max_frames=frames)
        if chunk is not None:
            an.set_tuning(an.TUNE_CHUNK_MSAMPLES, chunk)
        an.encode_device(d.data_ptr(), frames, 4096, 9, 48000)
        outs.append(an.fetch_frames(frames))
        res, _ = an.verify_device(48000, 9)
        assert (res.frames, res.bad_structure, res.bad_crc16, res.frames_pcm_differs, res.samples_differ) == (frames, 0, 0, 0, 0)
        an.close()
    assert outs[0] == outs[1]
    data, off = outs[0]
    oo = orc_options_for(4096, 6, 12, True, True)
    for f in (0, 8191, 8192, 8193, frames - 1):
        blk = pcm[f * 4096 * 2:(f + 1) * 4096 * 2].reshape(4096, 2).T
        rc, fb, _ = orc.encode_frame(oo, 48000, 24, np.ascontiguousarray(blk), frame_number=9 + f)
        assert rc == 0 and data[off[f]:off[f + 1]] == fb, f"frame {f} differs from the oracle"


def test_batch_size_independence():
    """The same stream encoded with different GPU batch sizes gives identical bytes."""
    from flac_codec_amd.encode import FlacSampleWriter, Options

    pcm = synth_fast(400, 2, 24, 4096 * 37 + 55)
    outs = []
    for bf in (1, 7, 64):
        w = FlacSampleWriter(None, Options.best().batch_frames(bf), 48000, 24, 2, pcm.size)
        w.write(pcm)
        w.finalize()
        outs.append(w.getvalue())
        w.close()
    assert outs[0] == outs[1] == outs[2]


def test_wav2flac_example(tmp_path):  # BASELINE config 1: 10 s 44.1 kHz/16-bit stereo sine WAV
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import wav2flac

    pcm = generate_sine_2(32767.0, 44100.0, 441000, 441.0, 0.5, 441.0, 0.0, 1.0)
    raw = pcm.astype("<i2").tobytes()
    wav = tmp_path / "sine.wav"
    with open(wav, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 2, 44100, 44100 * 4, 4, 16))
        f.write(b"data" + struct.pack("<I", len(raw)) + raw)
    out = tmp_path / "sine.flac"
    wav2flac.convert_wav(str(wav), str(out))
    data = open(out, "rb").read()
    rc, ref, _ = orc.encode_stream(orc.options("default"), 44100, 16, 2, pcm, total_known=True)
    assert rc == 0 and data == ref
