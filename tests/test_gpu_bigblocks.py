"""Block sizes above what the LDS-resident kernels hold (16384 < B <= 65535: arrays in HBM) and
batches of more than 65535 frames -- every block size the reference accepts
(/root/reference/src/encode.rs:1418-1423) in the shape of its own test
(/root/reference/tests/format.rs:1248-1305: noise through FlacByteWriter, block sizes default / 32 /
32768 / 65535 x Options default / fast / best), byte-identical to the oracle and round-tripped."""
import numpy as np
import pytest

import _oracle as orc
from _compare import compare_frame, orc_options_for, planar_frames
from _pcm import synth_fast

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bs,channels,bps", [(32768, 2, 16), (65535, 2, 24), (20000, 1, 16), (65535, 3, 8),
                                             (32768, 2, 32), (49152, 8, 24)])
def test_big_block_analysis_and_frames(bs, channels, bps):
    from flac_codec_amd.gpu import GpuAnalyzer

    pcm = synth_fast(3000 + bs % 977 + channels, channels, bps, bs * 2 + bs // 3)
    max_lpc, max_po = 12, 6
    frames = planar_frames(pcm, channels, bs)
    n_frames, last = len(frames), frames[-1].shape[1]
    an = GpuAnalyzer(bs, max_po, max_lpc, True, True, 2, 0.5, bps, channels, max_frames=n_frames)
    used = pcm[: ((n_frames - 1) * bs + last) * channels]
    plans, subs, res = an.analyze(used, n_frames, last)
    oopts = orc_options_for(bs, max_po, max_lpc, True, True)
    an.pack_device(5, 44100)
    data, off = an.fetch_frames(n_frames)
    for f, planar in enumerate(frames):
        rc, fb, oplan = orc.encode_frame(oopts, 44100, bps, planar, frame_number=5 + f)
        assert rc == 0
        compare_frame(plans[f], subs[f * channels:(f + 1) * channels], res[f], oplan, planar, planar.shape[1],
                      where=f"bs {bs} frame {f}")
        assert data[off[f]:off[f + 1]] == fb, f"bs {bs} frame {f}: packed bytes differ from the oracle"
    vres, _ = an.verify_device(44100, 5)
    assert (vres.bad_structure, vres.bad_crc16, vres.frames_pcm_differs) == (0, 0, 0)
    an.close()


@pytest.mark.parametrize("channels,bps", [(1, 8), (2, 16), (2, 24)])
def test_noise_like_the_reference(channels, bps):
    from flac_codec_amd.encode import FlacByteWriter, Options
    from flac_codec_amd.gpu import decode_stream

    rng = np.random.Generator(np.random.PCG64(77 + channels + bps))
    noise = rng.integers(0, 256, size=393216, dtype=np.uint8).tobytes()
    width = (bps + 7) // 8
    pcm = np.frombuffer(noise, dtype=np.uint8).reshape(-1, width)
    val = np.zeros(pcm.shape[0], dtype=np.int64)
    for k in range(width):
        val |= pcm[:, k].astype(np.int64) << (8 * k)
    val = ((val + (1 << (8 * width - 1))) % (1 << (8 * width))) - (1 << (8 * width - 1))
    samples = val.astype(np.int32)
    for preset in ("default", "fast", "best"):
        for block_size in (None, 32, 32768, 65535):
            opt = getattr(Options, preset)()
            oo = orc.options(preset, padding=-1)
            if block_size:
                opt = opt.block_size(block_size)
                oo = oo.copy(block_size=block_size)
            w = FlacByteWriter(None, opt.no_padding(), 44100, bps, channels, len(noise))
            w.write(noise)
            w.finalize()
            data = w.getvalue()
            w.close()
            rc, ref, _ = orc.encode_stream(oo, 44100, bps, channels, samples, total_known=True)
            assert rc == 0 and data == ref, f"{preset} block {block_size}: differs from the oracle"
            out, info = decode_stream(data)      # the reference's check: the file round-trips
            assert info.md5_status == 1 and info.bad_frames == 0 and np.array_equal(out, samples)


def test_batch_of_more_than_65535_frames():
    """K0 indexes frames with blockIdx.y: one batch of 70 001 small frames takes two launches."""
    from flac_codec_amd.gpu import GpuAnalyzer

    bs, n_frames = 16, 70001
    pcm = synth_fast(3100, 2, 16, bs * n_frames)
    an = GpuAnalyzer(bs, 4, 8, True, True, 2, 0.5, 16, 2, max_frames=n_frames)
    data, off = an.encode_frames(pcm, n_frames, bs, 0, 44100)
    oopts = orc_options_for(bs, 4, 8, True, True)
    frames = pcm.reshape(n_frames, bs, 2)
    for f in list(range(0, 40)) + list(range(65500, 65600)) + list(range(n_frames - 40, n_frames)):
        rc, fb, _ = orc.encode_frame(oopts, 44100, 16, np.ascontiguousarray(frames[f].T), frame_number=f)
        assert rc == 0 and data[off[f]:off[f + 1]] == fb, f"frame {f}"
    vres, _ = an.verify_device(44100, 0)
    assert (vres.frames, vres.bad_structure, vres.bad_crc16, vres.frames_pcm_differs) == (n_frames, 0, 0, 0)
    an.close()
