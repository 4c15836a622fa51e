"""Regression tests of the r03 advisor findings that need a GPU: the caller's buffer under FLACGPU_TUNE_COPY_INPUT in
every layout / range split, FLACGPU_NO_DIRECT surviving a set_tuning(COPY_INPUT, 0), and the worst-case output bound
with caller-sized VORBIS_COMMENT bodies on incompressible audio."""
import numpy as np
import pytest

import _oracle as orc
from _compare import orc_options_for, planar_frames
from _pcm import synth_fast

pytestmark = pytest.mark.gpu
B = 4096


@pytest.mark.parametrize("layout", ["interleaved", "planar"])
@pytest.mark.parametrize("two_ranges", [False, True])
@pytest.mark.parametrize("via", ["tuning", "env_then_tuning_off"])
def test_copy_input_in_every_layout_and_range_split(monkeypatch, layout, two_ranges, via):
    """The caller overwrites d_pcm right after the stream has drained; what is fetched, verified against the input and
    re-decided afterwards must still be the submitted batch (include/flacenc_gpu.h, FLACGPU_TUNE_COPY_INPUT)."""
    import torch

    from flac_codec_amd.gpu import GpuAnalyzer

    n = 272                                   # >= 256 frames: the two-range split is eligible
    pcm = np.tile(synth_fast(941, 2, 24, B * 16), n // 16)
    planar_all = np.ascontiguousarray(pcm.reshape(n, B, 2).transpose(0, 2, 1)).reshape(-1)
    if via == "env_then_tuning_off":
        monkeypatch.setenv("FLACGPU_NO_DIRECT", "1")
    else:
        monkeypatch.delenv("FLACGPU_NO_DIRECT", raising=False)
    try:
        d = torch.from_numpy(planar_all if layout == "planar" else pcm).cuda()
    except RuntimeError as e:
        pytest.skip(f"torch cannot use the GPU here: {e}")
    an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, 24, 2, max_frames=n)
    if via == "tuning":
        an.set_tuning(an.TUNE_COPY_INPUT, 1)
    else:
        an.set_tuning(an.TUNE_COPY_INPUT, 0)   # must not cancel the environment's FLACGPU_NO_DIRECT=1
    an.set_two_ranges(two_ranges)
    an.encode_device(d.data_ptr(), n, B, 3, 48000, layout=1 if layout == "planar" else 0)
    torch.cuda.synchronize()
    d.random_(-(1 << 23), 1 << 23)            # the caller recycles its buffer
    torch.cuda.synchronize()
    data, off = an.fetch_frames(n)
    oopts = orc_options_for(B, 6, 12, True, True)
    for f, planar in enumerate(planar_frames(pcm[: 16 * B * 2], 2, B)):
        rc, fb, _ = orc.encode_frame(oopts, 48000, 24, planar, frame_number=3 + f)
        assert rc == 0 and data[off[f]:off[f + 1]] == fb, f
    res, _ = an.verify_device(48000, 3)
    assert (res.frames, res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (n, 0, 0, 0)
    an.close()


def test_worst_case_bytes_with_large_tags_on_noise():
    """flacenc_worst_case_bytes sizes flacenc_encode_many's output buffers: full-scale noise (every subframe VERBATIM)
    with a multi-KB VORBIS_COMMENT must fit, and be the oracle's stream."""
    from flac_codec_amd.encode import BatchEncoder, Options

    rng = np.random.Generator(np.random.PCG64(5))
    pcm = rng.integers(-(1 << 23), 1 << 23, size=B * 2 * 6 + 2 * 777, dtype=np.int64).astype(np.int32)
    tags = [f"COMMENT{i}=" + "x" * 1500 for i in range(8)] + ["TITLE=noise"]
    vendor = "a vendor string " * 40
    o = Options.best().comment(tags, vendor_string=vendor).no_padding()
    out = BatchEncoder(o, threads=2).encode([pcm, pcm[: B * 2 * 2]], 48000, 24, 2)
    for got, src in zip(out, (pcm, pcm[: B * 2 * 2])):
        rc, ref, _ = orc.encode_stream(orc.options("best", padding=-1), 48000, 24, 2, src, total_known=True,
                                       tags=tags, vendor=vendor)
        assert rc == 0 and got == ref
