"""DIRECT input (Params::inter): interleaved i32 stereo PCM of whole blocks of a wave block length is analysed and
assembled in place -- no K0 split, the wasted-bits ORs come out of the autocorrelation kernel, whose sums
are scaled by 2^(-2 wasted) instead of being formed from shifted samples (kernels/autocorr.inc).  Every case
must give the oracle's bytes, the bytes of the K0 path (FLACGPU_NO_DIRECT), and survive the consumers that
need planar rows afterwards (verification against the input, residual rows for the host packer)."""
import numpy as np
import pytest

import _oracle as orc
from _compare import orc_options_for, planar_frames
from _pcm import synth_fast

pytestmark = pytest.mark.gpu
B = 4096


def encode(pcm, bps, max_lpc, monkeypatch, direct, mid_side=True, rate=48000, first=5, B=B, exhaustive=True, max_po=6):
    from flac_codec_amd.gpu import GpuAnalyzer

    if direct:
        monkeypatch.delenv("FLACGPU_NO_DIRECT", raising=False)
    else:
        monkeypatch.setenv("FLACGPU_NO_DIRECT", "1")
    n = pcm.size // (2 * B)
    an = GpuAnalyzer(B, max_po, max_lpc, mid_side, exhaustive, 2, 0.5, bps, 2, max_frames=n)
    data, off = an.encode_frames(pcm, n, B, first, rate)
    kernels_ms = None
    an.set_timing(True)
    an.analyze(pcm, n, B)
    kernels_ms = an.kernel_ms()
    an.set_timing(False)
    an.analyze(pcm, n, B)
    an.pack_device(first, rate)
    res, _ = an.verify_device(rate, first)
    assert (res.frames, res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (n, 0, 0, 0)
    plans, subs, resid = an.fetch(n, want_residuals=True)
    st = an.stats()
    an.close()
    return data, off, kernels_ms, (plans, subs, resid), st


def check(pcm, bps, monkeypatch, max_lpc=12, mid_side=True, rate=48000, B=B, exhaustive=True, max_po=6):
    from flac_codec_amd.gpu import host_pack_frames

    first = 5
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    n = pcm.size // (2 * B)
    d_data, d_off, d_ms, d_plans, _ = encode(pcm, bps, max_lpc, monkeypatch, True, mid_side, rate, first, B, exhaustive, max_po)
    k_data, k_off, k_ms, _, _ = encode(pcm, bps, max_lpc, monkeypatch, False, mid_side, rate, first, B, exhaustive, max_po)
    assert "k_deinterleave" not in d_ms and "k_deinterleave" in k_ms   # the split pass really was skipped
    assert d_off == k_off and d_data == k_data
    oopts = orc_options_for(B, max_po, max_lpc, mid_side, exhaustive)
    for f, planar in enumerate(planar_frames(pcm, 2, B)):
        rc, fb, _ = orc.encode_frame(oopts, rate, bps, planar, frame_number=first + f)
        assert rc == 0 and d_data[d_off[f]:d_off[f + 1]] == fb, f
    # residual rows fetched after a direct analysis (planar rows made on demand) feed the host packer
    plans, subs, resid = d_plans
    h_data, h_off = host_pack_frames(rate, bps, 2, first, n, B, plans, subs, resid, threads=2)
    assert h_off == d_off and h_data == d_data


def test_plain_and_order32(monkeypatch):
    check(synth_fast(900, 2, 24, B * 9), 24, monkeypatch)
    check(synth_fast(901, 2, 16, B * 5), 16, monkeypatch, max_lpc=8, rate=44100)
    check(synth_fast(902, 2, 24, B * 4), 24, monkeypatch, max_lpc=32, rate=96000)
    check(synth_fast(903, 2, 20, B * 3), 20, monkeypatch, mid_side=False)


def test_wasted_bits_per_candidate(monkeypatch):
    """Different trailing-zero counts on L, R, mid and side: the scaling by 2^(-2 wasted) is per candidate."""
    x = synth_fast(910, 2, 16, B * 6).reshape(-1, 2).astype(np.int64)
    both = (x << 5).astype(np.int32)                               # L, R, side: 5; mid: 4 or more
    check(both.reshape(-1), 24, monkeypatch)
    left = x.copy()
    left[:, 0] <<= 3                                               # L: 3, R: 0, mid / side: 0
    check(left.astype(np.int32).reshape(-1), 24, monkeypatch)
    odd = x.copy()
    odd[:, 0] = (odd[:, 0] << 2) | 2                               # L: 1 ... and a frame-dependent mix
    odd[:, 1] <<= 2
    check(odd.astype(np.int32).reshape(-1), 24, monkeypatch)
    eight = (x[: B * 3] << 8).astype(np.int32)                     # 16 significant bits in a 24-bit stream
    check(eight.reshape(-1), 24, monkeypatch, max_lpc=32)


def test_constant_silent_and_noise(monkeypatch):
    x = synth_fast(920, 2, 24, B * 6).reshape(-1, 2).copy()
    x[:, 1] = 0                                                    # R all zero: CONSTANT; side == L, mid = L >> 1
    check(x.reshape(-1), 24, monkeypatch)
    y = synth_fast(921, 2, 24, B * 6).reshape(-1, 2).copy()
    y[B:3 * B] = 0                                                 # two silent frames in the middle
    y[3 * B:4 * B, 0] = y[3 * B:4 * B, 1]                          # L == R: side all zero
    check(y.reshape(-1), 24, monkeypatch)
    rng = np.random.Generator(np.random.PCG64(922))
    check(rng.integers(-(1 << 23), 1 << 23, size=B * 2 * 3, dtype=np.int64).astype(np.int32), 24, monkeypatch)
    check(np.full(B * 2 * 2, -(1 << 23), dtype=np.int32), 24, monkeypatch)   # DC at the negative rail


def test_without_lpc(monkeypatch):
    """No LPC: no autocorrelation kernel to take the ORs from -- the candidate waves OR their own samples."""
    check(synth_fast(950, 2, 16, B * 7), 16, monkeypatch, max_lpc=0, rate=44100)
    x = synth_fast(951, 2, 16, B * 5).reshape(-1, 2).astype(np.int64)
    x[:, 0] <<= 4
    x[:, 1] <<= 2                                                  # L: 4, R: 2, mid: 1 (or more), side: 2
    x[B:2 * B] = 0
    x[2 * B:3 * B, 1] = 0
    check(x.astype(np.int32).reshape(-1), 24, monkeypatch, max_lpc=0)
    check(synth_fast(952, 2, 24, B * 3), 24, monkeypatch, max_lpc=0, mid_side=False)


def test_fast_channel_choice(monkeypatch):
    """The fast channel choice with direct input: the abs sums come from the interleaved pairs
    (k_stereo_stats_t<INTER>), the candidates' activity from the assignment; with and without LPC."""
    check(synth_fast(965, 2, 24, 4096 * 5), 24, monkeypatch, max_lpc=12, exhaustive=False)
    check(synth_fast(966, 2, 16, 4096 * 4), 16, monkeypatch, max_lpc=0, exhaustive=False, mid_side=False)
    check(synth_fast(968, 2, 16, 4096 * 4), 16, monkeypatch, max_lpc=8, exhaustive=False, mid_side=False, max_po=5)
    x = synth_fast(967, 2, 16, 4096 * 6).reshape(-1, 2).astype(np.int64)
    x[:, 0] <<= 3
    x[4096:8192] = 0
    x[8192:12288, 1] = x[8192:12288, 0]
    check(x.astype(np.int32).reshape(-1), 24, monkeypatch, max_lpc=0, exhaustive=False)
    check(x.astype(np.int32).reshape(-1), 24, monkeypatch, max_lpc=12, exhaustive=False)


@pytest.mark.parametrize("blk", [1024, 1152, 2048, 2304])
def test_shorter_wave_block_lengths(monkeypatch, blk):
    """1024 / 1152 / 2048 / 2304-sample blocks (Options::fast(), encode.rs:1635-1644): the persistent candidate kernel
    stages the interleaved frame in LDS as segments of whole consumer lanes (7 lanes = 63 pieces per segment at 1152
    samples, the last segment a single lane clamped inside the frame), k_frame64 reads 9 / 18 pieces per lane.  A small
    grid makes every workgroup walk several frames (the prefetch of the next frame behind the current one)."""
    monkeypatch.setenv("FLACGPU_CAND_GRID", "3")
    check(synth_fast(980 + blk, 2, 24, blk * 11), 24, monkeypatch, B=blk)
    check(synth_fast(981 + blk, 2, 16, blk * 7), 16, monkeypatch, max_lpc=8, rate=44100, B=blk, max_po=4)
    # Options::fast(): no LPC, no mid-side, channel choice by abs sums
    check(synth_fast(982 + blk, 2, 16, blk * 9), 16, monkeypatch, max_lpc=0, exhaustive=False, mid_side=False, B=blk)
    check(synth_fast(983 + blk, 2, 24, blk * 6), 24, monkeypatch, max_lpc=12, exhaustive=False, B=blk)
    x = synth_fast(984 + blk, 2, 16, blk * 8).reshape(-1, 2).astype(np.int64)
    x[:, 0] <<= 4
    x[:, 1] <<= 2                                                  # wasted bits: L 4, R 2, mid 1 (or more), side 2
    x[blk:2 * blk] = 0                                             # a silent frame
    x[2 * blk:3 * blk, 1] = 0                                      # R all zero
    x[3 * blk:4 * blk, 0] = x[3 * blk:4 * blk, 1]                  # side all zero
    check(x.astype(np.int32).reshape(-1), 24, monkeypatch, B=blk)
    check(x.astype(np.int32).reshape(-1), 24, monkeypatch, max_lpc=0, B=blk)
    # fast channel choice without LPC: two waves per frame (k_cand64p<..., PAIR>), every assignment in reach
    check(x.astype(np.int32).reshape(-1), 24, monkeypatch, max_lpc=0, exhaustive=False, mid_side=True, B=blk)
    check(x.astype(np.int32).reshape(-1), 24, monkeypatch, max_lpc=0, exhaustive=False, mid_side=False, B=blk)
    y = synth_fast(986 + blk, 2, 16, blk * 10).reshape(-1, 2).astype(np.int64)
    y[2 * blk:4 * blk, 1] = y[2 * blk:4 * blk, 0] + (y[2 * blk:4 * blk, 1] >> 6)    # R ~ L: left-side / mid-side frames
    y[5 * blk:7 * blk, 0] = -y[5 * blk:7 * blk, 1]                                  # L = -R: mid ~ 0
    check(y.astype(np.int32).reshape(-1), 24, monkeypatch, max_lpc=0, exhaustive=False, mid_side=True, B=blk)
    check(y.astype(np.int32).reshape(-1), 24, monkeypatch, max_lpc=0, exhaustive=False, mid_side=False, B=blk, max_po=3)
    monkeypatch.setenv("FLACGPU_NO_CAND_PAIR", "1")   # the four-wave kernel on the same input
    check(y.astype(np.int32).reshape(-1), 24, monkeypatch, max_lpc=0, exhaustive=False, mid_side=True, B=blk)
    monkeypatch.delenv("FLACGPU_NO_CAND_PAIR")
    monkeypatch.delenv("FLACGPU_CAND_GRID")
    rng = np.random.Generator(np.random.PCG64(985 + blk))
    check(rng.integers(-(1 << 23), 1 << 23, size=blk * 2 * 5, dtype=np.int64).astype(np.int32), 24, monkeypatch, B=blk)


def test_shorter_blocks_knob_and_last_short_frame(monkeypatch):
    """FLACGPU_NO_DIRECT_SHORT keeps the shorter block lengths on the split path (A/B runs); a batch that ends in a
    short frame is not eligible for direct input at any block length."""
    from flac_codec_amd.gpu import GpuAnalyzer

    blk = 1152
    pcm = synth_fast(990, 2, 16, blk * 6)
    for env, want_split in ((None, False), ("1", True)):
        if env:
            monkeypatch.setenv("FLACGPU_NO_DIRECT_SHORT", env)
        an = GpuAnalyzer(blk, 6, 0, False, False, 2, 0.5, 16, 2, max_frames=6)
        an.set_timing(True)
        an.analyze(pcm, 6, blk)
        assert ("k_deinterleave" in an.kernel_ms()) == want_split
        an.analyze(pcm[: 2 * (blk * 5 + 100)], 6, 100)
        assert "k_deinterleave" in an.kernel_ms()
        an.close()


def test_device_buffer_input(monkeypatch):
    """The bench's entry point: PCM already in HBM, owned by the caller (a torch tensor)."""
    import torch

    from flac_codec_amd.gpu import GpuAnalyzer

    monkeypatch.delenv("FLACGPU_NO_DIRECT", raising=False)
    n = 40
    pcm = synth_fast(930, 2, 24, B * n)
    try:
        d = torch.from_numpy(pcm).cuda()
    except RuntimeError as e:   # torch initialised after the library in this process does not always find the GPU
        pytest.skip(f"torch cannot use the GPU here: {e}")
    an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, 24, 2, max_frames=n)
    for first in (0, 1000):
        an.encode_device(d.data_ptr(), n, B, first, 48000)
        data, off = an.fetch_frames(n)
        oopts = orc_options_for(B, 6, 12, True, True)
        for f, planar in enumerate(planar_frames(pcm, 2, B)):
            rc, fb, _ = orc.encode_frame(oopts, 48000, 24, planar, frame_number=first + f)
            assert rc == 0 and data[off[f]:off[f + 1]] == fb, f
        res, _ = an.verify_device(48000, first)
        assert (res.frames, res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (n, 0, 0, 0)
    an.close()


def test_copy_input_tuning_frees_the_callers_buffer(monkeypatch):
    """FLACGPU_TUNE_COPY_INPUT: the context works from its own copy, so a streaming caller may refill its device
    buffer as soon as the submitted work has run -- the frames fetched, the residual rows and the device round trip
    afterwards are still those of the batch that was submitted (the direct input's lifetime rule)."""
    import torch

    from flac_codec_amd.gpu import GpuAnalyzer

    monkeypatch.delenv("FLACGPU_NO_DIRECT", raising=False)
    n = 12
    pcm = synth_fast(931, 2, 24, B * n)
    try:
        d = torch.from_numpy(pcm).cuda()
    except RuntimeError as e:
        pytest.skip(f"torch cannot use the GPU here: {e}")
    an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, 24, 2, max_frames=n)
    an.set_tuning(an.TUNE_COPY_INPUT, 1)
    an.encode_device(d.data_ptr(), n, B, 3, 48000)
    torch.cuda.synchronize()
    d.random_(-(1 << 23), 1 << 23)        # the caller recycles its buffer
    torch.cuda.synchronize()
    data, off = an.fetch_frames(n)
    oopts = orc_options_for(B, 6, 12, True, True)
    for f, planar in enumerate(planar_frames(pcm, 2, B)):
        rc, fb, _ = orc.encode_frame(oopts, 48000, 24, planar, frame_number=3 + f)
        assert rc == 0 and data[off[f]:off[f + 1]] == fb, f
    res, _ = an.verify_device(48000, 3)
    assert (res.frames, res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (n, 0, 0, 0)
    an.close()


def test_independent_channels_split_inside_the_autocorrelation(monkeypatch):
    """3 / 4 / 8 interleaved independent channels with LPC: k_autocorr4's producers split the batch into the planar
    rows themselves (no k_deinterleave_n launch); same bytes as the K0 path and as the oracle, different wasted bits
    per channel, an all-zero channel and a rail-DC channel included."""
    from flac_codec_amd.gpu import GpuAnalyzer

    for ch, bps, seed in ((8, 24, 950), (4, 16, 951), (3, 24, 952), (6, 20, 953)):
        n = 5
        x = synth_fast(seed, ch, bps, B * n).reshape(-1, ch).astype(np.int64)
        x[:, 0] = (x[:, 0] >> 3) << 3                       # three wasted bits on channel 0
        x[B:2 * B, 1] = 0                                   # an all-zero channel in frame 1
        x[2 * B:3 * B, ch - 1] = (1 << (bps - 1)) - 1      # rail DC in frame 2
        pcm = np.ascontiguousarray(x.astype(np.int32).reshape(-1))
        outs = {}
        for direct in (True, False):
            if direct:
                monkeypatch.delenv("FLACGPU_NO_DIRECT", raising=False)
            else:
                monkeypatch.setenv("FLACGPU_NO_DIRECT", "1")
            an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, bps, ch, max_frames=n)
            data, off = an.encode_frames(pcm, n, B, 11, 96000)
            an.set_timing(True)
            an.analyze(pcm, n, B)
            ms = {k: v for k, v in an.kernel_ms().items() if v > 0}
            an.close()
            outs[direct] = (data, off)
            assert ("k_deinterleave" in ms) == (not direct), (ch, direct, ms)
        assert outs[True] == outs[False]
        data, off = outs[True]
        oopts = orc_options_for(B, 6, 12, True, True)
        for f, planar in enumerate(planar_frames(pcm, ch, B)):
            rc, fb, _ = orc.encode_frame(oopts, 96000, bps, planar, frame_number=11 + f)
            assert rc == 0 and data[off[f]:off[f + 1]] == fb, (ch, f)


def test_mono_is_read_in_place(monkeypatch):
    """One channel: the interleaved buffer is the planar row -- analysed without the K0 copy; with LPC the ORs for the
    wasted bits come out of the autocorrelation as well (no pass of its own over the batch), without LPC from k_orbits;
    wasted bits, a silent frame and FLACGPU_NO_DIRECT (the copying path) give the same bytes."""
    from flac_codec_amd.gpu import GpuAnalyzer

    monkeypatch.delenv("FLACGPU_NO_DIRECT", raising=False)
    n = 6
    pcm = synth_fast(940, 1, 16, B * n)
    an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, 16, 1, max_frames=n)
    data, off = an.encode_frames(pcm, n, B, 7, 44100)
    oopts = orc_options_for(B, 6, 12, True, True)
    for f in range(n):
        rc, fb, _ = orc.encode_frame(oopts, 44100, 16, pcm[f * B:(f + 1) * B].reshape(1, B), frame_number=7 + f)
        assert rc == 0 and data[off[f]:off[f + 1]] == fb, f
    an.analyze(pcm, n, B)
    an.pack_device(7, 44100)
    res, _ = an.verify_device(44100, 7)
    assert (res.frames, res.bad_structure, res.bad_crc16) == (n, 0, 0)
    an.set_timing(True)
    an.analyze(pcm, n, B)
    assert "k_deinterleave" not in an.kernel_ms()       # neither a copy nor an OR pass
    an.close()
    x = (synth_fast(941, 1, 16, B * 5).astype(np.int64) << 3).astype(np.int32)   # three wasted bits
    x[B:2 * B] = 0
    outs = []
    for lpc in (12, 0):
        for env in (None, "1"):
            if env:
                monkeypatch.setenv("FLACGPU_NO_DIRECT", env)
            else:
                monkeypatch.delenv("FLACGPU_NO_DIRECT", raising=False)
            a2 = GpuAnalyzer(B, 6, lpc, True, True, 2, 0.5, 24, 1, max_frames=5)
            outs.append(a2.encode_frames(x, 5, B, 0, 48000))
            a2.close()
        assert outs[-1] == outs[-2]
        oo = orc_options_for(B, 6, lpc, True, True)
        d2, o2 = outs[-1]
        for f in range(5):
            rc, fb, _ = orc.encode_frame(oo, 48000, 24, x[f * B:(f + 1) * B].reshape(1, B), frame_number=f)
            assert rc == 0 and d2[o2[f]:o2[f + 1]] == fb, (lpc, f)


@pytest.mark.parametrize("ch,bps,n,seed", [(8, 24, 21, 960), (4, 24, 9, 961), (8, 16, 3, 962), (4, 12, 17, 963), (3, 24, 7, 964),
                                           (6, 24, 11, 965), (5, 20, 6, 966), (6, 8, 4, 967), (3, 4, 3, 968), (7, 24, 3, 969), (6, 16, 19, 970)])
def test_interleaved_channels_are_read_in_place(monkeypatch, ch, bps, n, seed):
    """3, 4, 6 / 8 interleaved channels (XPOSE): the candidate and subframe kernels fetch a whole frame (or half of a 6- or
    8-channel one) per workgroup straight from the interleaved batch (transposed through LDS), no planar row is written
    (4-bit samples and 5, 7 channels: the planar rows of k_autocorr4's producers, as before).  Same bytes as the planar-copy
    path (FLACGPU_NO_XPOSE), as the K0 path and as the oracle; frame counts that leave a tail behind the XCD-paired
    workgroup ids; and the consumers that want planar rows afterwards (verification, residual rows) still get them."""
    from flac_codec_amd.gpu import GpuAnalyzer, host_pack_frames

    x = synth_fast(seed, ch, bps, B * n).reshape(-1, ch).astype(np.int64)
    x[:, 1] = (x[:, 1] >> 2) << 2                       # two wasted bits on channel 1
    x[B:2 * B, ch - 2] = 0                              # a silent channel in frame 1
    x[2 * B:3 * B, 0] = -(1 << (bps - 1))               # rail DC in frame 2
    x[:, ch - 1] = np.random.Generator(np.random.PCG64(seed)).integers(-(1 << (bps - 1)), 1 << (bps - 1), size=x.shape[0])
    pcm = np.ascontiguousarray(x.astype(np.int32).reshape(-1))
    outs = {}
    for mode in ("xpose", "rows", "k0"):
        monkeypatch.delenv("FLACGPU_NO_XPOSE", raising=False)
        monkeypatch.delenv("FLACGPU_NO_DIRECT", raising=False)
        if mode == "rows":
            monkeypatch.setenv("FLACGPU_NO_XPOSE", "1")
        if mode == "k0":
            monkeypatch.setenv("FLACGPU_NO_DIRECT", "1")
        an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, bps, ch, max_frames=n)
        for rep in range(2):     # the second batch of a context meets the first one's leftovers
            data, off = an.encode_frames(pcm, n, B, 40 + rep, 192000)
            outs[(mode, rep)] = (data, off)
        res, _ = an.verify_device(192000, 41)
        assert (res.frames, res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (n, 0, 0, 0), mode
        if mode == "xpose":
            an.analyze(pcm, n, B)
            plans, subs, resid = an.fetch(n, want_residuals=True)
            h_data, h_off = host_pack_frames(192000, bps, ch, 41, n, B, plans, subs, resid, threads=2)
            assert (h_data, h_off) == outs[(mode, 1)]
        an.close()
    for rep in range(2):
        assert outs[("xpose", rep)] == outs[("rows", rep)] == outs[("k0", rep)], rep
    data, off = outs[("xpose", 0)]
    oopts = orc_options_for(B, 6, 12, True, True)
    for f, planar in enumerate(planar_frames(pcm, ch, B)):
        rc, fb, _ = orc.encode_frame(oopts, 192000, bps, planar, frame_number=40 + f)
        assert rc == 0 and data[off[f]:off[f + 1]] == fb, f


def test_batches_that_do_not_start_on_16_bytes_take_the_copying_path():
    """The in-place kernels fetch 16-byte pieces: a device batch that starts 4, 8 or 12 bytes into its allocation goes through
    K0 instead -- same bytes, whatever the channel count."""
    import torch

    from flac_codec_amd.gpu import GpuAnalyzer

    for ch, bps in ((2, 24), (8, 24), (4, 16), (1, 16), (6, 24)):
        n = 4
        pcm = np.ascontiguousarray(synth_fast(990 + ch, ch, bps, B * n))
        an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, bps, ch, max_frames=n)
        want = an.encode_frames(pcm, n, B, 3, 48000)
        for shift in (1, 2, 3):
            try:
                big = torch.zeros(pcm.size + 8, dtype=torch.int32, device="cuda")
            except RuntimeError as e:
                pytest.skip(f"torch cannot use the GPU here: {e}")
            big[shift:shift + pcm.size] = torch.from_numpy(pcm).cuda()
            view = big[shift:shift + pcm.size]
            assert view.data_ptr() % 16 == 4 * shift
            an.encode_device(view.data_ptr(), n, B, 3, 48000)
            torch.cuda.synchronize()
            assert an.fetch_frames(n) == want, (ch, shift)
        an.close()


@pytest.mark.parametrize("grid", ["1", "3", "64", "5000"])
def test_dynamic_turns_any_grid_and_batch_after_batch(monkeypatch, grid):
    """k_cand64p draws its frames from a per-launch counter (Params::turn_counter): whatever the number of resident
    workgroups -- fewer than frames, more than frames, one -- and batch after batch of different sizes on ONE context (the
    launch's last draw resets the counter), every frame is analysed exactly once: the oracle's bytes, and nothing drawn twice."""
    from flac_codec_amd.gpu import GpuAnalyzer

    monkeypatch.setenv("FLACGPU_CAND_GRID", grid)
    first, rate, bps = 11, 48000, 24
    pcm = synth_fast(990, 2, bps, B * 37)
    oopts = orc_options_for(B, 6, 12, True, True)
    want = [orc.encode_frame(oopts, rate, bps, planar, frame_number=first + f)[1] for f, planar in enumerate(planar_frames(pcm, 2, B))]
    an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, bps, 2, max_frames=37)
    for n, f0 in ((37, 0), (5, 3), (1, 36), (20, 17), (37, 0)):   # several launches, sizes on both sides of the grid
        data, off = an.encode_frames(pcm[f0 * 2 * B:(f0 + n) * 2 * B], n, B, first + f0, rate)
        for f in range(n):
            assert data[off[f]:off[f + 1]] == want[f0 + f], (grid, n, f0, f)
    an.close()
    # without LPC (the SELF instantiation: its next image is requested EARLY, the ticket is published at the first barrier)
    oopts0 = orc_options_for(B, 6, 0, True, True)
    an = GpuAnalyzer(B, 6, 0, True, True, 2, 0.5, bps, 2, max_frames=37)
    for n, f0 in ((37, 0), (2, 9)):
        data, off = an.encode_frames(pcm[f0 * 2 * B:(f0 + n) * 2 * B], n, B, first + f0, rate)
        for f in range(n):
            rc, fb, _ = orc.encode_frame(oopts0, rate, bps, list(planar_frames(pcm, 2, B))[f0 + f], frame_number=first + f0 + f)
            assert rc == 0 and data[off[f]:off[f + 1]] == fb, (grid, n, f0, f)
    an.close()
