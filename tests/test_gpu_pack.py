"""Device-side frame assembly (k_layout / k_pack / k_crc) against (a) the host bit-packer of
the product and (b) the oracle's frame bytes -- all three must agree byte for byte."""
import numpy as np
import pytest

import _oracle as orc
from _compare import orc_options_for, planar_frames
from _pcm import generate_sine_2, read_raw, synth, synth_fast

pytestmark = pytest.mark.gpu


def run_case(pcm, channels, bps, block_size=4096, max_po=6, max_lpc=12, mid_side=True,
             exhaustive=True, rate=48000, first_frame=0):
    from flac_codec_amd.gpu import GpuAnalyzer, host_pack_frames

    frames = planar_frames(pcm, channels, block_size)
    n_frames, last = len(frames), frames[-1].shape[1]
    an = GpuAnalyzer(block_size, max_po, max_lpc, mid_side, exhaustive, 2, 0.5, bps, channels,
                     max_frames=n_frames)
    used = pcm[: ((n_frames - 1) * block_size + last) * channels]
    plans, subs, res = an.analyze(used, n_frames, last)
    an.pack_device(first_frame, rate)
    dev_bytes, dev_off = an.fetch_frames(n_frames)
    host_bytes, host_off = host_pack_frames(rate, bps, channels, first_frame, n_frames, block_size,
                                            plans, subs, res, threads=2)
    assert dev_off == host_off
    oopts = orc_options_for(block_size, max_po, max_lpc, mid_side, exhaustive)
    for f, planar in enumerate(frames):
        rc, fb, _ = orc.encode_frame(oopts, rate, bps, planar, frame_number=first_frame + f)
        assert rc == 0
        d = dev_bytes[dev_off[f]:dev_off[f + 1]]
        h = host_bytes[host_off[f]:host_off[f + 1]]
        assert h == fb, f"frame {f}: host packer differs from oracle"
        if d != fb:
            diff = next(i for i in range(min(len(d), len(fb))) if d[i] != fb[i]) if len(d) == len(fb) else -1
            raise AssertionError(f"frame {f}: device packer differs from oracle at byte {diff} "
                                 f"(len {len(d)} vs {len(fb)})")
    # one-call variant
    b2, o2 = an.encode_frames(used, n_frames, last, first_frame, rate)
    assert b2 == dev_bytes and o2 == dev_off
    an.close()


def test_headline_shape():
    run_case(synth_fast(200, 2, 24, 4096 * 20 + 333), 2, 24)


def test_16bit_default_and_fixed():
    run_case(synth_fast(201, 2, 16, 4096 * 6), 2, 16, max_po=5, max_lpc=8, rate=44100)
    run_case(synth_fast(202, 2, 16, 4096 * 6), 2, 16, max_po=5, max_lpc=0, rate=44100)


def test_multichannel_and_mono():
    run_case(synth_fast(203, 8, 24, 4096 * 3), 8, 24, rate=192000)
    run_case(synth_fast(204, 1, 16, 4096 * 3 + 10), 1, 16)
    run_case(synth_fast(205, 3, 20, 4096 * 2), 3, 20, rate=32000)


def test_order32_and_uncommon_rates():
    run_case(synth_fast(206, 2, 24, 4096 * 3), 2, 24, max_lpc=32, rate=96000)
    for rate in (11025, 12345, 100000, 65530, 700001, 384000):
        run_case(synth_fast(207, 2, 16, 1000 * 2), 2, 16, block_size=1000, rate=rate)


def test_frame_number_lengths():
    pcm = synth_fast(208, 2, 16, 576 * 4)
    for first in (0, 126, 0x7FE, 0xFFFE, 0x1FFFFE, 0x3FFFFFE, 0x7FFFFFFE, 0xFFFFFFFF0):
        run_case(pcm, 2, 16, block_size=576, first_frame=first)


def test_verbatim_constant_wasted_escape():
    rng = np.random.Generator(np.random.PCG64(209))
    run_case(rng.integers(-(1 << 23), 1 << 23, size=4096 * 4, dtype=np.int64).astype(np.int32), 2, 24)
    run_case(np.zeros(4096 * 4, dtype=np.int32), 2, 16)
    run_case(read_raw("wasted-bits.raw", 16), 1, 16)
    run_case(rng.integers(-(1 << 31), 1 << 31, size=4096 * 4, dtype=np.int64).astype(np.int32), 2, 32)
    # mostly silent with rare huge spikes: long unary runs / escaped partitions
    x = np.zeros(4096 * 4, dtype=np.int32)
    x[::997] = (1 << 22)
    x[5::1013] = -(1 << 22)
    run_case(x, 2, 24)
    run_case(x, 1, 24, max_lpc=0)


def test_largest_possible_stereo_frames():
    """Full-scale noise, every subframe VERBATIM: the frames k_frame64 builds in its LDS bit string are as large as they
    get.  With the exhaustive channel choice the LDS frame is sized for two bps-bit subframes (the chosen pair never
    exceeds L + R: frame_fb_words_exhaustive_stereo); the fast choice picks its pair before any bit count exists and may
    take a 25-bit side channel VERBATIM (frame_fb_words)."""
    rng = np.random.Generator(np.random.PCG64(213))
    for bps in (24, 16, 12):
        lo, hi = -(1 << (bps - 1)), 1 << (bps - 1)
        noise = rng.integers(lo, hi, size=4096 * 2 * 5, dtype=np.int64).astype(np.int32)
        for exhaustive in (True, False):
            run_case(noise, 2, bps, exhaustive=exhaustive)
            run_case(noise, 2, bps, exhaustive=exhaustive, mid_side=False, max_lpc=0)
    # anti-correlated rails: |side| needs bps + 1 bits on every sample
    l = np.where(rng.integers(0, 2, size=4096 * 4) == 1, (1 << 23) - 1, -(1 << 23)).astype(np.int64)
    r = -l - 1 + rng.integers(0, 2, size=l.size)
    pcm = np.stack([l, np.clip(r, -(1 << 23), (1 << 23) - 1)], axis=1).astype(np.int32).reshape(-1)
    for exhaustive in (True, False):
        run_case(pcm, 2, 24, exhaustive=exhaustive)


def test_block_sizes_and_short_tail():
    for bs in (16, 33, 192, 1152, 4608, 16384):
        run_case(synth_fast(210 + bs, 2, 16, bs * 2 + bs // 3), 2, 16, block_size=bs)
    run_case(read_raw("noise32.raw", 8), 1, 8, block_size=17, max_lpc=16)


def test_fast_preset_and_sines():
    run_case(synth_fast(220, 2, 16, 1152 * 5), 2, 16, block_size=1152, max_po=3, max_lpc=0,
             mid_side=False, exhaustive=False)
    run_case(generate_sine_2(8388607.0, 48000.0, 4096 * 3, 441.0, 0.0, 4410.0, 0.1, 1.3), 2, 24)


@pytest.mark.parametrize("channels,bps,max_po,max_lpc,exhaustive", [
    (2, 24, 6, 12, True),     # Options::best
    (2, 16, 5, 8, False),     # Options::default (fast channel correlation)
    (2, 16, 5, 0, True),      # FIXED only
    (1, 24, 6, 12, True),
    (3, 20, 6, 16, True),
    (8, 24, 6, 12, True),     # 5..8 channels: k_sub64 (one workgroup per subframe) on both ranges
    (6, 16, 5, 8, True),
])
def test_two_range_pipeline_matches_serial(channels, bps, max_po, max_lpc, exhaustive):
    """flacgpu_encode_device with FLACGPU_TUNE_TWO_RANGES cuts big batches of 4096-sample frames into
    two frame ranges on two HIP streams: same bytes/offsets as analyze_device + pack_device, and
    the frames either side of the cut (and the first and last) equal the oracle's."""
    from flac_codec_amd.gpu import GpuAnalyzer

    n_frames, block, rate, first = 304, 4096, 48000, 70000   # >= 256 frames -> pipelined; odd cut
    pcm = synth_fast(300 + channels, channels, bps, n_frames * block)
    an = GpuAnalyzer(block, max_po, max_lpc, True, exhaustive, 2, 0.5, bps, channels, max_frames=n_frames)
    plans_a, subs_a, _ = an.analyze(pcm, n_frames, block)
    an.pack_device(first, rate)
    ser_bytes, ser_off = an.fetch_frames(n_frames)
    an.set_two_ranges(True)
    an.set_tuning(an.TUNE_LAG_SPLIT, 2)   # autocorrelation lags over 2 waves instead of 4: same sums
    for _ in range(2):   # twice: the second call reuses every buffer
        pip_bytes, pip_off = an.encode_frames(pcm, n_frames, block, first, rate)
    plans_b, subs_b, _ = an.fetch(n_frames, want_residuals=False)
    assert pip_off == ser_off and pip_bytes == ser_bytes
    assert bytes(plans_a) == bytes(plans_b) and bytes(subs_a) == bytes(subs_b)
    frames = planar_frames(pcm, channels, block)
    oopts = orc_options_for(block, max_po, max_lpc, True, exhaustive)
    cut = ((n_frames // 2) + 15) & ~15
    for f in (0, cut - 1, cut, cut + 1, n_frames - 1):
        rc, fb, _ = orc.encode_frame(oopts, rate, bps, frames[f], frame_number=first + f)
        assert rc == 0 and pip_bytes[pip_off[f]:pip_off[f + 1]] == fb, f"frame {f}"
    an.close()


@pytest.mark.parametrize("block", [1024, 1152, 2048, 2304])
def test_wave_kernels_other_block_sizes(block):
    """The wave kernels (k_cand64 / k_frame64) are instantiated for 64 x {16, 18, 32, 36, 64}
    samples; every frame is compared with the oracle and the host packer (run_case)."""
    run_case(synth_fast(400 + block, 2, 24, block * 7), 2, 24, block_size=block)                      # best
    run_case(synth_fast(401 + block, 2, 16, block * 5 + 77), 2, 16, block_size=block, max_po=5,
             max_lpc=8, exhaustive=False, rate=44100)                                               # default + short last
    run_case(synth_fast(402 + block, 2, 16, block * 4), 2, 16, block_size=block, max_po=3, max_lpc=0,
             mid_side=False, exhaustive=False, rate=44100)                                          # Options::fast
    run_case(synth_fast(403 + block, 1, 24, block * 3), 1, 24, block_size=block, max_lpc=16)         # mono, order 16
    run_case(synth_fast(404 + block, 3, 20, block * 3), 3, 20, block_size=block, max_po=6, max_lpc=4)


@pytest.mark.parametrize("block", [1024, 1152, 2048, 2304, 4096])
def test_wave_kernels_edge_inputs(block):
    """Silence (CONSTANT), wasted bits, white noise / full-scale patterns (VERBATIM, escaped
    partitions), 8-bit, half-silent frames -- through the wave kernels at every block length they
    are instantiated for, every frame against the oracle."""
    rng = np.random.Generator(np.random.PCG64(500 + block))
    run_case(np.zeros(block * 2 * 2, dtype=np.int32), 2, 16, block_size=block)
    half = synth_fast(501 + block, 2, 16, block * 3)
    half[block * 2:block * 4] = 0
    run_case(half, 2, 16, block_size=block, max_po=5, max_lpc=8)
    run_case((synth_fast(502 + block, 2, 16, block * 2) << 3).astype(np.int32), 2, 24, block_size=block)
    run_case(rng.integers(-(1 << 23), 1 << 23, size=block * 2 * 2, dtype=np.int64).astype(np.int32), 2, 24,
             block_size=block)
    hi, lo = (1 << 23) - 1, -(1 << 23)
    pat = np.array(([hi, lo, hi, hi, lo, lo, 0] * (block * 2 // 7 + 1))[:block * 2], dtype=np.int32)
    run_case(pat, 2, 24, block_size=block)
    run_case(pat, 1, 24, block_size=block)
    run_case((synth_fast(503 + block, 2, 16, block * 2) >> 8).astype(np.int32), 2, 8, block_size=block)
    # a lone loud click in silence: tiny partitions with k = 0 next to an escaped one
    click = np.zeros(block * 2 * 2, dtype=np.int32)
    click[block + 10] = 30000
    click[block * 3 + 1] = -32768
    run_case(click, 2, 16, block_size=block)


def test_generic_kernels_at_wave_block_lengths(monkeypatch):
    """FLACGPU_NO_W64 / FLACGPU_NO_FRAME64 route 4096- and 1152-sample frames through the generic
    LDS kernels (k_fixed, k_fir, k_autocorr*, k_emit, k_pack, k_crc): same bytes."""
    monkeypatch.setenv("FLACGPU_NO_W64", "1")
    monkeypatch.setenv("FLACGPU_NO_FRAME64", "1")
    run_case(synth_fast(600, 2, 24, 4096 * 4 + 100), 2, 24)
    run_case(synth_fast(601, 2, 16, 1152 * 5), 2, 16, block_size=1152, max_po=3, max_lpc=0, mid_side=False,
             exhaustive=False)
    run_case(synth_fast(602, 2, 24, 4096 * 3), 2, 24, max_lpc=32)


@pytest.mark.parametrize("knob", ["FLACGPU_NO_FRAME64", "FLACGPU_NO_FUSED_PACK"])
@pytest.mark.parametrize("channels", [3, 4, 6, 8])
def test_generic_packer_alone_on_multichannel_input(monkeypatch, knob, channels):
    """with the wave frame kernels switched off ALONE, 3/4/6/8 interleaved channels must not take the
    in-place (XPOSE) analysis -- the generic k_emit / k_pack read the planar rows, which that path never writes."""
    monkeypatch.setenv(knob, "1")
    run_case(synth_fast(610 + channels, channels, 24, 4096 * 3), channels, 24, rate=96000)


def test_resolve_entry_point_is_idempotent():
    """flacgpu_resolve: the explicit 'make the device buffers final' call for consumers of flacgpu_device_buffer."""
    from flac_codec_amd import _lib
    from flac_codec_amd.gpu import GpuAnalyzer

    pcm = synth_fast(620, 2, 24, 4096 * 4)
    an = GpuAnalyzer(4096, 6, 12, True, True, 2, 0.5, 24, 2, max_frames=4)
    want, off = an.encode_frames(pcm, 4, 4096, 0, 48000)
    assert _lib.lib().flacgpu_resolve(an._h) == 0 and _lib.lib().flacgpu_resolve(an._h) == 0
    got, off2 = an.fetch_frames(4)
    an.close()
    assert got == want and off == off2


def test_more_than_8192_frames_in_one_batch():
    """k_layout gives each of its 1024 lanes a run of frames; above 8192 frames the runs are longer
    than its register path (9 frames per lane here).  Spot frames against the oracle, everything
    through the device round trip (decode + CRC + PCM compare)."""
    from flac_codec_amd.gpu import GpuAnalyzer

    n_frames, block = 9000, 1024
    pcm = synth_fast(800, 2, 16, n_frames * block)
    an = GpuAnalyzer(block, 6, 8, True, True, 2, 0.5, 16, 2, max_frames=n_frames)
    data, off = an.encode_frames(pcm, n_frames, block, 3, 44100)
    oopts = orc_options_for(block, 6, 8, True, True)
    for f in (0, 1, 4499, 4500, 8191, 8192, n_frames - 1):
        planar = np.ascontiguousarray(pcm[f * block * 2:(f + 1) * block * 2].reshape(block, 2).T)
        rc, fb, _ = orc.encode_frame(oopts, 44100, 16, planar, frame_number=3 + f)
        assert rc == 0 and data[off[f]:off[f + 1]] == fb, f
    an.analyze(pcm, n_frames, block)
    an.pack_device(3, 44100)
    res, _ = an.verify_device(44100, 3)
    assert (res.frames, res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (n_frames, 0, 0, 0)
    an.close()


def test_pinned_host_buffers_give_the_same_frames():
    """flacgpu_encode_frames with both host buffers from flacgpu_host_alloc (the PCIe-inclusive figure of the bench):
    same bytes and offsets as from pageable arrays, call after call, with a short last frame."""
    from flac_codec_amd.gpu import GpuAnalyzer

    B = 4096
    pcm = synth_fast(4242, 2, 24, B * 6 + 1000)
    an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, 24, 2, max_frames=7)
    want, want_off = an.encode_frames(pcm, 7, 1000, 9, 48000)
    got, off, times = an.encode_frames_pinned(pcm, 7, 1000, 9, 48000, repeat=3)
    an.close()
    assert got == want and off == want_off and len(times) == 3


@pytest.mark.parametrize("channels", [2, 4, 8])
def test_pack_plans_of_wave_block_frames(channels):
    """flacgpu_pack_plans with the decisions an analysis made (fetched to the host and handed back): frames of 4096 samples take
    the wave assembly kernels -- k_frame64, or k_sub64 + k_sub_finish for 8 channels -- and must come out as the bytes of
    the ordinary path, with a short last frame (generic packer) behind them."""
    import ctypes as C

    from flac_codec_amd import _lib
    from flac_codec_amd._lib import FramePlan, SubframePlan
    from flac_codec_amd.gpu import GpuAnalyzer

    n, block, last = 9, 4096, 1500
    pcm = np.ascontiguousarray(synth_fast(640 + channels, channels, 24, (n - 1) * block + last))
    an = GpuAnalyzer(block, 6, 12, True, True, 2, 0.5, 24, channels, max_frames=n)
    want, want_off = an.encode_frames(pcm, n, last, 41, 96000)
    plans, subs, _ = an.analyze(pcm, n, last)
    L = _lib.lib()
    L.flacgpu_pack_plans.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_uint32, C.c_uint32, C.POINTER(FramePlan),
                                     C.POINTER(SubframePlan), C.c_uint64, C.c_uint32]
    rc = L.flacgpu_pack_plans(an._h, pcm.ctypes.data_as(C.POINTER(C.c_int32)), n, last, plans, subs, 41, 96000)
    assert rc == 0, L.flacgpu_last_error()
    got, got_off = an.fetch_frames(n)
    assert got_off == want_off and got == want
    an.close()


@pytest.mark.parametrize("channels,bps,max_lpc,n_frames,chunk_m", [(8, 24, 12, 700, 8), (2, 24, 12, 1500, 4), (4, 16, 8, 1100, 6),
                                                                   (3, 24, 12, 900, 4), (6, 24, 12, 600, 8)])
def test_batches_cut_into_ranges_give_the_same_bytes(channels, bps, max_lpc, n_frames, chunk_m):
    """flacgpu_encode_device runs a big in-place batch range by range (FLACGPU_TUNE_CHUNK_MSAMPLES; default 64 Mi samples):
    the whole kernel chain per range, frame offsets continuing from the range before.  With a small range size the same
    happens to a test-sized batch: same bytes, offsets and counters as the uncut batch, batch after batch; and the verifier
    and the order-tie / FIR re-checks, which work on the whole batch afterwards, still see all of it."""
    import torch

    from flac_codec_amd.gpu import GpuAnalyzer

    block = 4096
    pcm = np.ascontiguousarray(synth_fast(800 + channels, channels, bps, n_frames * block))
    try:
        d = torch.from_numpy(pcm).cuda()
    except RuntimeError as e:
        pytest.skip(f"torch cannot use the GPU here: {e}")
    whole = GpuAnalyzer(block, 6, max_lpc, True, True, 2, 0.5, bps, channels, max_frames=n_frames)
    whole.set_tuning(whole.TUNE_CHUNK_MSAMPLES, 0)
    cut = GpuAnalyzer(block, 6, max_lpc, True, True, 2, 0.5, bps, channels, max_frames=n_frames)
    cut.set_tuning(cut.TUNE_CHUNK_MSAMPLES, chunk_m)
    assert n_frames * block * channels > 1.5 * (chunk_m << 20)
    if channels == 2:    # stereo batches are cut without being asked, at 64 Mi samples: the default must not change bytes either
        dflt = GpuAnalyzer(block, 6, max_lpc, True, True, 2, 0.5, bps, channels, max_frames=n_frames)
        dflt.encode_device(d.data_ptr(), n_frames, block, 0, 96000)
        torch.cuda.synchronize()
        whole.encode_device(d.data_ptr(), n_frames, block, 0, 96000)
        torch.cuda.synchronize()
        assert dflt.fetch_frames(n_frames) == whole.fetch_frames(n_frames)
        dflt.close()
    for call in range(2):
        whole.encode_device(d.data_ptr(), n_frames, block, 77 * call, 96000)
        cut.encode_device(d.data_ptr(), n_frames, block, 77 * call, 96000)
        torch.cuda.synchronize()
        want = whole.fetch_frames(n_frames)
        got = cut.fetch_frames(n_frames)
        assert got[1] == want[1] and got[0] == want[0], call
    sw, sc = whole.stats(), cut.stats()
    assert (sw.order_ties, sw.fir_recheck) == (sc.order_ties, sc.fir_recheck)
    res, _ = cut.verify_device(96000, 77)
    assert (res.frames, res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (n_frames, 0, 0, 0)
    whole.close()
    cut.close()


@pytest.mark.parametrize("channels,block", [(8, 4096), (4, 4096), (3, 4096), (6, 4096), (2, 4096), (2, 1152), (1, 4096), (5, 4096)])
def test_frame_headers_of_every_frame_kernel(channels, block):
    """The frame header (sync, codes, UTF-8-like frame number of 1..7 bytes, block-size / sample-rate tails, CRC-8) is
    written by one lane from registers in every frame kernel -- k_frame64 (1..4 channels, in place and from rows),
    k_sub64 (one wave or four per workgroup) --: frame numbers at every length boundary and rates with 0-, 1- and 2-byte
    tails against the oracle, for the channel counts that select each of them."""
    pcm = synth_fast(230 + channels, channels, 24, block * 3)
    for first, rate in ((0, 48000), (0x7E, 44100), (0x7FE, 192000), (0xFFFE, 12345), (0x1FFFFE, 352800), (0x3FFFFFE, 655350),
                        (0x7FFFFFFE, 7000), (0xFFFFFFFF0, 96000)):
        run_case(pcm, channels, 24, block_size=block, first_frame=first, rate=rate)
