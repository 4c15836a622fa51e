"""GPU parity tests proper: the HIP analysis path, called through the C ABI
(include/flacenc_gpu.h), against the CPU oracle on the same seeded inputs.  Bit-exact:
every decision (assignment, subframe type, order, coefficients, shift, partition order,
Rice parameters, exact bit counts) and every residual value must be identical."""
import numpy as np
import pytest

import _oracle as orc
from _compare import compare_frame, orc_options_for, planar_frames
from _pcm import generate_sine_2, read_raw, synth, synth_fast

pytestmark = pytest.mark.gpu


def run_case(pcm, channels, bps, block_size=4096, max_po=6, max_lpc=12, mid_side=True,
             exhaustive=True, window=(2, 0.5), rate=48000, layout="interleaved"):
    from flac_codec_amd.gpu import GpuAnalyzer, LAYOUT_INTERLEAVED, LAYOUT_PLANAR

    frames = planar_frames(pcm, channels, block_size)
    n_frames = len(frames)
    last = frames[-1].shape[1]
    an = GpuAnalyzer(block_size, max_po, max_lpc, mid_side, exhaustive, window[0], window[1], bps,
                     channels, max_frames=n_frames)
    if layout == "interleaved":
        plans, subs, res = an.analyze(pcm[: ((n_frames - 1) * block_size + last) * channels],
                                      n_frames, last, LAYOUT_INTERLEAVED)
    else:
        flat = np.concatenate([f.reshape(-1) for f in frames])
        plans, subs, res = an.analyze(flat, n_frames, last, LAYOUT_PLANAR)
    oopts = orc_options_for(block_size, max_po, max_lpc, mid_side, exhaustive, window[0], window[1])
    for f, planar in enumerate(frames):
        n = planar.shape[1]
        rc, _, oplan = orc.encode_frame(oopts, rate, bps, planar, frame_number=f)
        assert rc == 0
        compare_frame(plans[f], subs[f * channels:(f + 1) * channels], res[f], oplan, planar, n,
                      where=f"frame {f}")
    st = an.stats()
    an.close()
    return st


def test_stereo16_fixed_only():  # BASELINE config 2: L5-fixed
    run_case(synth_fast(20, 2, 16, 4096 * 12), 2, 16, max_po=5, max_lpc=0)


def test_stereo16_default():  # L5 = Options::default()
    run_case(synth_fast(21, 2, 16, 4096 * 12), 2, 16, max_po=5, max_lpc=8)


def test_stereo24_best():  # BASELINE config 3 (headline): L8 = Options::best()
    st = run_case(synth_fast(30, 2, 24, 4096 * 24), 2, 24)
    assert st.order_ties == 0


def test_8ch24_best():  # BASELINE config 4 shape
    run_case(synth_fast(40, 8, 24, 4096 * 4), 8, 24, rate=192000)


def test_stereo24_order32():  # BASELINE config 5: L8x
    run_case(synth_fast(50, 2, 24, 4096 * 8), 2, 24, max_lpc=32, rate=96000)


def test_mono_and_short_last_frame():
    run_case(synth(60, 1, 16, 4096 * 2 + 1234), 1, 16)
    run_case(synth(61, 2, 24, 4096 + 17), 2, 24)
    run_case(synth(62, 2, 16, 16), 2, 16)


def test_planar_layout():
    run_case(synth_fast(63, 2, 24, 4096 * 3), 2, 24, layout="planar")
    run_case(synth_fast(64, 3, 16, 4096 * 2 + 100), 3, 16, layout="planar")


def test_fast_preset():  # Options::fast(): block 1152, no LPC, no mid-side, abs-sum correlation
    run_case(synth_fast(70, 2, 16, 1152 * 9 + 5), 2, 16, block_size=1152, max_po=3, max_lpc=0,
             mid_side=False, exhaustive=False)
    run_case(synth_fast(71, 2, 16, 1152 * 5), 2, 16, block_size=1152, max_po=3, max_lpc=8,
             mid_side=True, exhaustive=False)


def test_fast_channel_choice_wide_samples():
    """31-bit stereo, channel choice by abs sums (encode.rs:2463-2674 sums in u64): near-full-scale
    anti-correlated channels make |side| reach 2^31 per sample, so partial sums held in 32 bits would
    wrap and could pick another assignment than the reference."""
    rng = np.random.Generator(np.random.PCG64(77))
    n = 4096 * 3 + 500
    l = rng.integers((1 << 30) - (1 << 27), (1 << 30) - 1, size=n, dtype=np.int64)
    l *= np.where(np.arange(n) % 7 < 4, 1, -1)                     # long runs at either rail
    r = -l + rng.integers(-(1 << 12), 1 << 12, size=n, dtype=np.int64)
    r = np.clip(r, -(1 << 30), (1 << 30) - 1)
    pcm = np.stack([l, r], axis=1).astype(np.int32).reshape(-1)
    for ms in (True, False):
        run_case(pcm, 2, 31, max_lpc=8, mid_side=ms, exhaustive=False)
        run_case(pcm, 2, 31, max_lpc=0, mid_side=ms, exhaustive=False, block_size=1152, max_po=3)
    # the same shape at 29 and 30 bits (the extreme case of the former 8-sample u32 partial sums)
    for bps in (29, 30):
        sh = 31 - bps
        run_case((pcm >> sh).astype(np.int32), 2, bps, max_lpc=8, exhaustive=False)


def test_no_mid_side_exhaustive():
    run_case(synth_fast(72, 2, 24, 4096 * 4), 2, 24, mid_side=False)


def test_wasted_bits_and_silence():
    pcm = read_raw("wasted-bits.raw", 16)
    run_case(pcm, 1, 16)
    z = np.zeros(4096 * 2 * 2, dtype=np.int32)
    run_case(z, 2, 16)
    half = synth_fast(73, 2, 16, 4096 * 2)
    half[4096 * 2:] = 0  # second frame silent
    run_case(half, 2, 16)
    run_case((synth_fast(74, 2, 16, 4096 * 2) << 3).astype(np.int32), 2, 24)  # 3 wasted bits


def test_noise_goes_verbatim_and_full_scale():
    rng = np.random.Generator(np.random.PCG64(75))
    run_case(rng.integers(-(1 << 23), 1 << 23, size=4096 * 2 * 2, dtype=np.int64).astype(np.int32), 2, 24)
    hi, lo = (1 << 23) - 1, -(1 << 23)
    pat = np.array(([hi, lo, hi, hi, lo, lo, 0] * 1200)[:8192], dtype=np.int32)
    run_case(pat, 2, 24)
    run_case(pat, 1, 24)


def test_32bps_and_8bps():
    rng = np.random.Generator(np.random.PCG64(76))
    run_case(rng.integers(-(1 << 31), 1 << 31, size=4096 * 2, dtype=np.int64).astype(np.int32), 2, 32)
    run_case((synth_fast(77, 2, 24, 4096 * 2).astype(np.int64) << 8).astype(np.int32), 2, 32)
    run_case((synth_fast(78, 2, 16, 4096 * 2) >> 8).astype(np.int32), 2, 8)


def test_sine_streams():  # tests/format.rs:776-1004 shape: tonal input stresses Levinson
    for bps in (16, 24):
        fs = float((1 << (bps - 1)) - 1)
        run_case(generate_sine_2(fs, 48000.0, 4096 * 4, 441.0, 0.0, 4410.0, 0.1, 1.3), 2, bps)
        run_case(generate_sine_2(fs, 44100.0, 4096 * 4, 441.0, 0.5, 441.0, 0.0, 1.0), 2, bps)


@pytest.mark.parametrize("bs", [16, 17, 32, 33, 192, 256, 576, 1000, 2304, 4608, 8192, 16384])
def test_block_sizes(bs):
    run_case(synth_fast(80 + bs, 2, 16, bs * 3 + bs // 2), 2, 16, block_size=bs, max_lpc=12)


@pytest.mark.parametrize("lpc", [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32])
def test_lpc_orders_small_blocks(lpc):  # tests/format.rs:84-134
    pcm = read_raw("noise32.raw", 8)
    for bs in (16, 21, 32, 33):
        run_case(pcm, 1, 8, block_size=bs, max_lpc=lpc)


def test_windows():
    pcm = synth_fast(90, 2, 16, 4096 * 2)
    run_case(pcm, 2, 16, window=(0, 0.0))
    run_case(pcm, 2, 16, window=(1, 0.0))
    run_case(pcm, 2, 16, window=(2, 0.25))
    run_case(pcm, 2, 16, window=(2, 1.5))


def test_roundtrip_fixture_files():  # tests/format.rs:207-435 inputs
    for ch in (1, 2, 4, 8):
        for bps in (8, 16, 24):
            run_case(read_raw(f"roundtrip-{ch}-{bps}-4777.raw", bps), ch, bps, max_po=5, max_lpc=8)


def test_mfma_autocorr_experiment_is_close_but_not_the_product_path():
    """The f64-MFMA autocorrelation experiment agrees with the exact path to rounding error,
    and running it leaves the context's exact results untouched."""
    from flac_codec_amd.gpu import GpuAnalyzer

    pcm = synth_fast(95, 2, 24, 4096 * 16)
    an = GpuAnalyzer(4096, 6, 12, True, True, 2, 0.5, 24, 2, max_frames=16)
    plans0, subs0, res0 = an.analyze(pcm, 16, 4096)
    r = an.experiment_mfma_autocorr()
    assert r["compared"] == 64 and r["max_rel_err"] < 1e-9
    plans1, subs1, res1 = an.fetch(16)
    assert bytes(subs0) == bytes(subs1) and np.array_equal(res0, res1)
    an.close()


def test_order_ties_are_redecided_on_the_host(monkeypatch):
    """LPC order choice near a tie (encode.rs:3656-3702): candidates whose two best estimates lie
    inside the band are re-decided on the host with its libm.  The band is widened to 1e-3 and the
    device's estimates are skewed by 1e-4 (test knobs of the context), so that the device alone picks
    other orders for many candidates: the output must still be the oracle's, because every skewed
    decision falls inside the band and is redone on the host."""
    import os
    import _oracle as orc
    from _compare import orc_options_for, planar_frames
    from flac_codec_amd.gpu import GpuAnalyzer

    # a first-order autoregressive signal: order 1 is the reference's choice, the higher orders lose
    # only by their header bits (a few 1e-4 of the estimate), and LPC beats FIXED either way
    from scipy.signal import lfilter

    rng = np.random.Generator(np.random.PCG64(9))
    chans = [np.clip(np.rint(lfilter([1.0], [1.0, -0.9], rng.uniform(-1500, 1500, size=4096 * 24))), -32768, 32767)
             for _ in range(2)]
    pcm = np.stack(chans, axis=1).astype(np.int32).reshape(-1)
    frames = planar_frames(pcm, 2, 4096)
    oopts = orc_options_for(4096, 6, 12, True, True)

    def encode(env):
        for k in ("FLACGPU_TIE_BAND", "FLACGPU_TIE_PERTURB"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        monkeypatch.setenv("FLACGPU_TEST_KNOBS", "1")   # the tie knobs are test-only: ignored without this
        an = GpuAnalyzer(4096, 6, 12, True, True, 2, 0.5, 16, 2, max_frames=len(frames))
        data, off = an.encode_frames(pcm, len(frames), 4096, 0, 48000)
        st = an.stats()
        an.close()
        return data, off, st

    # skewed device estimates, band too narrow to notice: the output differs from the oracle somewhere
    skew, off_s, st_s = encode({"FLACGPU_TIE_BAND": "1e-30", "FLACGPU_TIE_PERTURB": "-1e-3"})
    # skewed device estimates inside a wide band: everything that could differ is redone on the host
    data, off, st = encode({"FLACGPU_TIE_BAND": "1e-2", "FLACGPU_TIE_PERTURB": "-1e-3"})
    assert st.order_ties > 0 and st.order_ties_resolved == st.order_ties
    differs = 0
    for f, planar in enumerate(frames):
        rc, fb, _ = orc.encode_frame(oopts, 48000, 16, planar, frame_number=f)
        assert rc == 0
        assert data[off[f]:off[f + 1]] == fb, f"frame {f}: host re-decision did not restore the oracle's bytes"
        differs += skew[off_s[f]:off_s[f + 1]] != fb
    assert differs > 0, "the skew changed nothing: the test does not exercise the re-decision"
    # production settings: nothing flagged on this input, same bytes
    plain, off_p, st_p = encode({})
    assert st_p.order_ties == 0 and plain == data
