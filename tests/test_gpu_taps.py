"""Every tap-count instantiation of the in-place FIR is hit by a parity case.

The candidate kernels (`k_cand64p` / `k_cand64`, kernels/wave_cand_lpc.inc) and the frame assembly (`k_frame64`,
kernels/pack.inc) switch on the LPC order of a subframe into FIR instantiations of 2, 4, ... 16 taps (20, 24, 28, 32
above order 16, 4096-sample blocks only).  The usual test signals make the reference's order estimate
(encode.rs:3656-3702) choose a few orders only, so here every frame is an AR(k) process with k cycling over
1..max order (tests/_pcm.py synth_hi(orders=...)): the ORACLE's plans must show every order among the winning
subframes and more than one channel assignment, and the HIP path must give the oracle's bytes frame by frame --
at every wave block length, through the in-place (DIRECT) kernels and through the planar-row ones."""
import collections

import numpy as np
import pytest

import _oracle as orc
from _compare import orc_options_for, planar_frames
from _pcm import synth_hi

pytestmark = pytest.mark.gpu


def run_case(monkeypatch, B, bps, max_lpc, channels=2, reps=3, seed=77, rate=48000, direct=(True, False)):
    from flac_codec_amd.gpu import GpuAnalyzer

    orders = list(range(1, max_lpc + 1))
    n = len(orders) * reps
    pcm = synth_hi(seed, channels, bps, B * n, segment=B, orders=orders)
    oopts = orc_options_for(B, 6, max_lpc, True, True)
    first = 3
    expect = []
    won = collections.Counter()
    assignments = collections.Counter()
    for f, planar in enumerate(planar_frames(pcm, channels, B)):
        rc, fb, plan = orc.encode_frame(oopts, rate, bps, planar, frame_number=first + f)
        assert rc == 0
        expect.append(fb)
        assignments[plan.assignment] += 1
        for c in range(channels):
            if plan.sub[c].type == orc.SUB_LPC:
                won[plan.sub[c].order] += 1
    # the oracle's plans: every order wins somewhere, and (stereo) more than one channel assignment occurs
    assert sorted(won) == orders, f"block {B}: orders without a winning subframe: {sorted(set(orders) - set(won))}"
    if channels == 2:
        assert len(assignments) > 1, assignments
    for d in direct:
        if d:
            monkeypatch.delenv("FLACGPU_NO_DIRECT", raising=False)
        else:
            monkeypatch.setenv("FLACGPU_NO_DIRECT", "1")
        an = GpuAnalyzer(B, 6, max_lpc, True, True, 2, 0.5, bps, channels, max_frames=n)
        data, off = an.encode_frames(pcm, n, B, first, rate)
        for f in range(n):
            assert data[off[f]:off[f + 1]] == expect[f], f"block {B} direct={d}: frame {f} differs from the oracle"
        an.analyze(pcm, n, B)
        an.pack_device(first, rate)
        res, _ = an.verify_device(rate, first)
        assert (res.frames, res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (n, 0, 0, 0)
        an.close()
    return won, assignments


@pytest.mark.parametrize("B", [1024, 1152, 2048, 2304, 4096])
def test_orders_1_to_16_at_every_wave_block_length(monkeypatch, B):
    """SPL 16 / 18 / 32 / 36 / 64 (k_cand64p<SPL,16>, k_cand64<SPL,16>, k_frame64<128,SPL,16>): orders 1..16."""
    run_case(monkeypatch, B, 24, 16)


def test_orders_1_to_32_at_4096(monkeypatch):
    """k_cand64p<64,32> / k_frame64<128,64,32> and the deep autocorrelation: orders 1..32, the 20/24/28/32-tap FIRs."""
    won, _ = run_case(monkeypatch, 4096, 24, 32, reps=2, rate=96000)
    assert all(won[k] for k in range(17, 33))


def test_orders_16_bit_and_independent_channels(monkeypatch):
    """16-bit stereo (other bounds in fir_cannot_overflow), and 3 independent channels (k_cand64<64,16,false>,
    k_frame64<192>): orders 1..16 again."""
    run_case(monkeypatch, 4096, 16, 16, reps=2, seed=82, rate=44100)
    run_case(monkeypatch, 4096, 24, 16, channels=3, reps=2, seed=79)
    run_case(monkeypatch, 4096, 24, 12, channels=8, reps=2, seed=80, rate=192000, direct=(True,))


def test_fir_recheck_path(monkeypatch):
    """ResidualOverflow (encode.rs:3190-3197) is not tested per sample on the first pass: the fold of the residual
    rules it out (a lane's sum of folded residuals < 2^30) or the wave reports itself and the host has the candidate
    stage run again with the exact test (Params::check_fir).  No ordinary input gets there, so a TEST knob lowers the
    threshold until every candidate does: same bytes as the oracle, and as the build that always tests."""
    from flac_codec_amd.gpu import GpuAnalyzer

    B, n = 4096, 12
    pcm = synth_hi(91, 2, 24, B * n, sections=6)
    oopts = orc_options_for(B, 6, 12, True, True)
    expect = [orc.encode_frame(oopts, 48000, 24, planar, frame_number=f)[1]
              for f, planar in enumerate(planar_frames(pcm, 2, B))]

    def encode(env):
        for k in ("FLACGPU_FIR_SUSPECT_BITS", "FLACGPU_FIR_CHECK", "FLACGPU_TEST_KNOBS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, 24, 2, max_frames=n)
        data, off = an.encode_frames(pcm, n, B, 0, 48000)
        st = an.stats()
        an.close()
        return [data[off[f]:off[f + 1]] for f in range(n)], st

    plain, st = encode({})
    assert plain == expect and st.fir_recheck == 0 and st.fir_rechecked == 0
    low, st = encode({"FLACGPU_TEST_KNOBS": "1", "FLACGPU_FIR_SUSPECT_BITS": "4"})
    assert st.fir_recheck > 0 and st.fir_rechecked == st.fir_recheck
    assert low == expect
    always, st = encode({"FLACGPU_FIR_CHECK": "1"})
    assert always == expect and st.fir_recheck == 0


@pytest.mark.parametrize("bps,max_lpc,order,burst", [(24, 32, 32, 40), (24, 32, 32, 8), (24, 12, 12, 60), (16, 16, 16, 100)])
def test_bursts_that_make_the_predictor_run_wild(monkeypatch, bps, max_lpc, order, burst):
    """Blocks that end in full-scale noise behind a resonant signal (tests/_pcm.py synth_burst): the LPC residual reaches
    2^28 .. 2^30 over the burst, so the first analysis cannot rule a ResidualOverflow out from the folded residual for
    some candidates (flacgpu_stats.fir_recheck > 0 at order 32) and the host has the candidate stage run again with the
    exact test -- on ordinary input, no test knob.  Bytes: the oracle's, and the always-testing build's."""
    from _pcm import synth_burst
    from flac_codec_amd.gpu import GpuAnalyzer

    B, n = 4096, 10
    pcm = synth_burst(7 + burst, 2, bps, B * n, B, burst=burst, order=order)
    oopts = orc_options_for(B, 6, max_lpc, True, True)
    expect = [orc.encode_frame(oopts, 48000, bps, planar, frame_number=f)[1] for f, planar in enumerate(planar_frames(pcm, 2, B))]
    stats = {}
    for env in ({}, {"FLACGPU_FIR_CHECK": "1"}, {"FLACGPU_NO_DIRECT": "1"}):
        for k in ("FLACGPU_FIR_CHECK", "FLACGPU_NO_DIRECT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        an = GpuAnalyzer(B, 6, max_lpc, True, True, 2, 0.5, bps, 2, max_frames=n)
        data, off = an.encode_frames(pcm, n, B, 0, 48000)
        stats[tuple(env)] = an.stats()
        for f in range(n):
            assert data[off[f]:off[f + 1]] == expect[f], (env, f)
        res, _ = an.verify_device(48000, 0)
        assert (res.frames, res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (n, 0, 0, 0)
        an.close()
    st = stats[()]
    assert st.fir_rechecked == st.fir_recheck
    if (bps, max_lpc, burst) == (24, 32, 40):
        assert st.fir_recheck > 0, "this input is meant to reach the re-run"


@pytest.mark.parametrize("mode", ["0", "1", "2"])
@pytest.mark.parametrize("max_lpc", [12, 32])
def test_deferred_fixed_count_gives_the_same_bytes(monkeypatch, mode, max_lpc):
    """Params::defer_fixed (r05): the FIXED half's partition tree and exact bit count put off behind the LPC half and skipped
    where a lower bound of the FIXED size (from the 64 leaf sums) already exceeds the exact LPC size (encode.rs:2929-2934);
    the tree, a re-fetch of the samples and the count otherwise.  Never (0), by the LPC estimate (1, the default) and whenever
    LPC parameters exist (2: the late path for every undecided candidate) must all give the oracle's bytes; the counters show
    which paths ran."""
    import ctypes as C

    from _pcm import synth_fast
    from flac_codec_amd import _lib
    from flac_codec_amd.gpu import GpuAnalyzer

    monkeypatch.setenv("FLACGPU_DEFER_FIXED", mode)
    B, bps, n = 4096, 24, 24
    # resonant frames (LPC wins by a wide margin: the bound decides) next to plain ones (it does not: the re-fetch)
    hi = synth_hi(91, 2, bps, B * (n // 2), segment=B, orders=list(range(2, max_lpc + 1, 3)))
    lo = synth_fast(92, 2, bps, B * (n // 2))
    pcm = np.concatenate([hi, lo])
    oopts = orc_options_for(B, 6, max_lpc, True, True)
    an = GpuAnalyzer(B, 6, max_lpc, True, True, 2, 0.5, bps, 2, max_frames=n)
    data, off = an.encode_frames(pcm, n, B, 11, 48000)
    for f, planar in enumerate(planar_frames(pcm, 2, B)):
        rc, fb, _ = orc.encode_frame(oopts, 48000, bps, planar, frame_number=11 + f)
        assert rc == 0 and data[off[f]:off[f + 1]] == fb, f"mode {mode}: frame {f} differs from the oracle"
    st = an.stats()
    res, _ = an.verify_device(48000, 11)
    assert (res.frames, res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (n, 0, 0, 0)
    an.close()
    if mode == "0":
        assert st.fixed_decided == 0 and st.fixed_refetched == 0
    elif mode == "2":
        assert st.fixed_decided > 0 and st.fixed_refetched > 0, (st.fixed_decided, st.fixed_refetched)
        assert st.fixed_decided + st.fixed_refetched <= 4 * n
    else:
        assert st.fixed_decided > 0, "the estimate never chose to defer on resonant frames"
