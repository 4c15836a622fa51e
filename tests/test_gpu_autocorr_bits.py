"""The autocorrelation the device leaves for the Levinson stage (encode.rs:3403-3413: lag k = the left fold of
windowed[i] * windowed[i + k]) compared BIT FOR BIT with the oracle's, candidate by candidate -- not through the
quantised coefficients it leads to.  In particular the tiles of a Tukey window's flat middle, where the kernels fuse the
multiply and the add into one v_fma_f64 (the factors are integers there, the product is exact, so the fused rounding is the
reference's; Params::fma_t0), against the same kernels with FLACGPU_NO_AC_FMA=1."""
import ctypes as C

import numpy as np
import pytest

import _oracle as orc
from _pcm import synth_fast, synth_hi

pytestmark = pytest.mark.gpu
B = 4096
AC_LD = 36


def device_ac(an, rows):
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    out = np.empty((rows, AC_LD), dtype=np.float64)
    assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(an.device_buffer(6)), out.nbytes, 2) == 0
    return out


def candidates_of(frame, stereo):
    """The candidate sample rows of one frame in the device's order, before the wasted-bits shift."""
    if not stereo:
        return list(frame)
    l, r = frame[0].astype(np.int64), frame[1].astype(np.int64)
    return [l, r, (l + r) >> 1, l - r]


@pytest.mark.parametrize("ch,bps,lpc,window", [(2, 24, 12, (2, 0.5)), (2, 16, 8, (2, 0.5)), (8, 24, 12, (2, 0.5)), (2, 24, 32, (2, 0.5)),
                                               (1, 24, 12, (2, 0.5)), (3, 20, 12, (2, 0.5)), (2, 24, 12, (2, 0.1)), (2, 24, 12, (1, 0.0)),
                                               (2, 24, 12, (0, 0.0)), (6, 24, 10, (2, 0.9))])
def test_autocorrelation_is_the_references_bit_for_bit(monkeypatch, ch, bps, lpc, window):
    from flac_codec_amd.gpu import GpuAnalyzer

    n = 6
    x = (synth_hi(40 + ch + bps, ch, bps, B * n, sections=4) if ch <= 2 else synth_fast(40 + ch, ch, bps, B * n)).reshape(-1, ch).astype(np.int64)
    x[:, 0] = (x[:, 0] >> 2) << 2                    # wasted bits on channel 0 (the in-place kernels scale the sums instead)
    x[B:2 * B, ch - 1] = (1 << (bps - 1)) - 1        # rail DC: the largest products
    pcm = np.ascontiguousarray(x.astype(np.int32).reshape(-1))
    stereo = ch == 2
    ncand = 4 if stereo else ch
    w = orc.window(window[0], window[1], B)
    got = {}
    for fma in (True, False):
        if fma:
            monkeypatch.delenv("FLACGPU_NO_AC_FMA", raising=False)
        else:
            monkeypatch.setenv("FLACGPU_NO_AC_FMA", "1")
        an = GpuAnalyzer(B, 6, lpc, True, True, window[0], window[1], bps, ch, max_frames=n)
        an.analyze(pcm, n, B)
        an.stats()     # (synchronises)
        got[fma] = device_ac(an, n * ncand)[:, : lpc + 1].copy()
        an.close()
    assert np.array_equal(got[True].view(np.int64), got[False].view(np.int64))
    checked = 0
    for f in range(n):
        frame = x[f * B:(f + 1) * B].T
        for c, row in enumerate(candidates_of(frame, stereo)):
            row = np.asarray(row, dtype=np.int64)
            orv = int(np.bitwise_or.reduce(row))
            if orv == 0:
                continue                               # an all-zero candidate gets no LPC analysis
            wasted = (orv & -orv).bit_length() - 1
            want = orc.autocorrelate((row >> wasted).astype(np.float64) * w, lpc)
            have = got[True][f * ncand + c][: len(want)]
            assert np.array_equal(have.view(np.int64), want.view(np.int64)), (f, c, have, want)
            checked += 1
    assert checked >= n * ncand - 2


@pytest.mark.parametrize("ch,bps,last", [(2, 24, 1000), (2, 16, 4032), (4, 24, 2500), (1, 24, 64)])
def test_short_last_frame_keeps_its_own_window(ch, bps, last):
    """A batch whose last frame is short: the full frames go through the fused tiles of the full block's window, the last
    frame through a launch of its own with the window of its length (no fused tiles there) -- every row bit for bit."""
    from flac_codec_amd.gpu import GpuAnalyzer

    n, lpc = 5, 12
    total = (n - 1) * B + last
    x = synth_fast(70 + ch + bps, ch, bps, total).reshape(-1, ch).astype(np.int64)
    pcm = np.ascontiguousarray(x.astype(np.int32).reshape(-1))
    stereo = ch == 2
    ncand = 4 if stereo else ch
    an = GpuAnalyzer(B, 6, lpc, True, True, 2, 0.5, bps, ch, max_frames=n)
    an.analyze(pcm, n, last)
    an.stats()
    got = device_ac(an, n * ncand)[:, : lpc + 1].copy()
    an.close()
    for f in range(n):
        length = B if f < n - 1 else last
        w = orc.window(2, 0.5, length)
        frame = x[f * B:f * B + length].T
        for c, row in enumerate(candidates_of(frame, stereo)):
            row = np.asarray(row, dtype=np.int64)
            orv = int(np.bitwise_or.reduce(row))
            if orv == 0:
                continue
            wasted = (orv & -orv).bit_length() - 1
            want = orc.autocorrelate((row >> wasted).astype(np.float64) * w, lpc)
            have = got[f * ncand + c][: len(want)]
            assert np.array_equal(have.view(np.int64), want.view(np.int64)), (f, c)
