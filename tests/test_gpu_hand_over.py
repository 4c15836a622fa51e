"""The residual handed from the candidate kernel to the frame kernel (Params::hand_meta, r06): for 4096-sample stereo frames read
in place with LPC on, the wave of `k_cand64p` that wins a subframe with an LPC candidate stores the folded residual it still
holds (t = r ^ (r >> 31) with r's sign in bit 31), and `k_frame64` reads that row back instead of fetching both channels,
picking, shifting and running the FIR of encode.rs:3174-3203 a second time.  Subframes won by FIXED / CONSTANT / VERBATIM, and
those whose wave had to re-fetch its samples for the exact FIXED count (Params::defer_fixed), carry no residual and take the
frame kernel's own path.  Everything here must give the ORACLE's bytes; the cases choose inputs and call sequences so that
both paths, their mixtures and the consumers that invalidate the hand-over are hit."""
import numpy as np
import pytest

import _oracle as orc
from _compare import orc_options_for, planar_frames
from _pcm import synth_fast, synth_hi

pytestmark = pytest.mark.gpu
B = 4096


def oracle_frames(pcm, bps, max_lpc, rate, first, exhaustive=True, mid_side=True):
    oopts = orc_options_for(B, 6, max_lpc, mid_side, exhaustive)
    out, types = [], []
    for f, planar in enumerate(planar_frames(pcm, 2, B)):
        rc, fb, plan = orc.encode_frame(oopts, rate, bps, planar, frame_number=first + f)
        assert rc == 0
        out.append(fb)
        types += [plan.sub[c].type for c in range(2)]
    return out, types


def gpu_frames(pcm, bps, max_lpc, rate, first, exhaustive=True, mid_side=True):
    from flac_codec_amd.gpu import GpuAnalyzer

    n = pcm.size // (2 * B)
    an = GpuAnalyzer(B, 6, max_lpc, mid_side, exhaustive, 2, 0.5, bps, 2, max_frames=n)
    data, off = an.encode_frames(pcm, n, B, first, rate)
    handed = an.handed_subframes()
    an.close()
    return [bytes(data[off[f]:off[f + 1]]) for f in range(n)], handed


def test_handed_where_lpc_wins_and_bytes_are_the_oracles(monkeypatch):
    monkeypatch.delenv("FLACGPU_NO_HAND", raising=False)
    pcm = np.ascontiguousarray(synth_fast(4100, 2, 24, B * 24), dtype=np.int32)
    want, types = oracle_frames(pcm, 24, 12, 48000, 7)
    got, (handed, subs, on) = gpu_frames(pcm, 24, 12, 48000, 7)
    assert got == want
    assert on and subs == 48
    n_lpc = sum(1 for t in types if t == orc.SUB_LPC)
    assert n_lpc >= 40                       # (the signal is made for LPC to win)
    assert 0 < handed <= n_lpc               # only LPC winners are handed over
    assert handed >= n_lpc * 3 // 4          # ... and nearly all of them on this input (none defers)
    monkeypatch.setenv("FLACGPU_NO_HAND", "1")
    got2, (h2, s2, on2) = gpu_frames(pcm, 24, 12, 48000, 7)
    assert got2 == want and not on2 and h2 == 0


@pytest.mark.parametrize("max_lpc,bps", [(8, 16), (12, 20), (32, 24)])
def test_mixed_winners_orders_and_widths(monkeypatch, max_lpc, bps):
    """High-order resonant input (model order and channel relation change every frame), with frames that FIXED, CONSTANT or
    VERBATIM subframes win spliced in: handed and own-path subframes inside one batch, inside one frame."""
    monkeypatch.delenv("FLACGPU_NO_HAND", raising=False)
    n = 20
    x = synth_hi(4200 + max_lpc, 2, bps, B * n, segment=B, orders=list(range(1, max_lpc + 1))).reshape(-1, 2).copy()
    rng = np.random.Generator(np.random.PCG64(4201))
    x[2 * B:3 * B, 1] = 0                                                       # CONSTANT right, side == left
    x[5 * B:6 * B] = rng.integers(-(1 << (bps - 1)), 1 << (bps - 1), size=(B, 2))   # noise: VERBATIM
    ramp = (np.arange(B) * 3 - 2000).astype(np.int64)
    x[8 * B:9 * B, 0] = ramp                                                    # a ramp: FIXED order 2 is exact
    x[8 * B:9 * B, 1] = ramp // 2
    x[11 * B:12 * B] <<= 3                                                      # wasted bits on every candidate
    x[11 * B:12 * B] = np.clip(x[11 * B:12 * B], -(1 << (bps - 1)), (1 << (bps - 1)) - 8) & ~7
    pcm = np.ascontiguousarray(x.reshape(-1), dtype=np.int32)
    want, types = oracle_frames(pcm, bps, max_lpc, 96000, 0)
    got, (handed, subs, on) = gpu_frames(pcm, bps, max_lpc, 96000, 0)
    for f in range(n):
        assert got[f] == want[f], f
    assert on and 0 < handed < subs
    assert len(set(types)) >= 3                                                 # the oracle's plans really are mixed


def test_every_candidate_defers_and_refetches(monkeypatch):
    """FLACGPU_DEFER_FIXED=2 puts the exact FIXED count off for every candidate with LPC parameters; the ones the bound does not
    decide re-fetch their samples -- their registers no longer hold the LPC residual, and a subframe they win is not handed."""
    monkeypatch.delenv("FLACGPU_NO_HAND", raising=False)
    pcm = np.ascontiguousarray(synth_fast(4300, 2, 16, B * 16), dtype=np.int32)
    want, _ = oracle_frames(pcm, 16, 12, 44100, 3)
    base, (h0, subs, on) = gpu_frames(pcm, 16, 12, 44100, 3)
    monkeypatch.setenv("FLACGPU_DEFER_FIXED", "2")
    got, (h2, _, on2) = gpu_frames(pcm, 16, 12, 44100, 3)
    assert base == want and got == want
    assert on and on2 and h2 < h0            # the re-fetching winners are the difference


def test_fast_channel_choice_and_no_mid_side(monkeypatch):
    monkeypatch.delenv("FLACGPU_NO_HAND", raising=False)
    pcm = np.ascontiguousarray(synth_fast(4400, 2, 24, B * 12), dtype=np.int32)
    for exhaustive, mid_side in ((False, True), (True, False), (False, False)):
        want, _ = oracle_frames(pcm, 24, 12, 48000, 1, exhaustive, mid_side)
        got, (handed, subs, on) = gpu_frames(pcm, 24, 12, 48000, 1, exhaustive, mid_side)
        assert got == want, (exhaustive, mid_side)
        assert on and handed > 0


def test_consumers_between_analysis_and_assembly(monkeypatch):
    """analyze -> fetch(residual rows) -> pack: the rows k_emit writes on demand take the place of the handed words, and the
    assembly that follows must not read them as such; analyze -> pack -> verify on the device; plans packed from the host."""
    from flac_codec_amd.gpu import GpuAnalyzer, host_pack_frames

    monkeypatch.delenv("FLACGPU_NO_HAND", raising=False)
    n, first, rate, bps = 10, 9, 48000, 24
    pcm = np.ascontiguousarray(synth_fast(4500, 2, bps, B * n), dtype=np.int32)
    want, _ = oracle_frames(pcm, bps, 12, rate, first)
    an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, bps, 2, max_frames=n)
    import torch

    d = torch.from_numpy(pcm).cuda()
    # (1) separate calls, nothing in between
    an.analyze_device(d.data_ptr(), n, B)
    assert an.handed_subframes()[2] and an.handed_subframes()[0] > 0
    an.pack_device(first, rate)
    res, _ = an.verify_device(rate, first)
    assert (res.frames, res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (n, 0, 0, 0)
    data, off = an.fetch_frames(n)
    assert [bytes(data[off[f]:off[f + 1]]) for f in range(n)] == want
    # (2) the residual rows fetched in between (flacgpu_analyze with a residual buffer does the same)
    an.analyze_device(d.data_ptr(), n, B)
    plans, subs, resid = an.fetch(n, want_residuals=True)
    assert not an.handed_subframes()[2]      # switched off for this batch
    an.pack_device(first, rate)
    res, _ = an.verify_device(rate, first)
    assert (res.frames, res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (n, 0, 0, 0)
    h_data, h_off = host_pack_frames(rate, bps, 2, first, n, B, plans, subs, resid, threads=2)
    assert [bytes(h_data[h_off[f]:h_off[f + 1]]) for f in range(n)] == want
    # (3) a whole call again: the hand-over is back
    data, off = an.encode_frames(pcm, n, B, first, rate)
    assert [bytes(data[off[f]:off[f + 1]]) for f in range(n)] == want
    assert an.handed_subframes()[2] and an.handed_subframes()[0] > 0
    # (4) plans that come from the host are assembled from the samples
    d2, o2 = an.pack_plans(pcm, n, B, plans, subs, first, rate)
    assert [bytes(d2[o2[f]:o2[f + 1]]) for f in range(n)] == want
    an.close()


def test_context_reused_across_batches_of_different_content(monkeypatch):
    """The flags are rewritten by every batch: a batch of noise (nothing handed) between two LPC batches."""
    from flac_codec_amd.gpu import GpuAnalyzer

    monkeypatch.delenv("FLACGPU_NO_HAND", raising=False)
    n, rate, bps = 8, 48000, 16
    a = np.ascontiguousarray(synth_fast(4600, 2, bps, B * n), dtype=np.int32)
    rng = np.random.Generator(np.random.PCG64(4601))
    noise = rng.integers(-(1 << 15), 1 << 15, size=B * 2 * n, dtype=np.int64).astype(np.int32)
    b = np.ascontiguousarray(synth_hi(4602, 2, bps, B * n, segment=B, orders=[3, 9, 12]), dtype=np.int32)
    an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, bps, 2, max_frames=n)
    for k, pcm in enumerate((a, noise, b, a[: 2 * B * 5])):
        m = pcm.size // (2 * B)
        want, _ = oracle_frames(pcm, bps, 12, rate, 100 * k)
        data, off = an.encode_frames(pcm, m, B, 100 * k, rate)
        assert [bytes(data[off[f]:off[f + 1]]) for f in range(m)] == want, k
        handed, subs, on = an.handed_subframes()
        assert on and subs == 2 * m
        assert (handed == 0) if k == 1 else (handed > 0)
    an.close()
