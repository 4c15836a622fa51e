"""Seeded random sweep over options x stream parameters x signal kinds: device frames (analysis +
assembly through the C ABI) must equal the oracle's, frame by frame.  The sweep mixes the block
lengths of the wave kernels with arbitrary ones, 1..8 channels, 8..32 bits, every LPC order,
partition orders 0..6, all channel-correlation modes and short last frames.  Every case is also
decoded back on the device and compared with its input (but for the frames the reference's own decoder
rejects: _compare.frames_the_reference_cannot_decode)."""
import os

import numpy as np
import pytest

import _oracle as orc
from _compare import frames_the_reference_cannot_decode, orc_options_for, planar_frames
from _pcm import synth_burst, synth_fast, synth_hi

pytestmark = pytest.mark.gpu

N_CASES = int(os.environ.get("FLAC_RANDOM_CASES", "80"))
SEED0 = int(os.environ.get("FLAC_RANDOM_SEED0", "0"))


def make_signal(rng, kind, channels, bps, n):
    full = 1 << (bps - 1)
    if kind == "synth":
        return synth_fast(int(rng.integers(1 << 30)), channels, bps, n)
    if kind == "noise":
        return rng.integers(-full, full, size=n * channels, dtype=np.int64).astype(np.int32)
    if kind == "silence":
        return np.zeros(n * channels, dtype=np.int32)
    if kind == "sparse":
        x = np.zeros(n * channels, dtype=np.int32)
        idx = rng.integers(0, x.size, size=max(1, x.size // 500))
        x[idx] = rng.integers(-full, full, size=idx.size, dtype=np.int64).astype(np.int32)
        return x
    if kind == "quiet":
        return rng.integers(-3, 4, size=n * channels, dtype=np.int64).astype(np.int32)
    if kind == "shifted":  # wasted bits
        sh = int(rng.integers(1, min(8, bps - 2)))
        base = synth_fast(int(rng.integers(1 << 30)), channels, bps - sh, n).astype(np.int64)
        return (base << sh).astype(np.int32)
    if kind == "sine":
        t = np.arange(n, dtype=np.float64)
        cols = [np.round((full - 1) * 0.9 * np.sin(2 * np.pi * (0.001 + 0.01 * c) * t + c)) for c in range(channels)]
        return np.stack(cols, axis=1).reshape(-1).astype(np.int32)
    raise ValueError(kind)


def case_of(seed):
    """The parameters and the PCM of one seeded case (tools/soak/diag_case.py replays a case through this)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    block = int(rng.choice([1024, 1152, 2048, 2304, 4096, 4096, 4096, 192, 576, 1000, 4608, 16, 333, 8192]))
    channels = int(rng.choice([1, 2, 2, 2, 3, 4, 6, 8]))
    bps = int(rng.choice([8, 12, 16, 16, 20, 24, 24, 32]))
    max_lpc = int(rng.choice([0, 1, 4, 8, 12, 12, 16, 17, 32]))
    max_lpc = min(max_lpc, 32)
    tz = (block & -block).bit_length() - 1
    max_po = int(rng.integers(0, 7))
    if min(tz, max_po) > 6:
        max_po = 6
    mid_side = bool(rng.integers(2))
    exhaustive = bool(rng.integers(2))
    window = [(0, 0.0), (1, 0.0), (2, 0.5), (2, 0.25)][int(rng.integers(4))]
    n_frames = int(rng.integers(1, 6))
    last = block if rng.integers(3) else int(rng.integers(1, block + 1))
    if block <= max_lpc:   # the reference needs n > order for LPC; keep the case meaningful
        max_lpc = 0
    kind = str(rng.choice(["synth", "synth", "synth", "noise", "silence", "sparse", "quiet", "shifted", "sine"]))
    if kind == "shifted" and bps < 12:
        kind = "synth"
    n = (n_frames - 1) * block + last
    # r04: a third of the "synth" cases become resonant AR(k) signals (tests/_pcm.py synth_hi), on which the encoder
    # chooses high LPC orders.  Decided by a generator of its own: the draws of `rng` -- and with them every seed the
    # earlier rounds pinned (306266 below) -- stay what they were.
    rng2 = np.random.Generator(np.random.PCG64(seed ^ 0x5EED0004))
    if kind == "synth" and bps >= 12 and rng2.integers(3) == 0:
        kind = "resonant"
        orders = [int(v) for v in rng2.integers(1, 33, size=4)]
        hseed = int(rng2.integers(1 << 30))
        if rng2.integers(4) == 0 and block >= 256:   # ... a quarter of them with blocks that end in full-scale noise
            kind = "burst"                           # (drawn after the signal's seed: the resonant cases stay what they were)
            pcm = synth_burst(hseed, channels, bps, n, block, burst=int(rng2.integers(4, 120)), order=orders[0])
        else:
            pcm = synth_hi(hseed, channels, bps, n, segment=max(16, block), orders=orders)
    elif kind == "synth" and rng2.integers(3) == 0:
        # ... and another third change their kind FRAME BY FRAME (silence, noise, wasted bits, resonant, quiet, a channel
        # that is a copy or the negative of its neighbour): what one frame decides must not leak into the next
        kind = "mixed"
        parts = []
        for f0 in range(0, n, block):
            m = min(block, n - f0)
            k2 = str(rng2.choice(["synth", "noise", "silence", "quiet", "sparse", "resonant", "shifted", "twin"]))
            r3 = np.random.Generator(np.random.PCG64(int(rng2.integers(1 << 30))))
            if k2 == "resonant" or (k2 == "shifted" and bps < 12):
                part = synth_hi(int(r3.integers(1 << 30)), channels, bps, m, segment=max(16, m), orders=[int(r3.integers(1, 33))])
            elif k2 == "twin":
                part = make_signal(r3, "synth", channels, bps, m).reshape(-1, channels).copy()
                for c in range(1, channels, 2):
                    part[:, c] = part[:, c - 1] if r3.integers(2) else np.clip(-part[:, c - 1].astype(np.int64), -(1 << (bps - 1)), (1 << (bps - 1)) - 1)
                part = part.reshape(-1)
            else:
                part = make_signal(r3, k2, channels, bps, m)
            parts.append(np.asarray(part, dtype=np.int32))
        pcm = np.concatenate(parts)
    else:
        pcm = make_signal(rng, kind, channels, bps, n)
    rate = int(rng.choice([8000, 44100, 48000, 96000, 192000, 12345]))
    first = int(rng.choice([0, 127, 128, 70000, (1 << 31) - 8]))
    desc = (f"seed {seed}: block {block} ch {channels} bps {bps} lpc {max_lpc} po {max_po} ms {mid_side} "
            f"ex {exhaustive} win {window} frames {n_frames} last {last} {kind} rate {rate} first {first}")
    return dict(block=block, channels=channels, bps=bps, max_lpc=max_lpc, max_po=max_po, mid_side=mid_side,
                exhaustive=exhaustive, window=window, n_frames=n_frames, last=last, kind=kind, rate=rate, first=first, desc=desc), pcm


def one_case(seed):
    from flac_codec_amd.gpu import GpuAnalyzer

    c, pcm = case_of(seed)
    block, channels, bps, max_lpc, max_po = c["block"], c["channels"], c["bps"], c["max_lpc"], c["max_po"]
    mid_side, exhaustive, window, n_frames, last = c["mid_side"], c["exhaustive"], c["window"], c["n_frames"], c["last"]
    rate, first, desc = c["rate"], c["first"], c["desc"]
    an = GpuAnalyzer(block, max_po, max_lpc, mid_side, exhaustive, window[0], window[1], bps, channels,
                     max_frames=n_frames)
    try:
        data, off = an.encode_frames(pcm, n_frames, last, first, rate)
        # the device decoder must give the PCM back (k_decode / k_decode_finish / k_crc<VERIFY>)
        res, _ = an.verify_device(rate, first)
        # (a last frame shorter than twice the predictor order can come out of the reference's partition search in
        # a shape its own decoder rejects: reproduced byte for byte, and flagged by the device decoder as well)
        _, subs, _ = an.fetch(n_frames, want_residuals=False)
        undecodable = frames_the_reference_cannot_decode(subs, n_frames, channels, block, last)
        assert (res.frames, res.bad_structure, res.bad_crc16) == (n_frames, len(undecodable), 0), desc
        if not undecodable:
            assert res.compared_pcm == 1 and res.frames_pcm_differs == 0 and res.samples_differ == 0, desc
            assert np.array_equal(an.fetch_decoded(n_frames, last), pcm), desc
        else:
            assert undecodable == {n_frames - 1} and last < 64, desc
    finally:
        an.close()
    oopts = orc_options_for(block, max_po, max_lpc, mid_side, exhaustive, window[0], window[1])
    for f, planar in enumerate(planar_frames(pcm, channels, block)):
        rc, fb, _ = orc.encode_frame(oopts, rate, bps, planar, frame_number=first + f)
        assert rc == 0, desc
        assert data[off[f]:off[f + 1]] == fb, f"frame {f} differs: {desc}"


@pytest.mark.parametrize("chunk", range(8))
def test_random_configs(chunk):
    per = (N_CASES + 7) // 8
    for i in range(per):
        one_case(SEED0 + 1000 * chunk + i)


def test_partition_corner_of_very_short_frames():
    """The reference's partition search on a 4-sample frame (tests/test_oracle_roundtrip.py, same name): the device
    encoder writes the reference's bytes, the device decoder rejects the frame as the reference's decoder does."""
    from flac_codec_amd.gpu import GpuAnalyzer

    left = np.array([-197, -412, -452, -473], dtype=np.int32)
    right = np.array([-60, -323, -396, -450], dtype=np.int32)
    pcm = np.stack([left, right], axis=1).reshape(-1)
    an = GpuAnalyzer(4608, 3, 32, False, True, 1, 0.0, 16, 2, max_frames=1)
    data, off = an.encode_frames(pcm, 1, 4, 130, 8000)
    assert data.hex() == "fff86488c282035802ff3bfe64fe3cfe2714ffbbffe9c1477c00804d"
    res, _ = an.verify_device(8000, 130)
    _, subs, _ = an.fetch(1, want_residuals=False)
    an.close()
    assert frames_the_reference_cannot_decode(subs, 1, 2, 4608, 4) == {0}
    assert (res.frames, res.bad_structure, res.bad_crc16) == (1, 1, 0)
    one_case(306266)   # the seed of the sweep that met it (three frames, the last one this one)


def test_rice_codes_of_exactly_33_bits_behind_an_empty_bit_register():
    """Seed 1264 of the r04 sweep (8 channels, 32-bit samples, a resonant signal: Rice parameters 27 / 28 with quotients
    of 4 / 5, i.e. codes of exactly 33 bits): the device decoder's one-refill route read the last bit of such a code -- the
    sign of the residual -- as zero when the code before it had emptied the bit register.  The encoder's bytes were the
    oracle's all along; the verifier flagged sound frames and the stand-alone decoder would have returned a wrong sample."""
    from flac_codec_amd.gpu import decode_stream

    one_case(1264)
    c, pcm = case_of(1264)
    ch0 = np.ascontiguousarray(pcm.reshape(-1, c["channels"])[3 * c["block"]:, 0])     # the subframe the diagnosis used
    oo = orc_options_for(c["block"], c["max_po"], c["max_lpc"], c["mid_side"], c["exhaustive"], c["window"][0], c["window"][1])
    rc, flac, _ = orc.encode_stream(oo, c["rate"], 32, 1, ch0, total_known=True)
    assert rc == 0
    out, info = decode_stream(flac)
    assert info.bad_frames == 0 and info.bad_crc16 == 0 and info.md5_status == 1 and np.array_equal(out, ch0)
