"""Shared PCM input generators for the tests and bench (integer-only where stated)."""
import os

import numpy as np

REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refdata")


def read_raw(name, bps):
    """Little-endian signed PCM -> int32 array (interleaved)."""
    raw = np.fromfile(os.path.join(REF, name), dtype=np.uint8)
    b = (bps + 7) // 8
    raw = raw[: raw.size - raw.size % b].reshape(-1, b).astype(np.uint32)
    v = np.zeros(raw.shape[0], dtype=np.uint32)
    for i in range(b):
        v |= raw[:, i] << np.uint32(8 * i)
    shift = 32 - 8 * b
    return ((v << np.uint32(shift)).astype(np.int32) >> shift).astype(np.int32)


def generate_sine_1(full_scale, sample_rate, samples, f1, a1, f2, a2):
    """tests/format.rs:687-711 (same arithmetic; used only as realistic input)."""
    d1 = 2.0 * np.pi / (sample_rate / f1)
    d2 = 2.0 * np.pi / (sample_rate / f2)
    k = np.arange(samples, dtype=np.float64)
    val = a1 * np.sin(k * d1) + a2 * np.sin(k * d2) * full_scale
    return np.trunc(val).astype(np.int64).clip(-2**31, 2**31 - 1).astype(np.int32)


def generate_sine_2(full_scale, sample_rate, samples, f1, a1, f2, a2, fmult):
    """tests/format.rs:713-742, interleaved stereo."""
    d1 = 2.0 * np.pi / (sample_rate / f1)
    d2 = 2.0 * np.pi / (sample_rate / f2)
    k = np.arange(samples, dtype=np.float64)
    t1, t2 = k * d1, k * d2
    c0 = a1 * np.sin(t1) + a2 * np.sin(t2) * full_scale
    c1 = -(a1 * np.sin(t1 * fmult)) + a2 * np.sin(t2 * fmult) * full_scale
    out = np.empty(samples * 2, dtype=np.float64)
    out[0::2], out[1::2] = c0, c1
    return np.trunc(out).astype(np.int64).clip(-2**31, 2**31 - 1).astype(np.int32)


def _pcg32(seed, n):
    """PCG32 (XSH-RR) stream, vectorised in blocks via the LCG jump: plain loop kept simple."""
    mult = 6364136223846793005
    inc = ((seed << 1) | 1) & 0xFFFFFFFFFFFFFFFF
    state = (seed + inc) & 0xFFFFFFFFFFFFFFFF
    out = np.empty(n, dtype=np.uint32)
    mask = 0xFFFFFFFFFFFFFFFF
    for i in range(n):
        old = state
        state = (old * mult + inc) & mask
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        out[i] = ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF
    return out


def synth(seed, channels, bps, n):
    """Integer-only synthetic music-like PCM: per channel a 2-pole resonator
    in Q15 driven by uniform noise plus LSB dither; right = 3/4 left + independent resonator.
    Returns interleaved int32 [n*channels].  numpy.random.Generator(PCG64) supplies the noise
    (seeded, reproducible on any box with the same numpy)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    amp = 1 << max(bps - 9, 1)
    dith = 1 << max(bps - 14, 0)
    lo, hi = -(1 << (bps - 1)), (1 << (bps - 1)) - 1
    chans = []
    for c in range(channels):
        e = rng.integers(-amp, amp + 1, size=n, dtype=np.int64)
        d = rng.integers(-dith, dith + 1, size=n, dtype=np.int64)
        y = np.zeros(n, dtype=np.int64)
        y1 = y2 = 0
        a1, a2 = 58000, -29491
        el = e.tolist()
        yl = [0] * n
        for i in range(n):
            v = ((a1 * y1 + a2 * y2) >> 15) + el[i]
            yl[i] = v
            y2, y1 = y1, v
        y = np.array(yl, dtype=np.int64) + d
        if c % 2 == 1:
            y = ((3 * chans[c - 1]) >> 2) + (y >> 1)
        chans.append(np.clip(y, lo, hi))
    out = np.empty(n * channels, dtype=np.int32)
    for c in range(channels):
        out[c::channels] = chans[c].astype(np.int32)
    return out


def synth_fast(seed, channels, bps, n):
    """Vectorised variant for large benches: AR(2) resonator via scipy.signal.lfilter in f64,
    rounded to integers, + dither; right correlated with left.  Deterministic for a given
    numpy/scipy; only used as INPUT (both sides of every comparison see the same array)."""
    from scipy.signal import lfilter

    rng = np.random.Generator(np.random.PCG64(seed))
    amp = float(1 << max(bps - 9, 1))
    dith = 1 << max(bps - 14, 0)
    lo, hi = -(1 << (bps - 1)), (1 << (bps - 1)) - 1
    chans = []
    for c in range(channels):
        e = rng.uniform(-amp, amp, size=n)
        y = lfilter([1.0], [1.0, -58000 / 32768.0, 29491 / 32768.0], e)
        y = np.rint(y).astype(np.int64) + rng.integers(-dith, dith + 1, size=n, dtype=np.int64)
        if c % 2 == 1:
            y = ((3 * chans[c - 1]) >> 2) + (y >> 1)
        chans.append(np.clip(y, lo, hi))
    out = np.empty(n * channels, dtype=np.int32)
    for c in range(channels):
        out[c::channels] = chans[c].astype(np.int32)
    return out


def _resonator_bank(rng, sections, rate_frac=(0.01, 0.45), radius=(0.90, 0.985), real_pole=False):
    """`sections` two-pole resonators with Q15 coefficients (a1 = 2 r cos(w), a2 = -r^2, both rounded to Q15):
    the poles of an AR(2 * sections) process.  Frequencies are spread over the band (one per equal slice, jittered)
    so that every section adds prediction gain of its own."""
    out = []
    lo, hi = rate_frac
    for k in range(sections):
        f = lo + (hi - lo) * (k + rng.uniform(0.15, 0.85)) / sections
        r = rng.uniform(*radius)
        a1 = int(round(2.0 * r * np.cos(2.0 * np.pi * f) * 32768.0))
        a2 = int(round(-r * r * 32768.0))
        out.append((a1, a2))
    if real_pole:   # one real pole more: an odd model order
        out.append((int(round(rng.choice([-1.0, 1.0]) * rng.uniform(0.6, 0.9) * 32768.0)), 0))
    return out


STEREO_MODES = ("corr", "indep", "anti", "near")


def synth_hi(seed, channels, bps, n, sections=6, segment=4096 * 8, modes=STEREO_MODES, orders=None):
    """High-order test/bench input: every channel is white noise through a CASCADE of `sections`
    Q15 two-pole resonators -- an AR(2 * sections) process, which the reference's order estimate (encode.rs:3656-3702)
    answers with LPC orders near 2 * sections instead of synth()'s order 2 -- scaled to bps - 3 bits, rounded to integers,
    plus +-1 LSB dither.  The pole set and the number of sections (sections - sections // 3 .. sections) change every
    `segment` samples (so frames differ in their coefficients and orders), and
    so does the relation of an odd channel to the even one before it (`modes`: 3/4 correlated, independent,
    anti-correlated, nearly equal), so that more than one channel assignment wins.  Vectorised (scipy lfilter in f64 on
    integer-valued input); deterministic for a given numpy/scipy and only ever used as INPUT.  `orders`: the AR model
    order of segment s is orders[s % len(orders)] (a real pole is added for odd ones) -- the tap-count coverage test."""
    from scipy.signal import lfilter

    rng = np.random.Generator(np.random.PCG64(seed))
    lo, hi = -(1 << (bps - 1)), (1 << (bps - 1)) - 1
    target = float(1 << max(bps - 4, 2))
    nseg = (n + segment - 1) // segment

    def process(m, k):
        e = rng.integers(-32768, 32769, size=m + 512).astype(np.float64)
        y = e
        for a1, a2 in _resonator_bank(rng, k // 2, real_pole=bool(k & 1)):
            y = lfilter([1.0], [1.0, -a1 / 32768.0, -a2 / 32768.0], y)
        y = y[512:]                                   # the filters' start-up transient
        rms = float(np.sqrt(np.mean(y * y))) or 1.0
        return np.rint(y * (target / rms)).astype(np.int64)

    chans = [np.empty(n, dtype=np.int64) for _ in range(channels)]
    for s in range(nseg):
        a, b = s * segment, min(n, (s + 1) * segment)
        m = b - a
        mode = modes[int(rng.integers(0, len(modes)))]
        # the segment's model order: 2 * (sections - sections // 3 .. sections) poles
        k = 2 * int(rng.integers(sections - sections // 3, sections + 1))
        if orders is not None:     # the caller names the model order of every segment (any order, odd ones too)
            k = int(orders[s % len(orders)])
        for c in range(channels):
            y = process(m, k)
            if c % 2 == 1:
                left = chans[c - 1][a:b]
                if mode == "corr":
                    y = ((3 * left) >> 2) + (y >> 1)
                elif mode == "anti":
                    y = -((7 * left) >> 3) + (y >> 3)
                elif mode == "near":
                    y = left + (y >> 6)
            y = y + rng.integers(-1, 2, size=m, dtype=np.int64)
            chans[c][a:b] = np.clip(y, lo, hi)
    out = np.empty(n * channels, dtype=np.int32)
    for c in range(channels):
        out[c::channels] = chans[c].astype(np.int32)
    return out


def synth_burst(seed, channels, bps, n, block, burst=40, order=32):
    """A resonant AR(order) signal (synth_hi) whose blocks END in `burst` samples of full-scale noise.  The Tukey window
    hides the burst from the autocorrelation, so the predictor is the resonant one -- large quantised coefficients at a
    small shift --, and over the burst it predicts garbage: residuals of 2^28 .. 2^30 and, now and then, a prediction
    outside the i32 range (ResidualOverflow, encode.rs:3190-3197).  The input of the overflow-handling tests."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0xB0057))
    x = synth_hi(seed, channels, bps, n, segment=block, orders=[order]).reshape(-1, channels).copy()
    full = 1 << (bps - 1)
    for f0 in range(0, n, block):
        b = min(n, f0 + block)
        a = max(f0, b - burst)
        x[a:b] = rng.integers(-full, full, size=(b - a, channels), dtype=np.int64).astype(np.int32)
    return x.reshape(-1)
