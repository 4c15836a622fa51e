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
