#!/usr/bin/env python3
"""Soak of the residual hand-over (Params::hand_meta): 4096-sample stereo batches read in place with LPC on,
made of frames of random kinds -- resonant AR(k) of random order and channel relation, noise (VERBATIM), silence and one-sided
silence (CONSTANT), ramps (FIXED wins exactly), wasted bits, full-scale bursts (large residuals), near-rail DC -- under random
options (LPC order 1..32, bits per sample 8..24, mid-side / exhaustive on or off, deferral forced or not); every frame against the
ORACLE's bytes, through one reused context per option set.  usage: soak_hand.py [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _oracle as orc  # noqa: E402
from _compare import orc_options_for, planar_frames  # noqa: E402
from _pcm import synth_fast, synth_hi  # noqa: E402

B = 4096


def make_batch(rng, bps, n, max_lpc):
    lo, hi = -(1 << (bps - 1)), (1 << (bps - 1)) - 1
    x = np.zeros((n * B, 2), dtype=np.int64)
    for f in range(n):
        kind = int(rng.integers(10))
        sl = slice(f * B, (f + 1) * B)
        seed = int(rng.integers(1 << 30))
        if kind <= 3:
            order = int(rng.integers(1, max_lpc + 1))
            x[sl] = synth_hi(seed, 2, bps, B, segment=B, orders=[order]).reshape(-1, 2)
        elif kind == 4:
            x[sl] = synth_fast(seed, 2, bps, B).reshape(-1, 2)
        elif kind == 5:
            x[sl] = rng.integers(lo, hi + 1, size=(B, 2))
        elif kind == 6:
            x[sl] = synth_fast(seed, 2, bps, B).reshape(-1, 2)
            x[sl, int(rng.integers(2))] = 0 if rng.integers(2) else int(rng.integers(lo, hi + 1))
        elif kind == 7:
            ramp = (np.arange(B) * int(rng.integers(1, 4)) + int(rng.integers(-50, 50))) % (hi // 2 + 1)
            x[sl, 0] = ramp
            x[sl, 1] = ramp // 2 if rng.integers(2) else ramp
        elif kind == 8:
            sh = int(rng.integers(1, max(2, bps - 6)))
            y = synth_fast(seed, 2, max(4, bps - sh), B).reshape(-1, 2).astype(np.int64)
            x[sl] = y << sh
            if rng.integers(2):
                x[sl, 1] = y[:, 1] << max(0, sh - 1)
        else:
            y = synth_hi(seed, 2, bps, B, segment=B, orders=[int(rng.integers(1, max_lpc + 1))]).reshape(-1, 2).astype(np.int64)
            at = int(rng.integers(0, B - 64))
            y[at:at + 64] = rng.choice([lo, hi], size=(64, 2))      # a full-scale burst: residuals near the 32-bit edge
            x[sl] = y
    return np.ascontiguousarray(np.clip(x, lo, hi).reshape(-1), dtype=np.int32)


def main():
    from flac_codec_amd.gpu import GpuAnalyzer

    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.Generator(np.random.PCG64(int(os.environ.get("SOAK_SEED", "20260"))))
    t_end = time.time() + seconds
    batches = frames = handed_total = subs_total = mism = 0
    while time.time() < t_end:
        bps = int(rng.choice([8, 12, 16, 20, 24]))
        max_lpc = int(rng.choice([1, 2, 4, 8, 12, 16, 20, 32]))
        mid_side, exhaustive = bool(rng.integers(2)), bool(rng.integers(2))
        for k in ("FLACGPU_DEFER_FIXED",):
            os.environ.pop(k, None)
        mode = int(rng.integers(4))
        if mode == 0:
            os.environ["FLACGPU_DEFER_FIXED"] = "2"
        elif mode == 1:
            os.environ["FLACGPU_DEFER_FIXED"] = "0"
        nmax = int(rng.integers(1, 40))
        an = GpuAnalyzer(B, 6, max_lpc, mid_side, exhaustive, 2, 0.5, bps, 2, max_frames=nmax)
        oopts = orc_options_for(B, 6, max_lpc, mid_side, exhaustive)
        rate = int(rng.choice([44100, 48000, 96000, 192000]))
        for _ in range(int(rng.integers(1, 5))):
            n = int(rng.integers(1, nmax + 1))
            first = int(rng.integers(0, 1 << 20))
            pcm = make_batch(rng, bps, n, max_lpc)
            data, off = an.encode_frames(pcm, n, B, first, rate)
            h, s, on = an.handed_subframes()
            handed_total += h
            subs_total += s
            for f, planar in enumerate(planar_frames(pcm, 2, B)):
                rc, fb, _ = orc.encode_frame(oopts, rate, bps, planar, frame_number=first + f)
                if rc != 0 or bytes(data[off[f]:off[f + 1]]) != fb:
                    mism += 1
                    print(f"MISMATCH bps={bps} lpc={max_lpc} ms={mid_side} ex={exhaustive} mode={mode} frame {f} of {n}", flush=True)
            batches += 1
            frames += n
        an.close()
    print(f"soak_hand: {batches} batches, {frames} frames, {handed_total} of {subs_total} subframes handed, mismatches {mism}")
    return 1 if mism else 0


if __name__ == "__main__":
    sys.exit(main())
