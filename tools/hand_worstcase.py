#!/usr/bin/env python3
"""What the residual hand-over costs where nothing is handed: 8192-frame batches of NOISE (every subframe VERBATIM), of a signal
FIXED predictors win (ramps), and of SURVEY's signal, one context back to back, with the hand-over and with FLACGPU_NO_HAND=1
(contexts read the knob when they are created).  Prints ms per batch and the handed fraction."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from _pcm import synth_fast  # noqa: E402
from flac_codec_amd.gpu import GpuAnalyzer  # noqa: E402

B, F = 4096, 8192


def inputs():
    rng = np.random.Generator(np.random.PCG64(5))
    noise = rng.integers(-(1 << 23), 1 << 23, size=F * B * 2, dtype=np.int64).astype(np.int32)
    t = np.arange(F * B, dtype=np.int64)
    ramp = np.empty((F * B, 2), dtype=np.int64)
    ramp[:, 0] = (t * 3) % 100003 + rng.integers(-1, 2, size=F * B)      # a slow ramp + 1 LSB of dither: FIXED order 2 wins
    ramp[:, 1] = (t * 2) % 70001 + rng.integers(-1, 2, size=F * B)
    ar2 = np.tile(synth_fast(11, 2, 24, B * 512), F // 512)
    return {"noise": noise, "ramp": ramp.reshape(-1).astype(np.int32), "ar2": np.ascontiguousarray(ar2, dtype=np.int32)}


def run(pcm, no_hand):
    if no_hand:
        os.environ["FLACGPU_NO_HAND"] = "1"
    else:
        os.environ.pop("FLACGPU_NO_HAND", None)
    an = GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, 24, 2, max_frames=F)
    d = torch.from_numpy(pcm).cuda()
    for _ in range(30):
        an.encode_device(d.data_ptr(), F, B, 0, 48000)
    torch.cuda.synchronize()
    an.wait() if hasattr(an, "wait") else None
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(40):
            an.encode_device(d.data_ptr(), F, B, 0, 48000)
        an.handed_subframes()        # (synchronises the context)
        best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
    h = an.handed_subframes()
    an.close()
    return best, h


if __name__ == "__main__":
    for name, pcm in inputs().items():
        a, ha = run(pcm, False)
        b, hb = run(pcm, True)
        print(f"{name:6s}: hand-over {a:.4f} ms per batch ({ha[0]} of {ha[1]} handed), FLACGPU_NO_HAND=1 {b:.4f} ms")
