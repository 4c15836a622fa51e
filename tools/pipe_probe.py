#!/usr/bin/env python3
"""Sweep of the pipelined host -> host loop (flacgpu_pipeline_*): depth x batch size x upload width, Msamples/s and the
link rates they imply.  usage: tools/pipe_probe.py [--hi]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _pcm import synth_fast  # noqa: E402
from flac_codec_amd.gpu import PinnedBuffer, Pipeline  # noqa: E402

B, C, BPS = 4096, 2, 24
base = np.tile(synth_fast(5, C, BPS, B * 512), 16)


def run(frames, depth, width, batches):
    pcm = base[: frames * B * C]
    src = pcm.view(np.uint8) if width == 4 else np.ascontiguousarray(pcm.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3]).reshape(-1)
    pipe = Pipeline(B, 6, 12, True, True, 2, 0.5, BPS, C, max_frames=frames, depth=depth)
    bufs = [PinnedBuffer(src.size) for _ in range(depth)]
    for b in bufs:
        b.array[:] = src
    best, down = 0.0, 0
    for rep in range(4):
        t = time.perf_counter()
        for i in range(batches):
            if pipe.in_flight() == depth:
                down = pipe.retire(copy=False)[1]
            pipe.submit(bufs[i % depth].address, width, frames, B, 0, 48000)
        while pipe.in_flight():
            down = pipe.retire(copy=False)[1]
        dt = time.perf_counter() - t
        if rep:
            best = max(best, batches * pcm.size / dt / 1e6)
    pipe.close()
    for b in bufs:
        b.close()
    return best, down / pcm.size


for width in (4, 3):
    for frames in (512, 1024, 2048, 4096, 8192):
        for depth in (2, 3, 4, 6):
            if frames * depth > 8192 * 4:
                continue
            v, down = run(frames, depth, width, max(8, 32768 // frames))
            print(f"width {width} frames {frames:5d} depth {depth}: {v:8.0f} Msamples/s  up {v * width / 1e3:5.1f} GB/s down {v * down / 1e3:5.1f} GB/s", flush=True)
