#!/usr/bin/env python3
"""Step-time probe (no parity checks): the bench's timed loop with a choice of input layout.
   python3 tools/kprobe.py [--layout interleaved|planar] [--contexts 3] [--steps 200] [--config 3]
Planar input with a block length that is a multiple of 4 is analysed in place (no K0 split, one OR pass),
which shows what the split costs inside the overlapped step."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layout", default="interleaved", choices=("interleaved", "planar"))
    ap.add_argument("--contexts", type=int, default=3)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--frames", type=int, default=bench.FRAMES)
    ap.add_argument("--chunk-msamples", type=int, default=-1, help="FLACGPU_TUNE_CHUNK_MSAMPLES of every context (-1: default)")
    ap.add_argument("--own-buffers", action="store_true", help="every context reads its own copy of the batch (as bench.py does)")
    a = ap.parse_args()
    import torch
    from flac_codec_amd.gpu import GpuAnalyzer, LAYOUT_INTERLEAVED, LAYOUT_PLANAR

    cfg = bench.CONFIGS[a.config]
    C, BPS, RATE = cfg["ch"], cfg["bps"], cfg["rate"]
    F, B = a.frames, bench.BLOCK
    pcm = bench.make_pcm(1000 + 16 * a.config, F, C, BPS)
    if a.layout == "planar":
        pcm = np.ascontiguousarray(pcm.reshape(F, B, C).transpose(0, 2, 1)).reshape(-1)
    d = torch.from_numpy(pcm).cuda()
    lay = LAYOUT_PLANAR if a.layout == "planar" else LAYOUT_INTERLEAVED
    ans = [GpuAnalyzer(B, cfg["po"], cfg["lpc"], True, True, 2, 0.5, BPS, C, max_frames=F) for _ in range(a.contexts)]
    streams = [torch.cuda.Stream() for _ in ans]
    if a.chunk_msamples >= 0:
        for an in ans:
            an.set_tuning(an.TUNE_CHUNK_MSAMPLES, a.chunk_msamples)
    bufs = [d] + [d.clone() for _ in range(len(ans) - 1)] if a.own_buffers else [d] * len(ans)
    n = [0]

    def step():
        i = n[0] % len(ans)
        n[0] += 1
        ans[i].encode_device(bufs[i].data_ptr(), F, B, 0, RATE, stream=streams[i].cuda_stream, layout=lay)

    t = time.perf_counter()
    while time.perf_counter() - t < 0.3:
        for _ in range(8):
            step()
        torch.cuda.synchronize()
    res = []
    for _ in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t) / a.steps * 1e3)
    data, off = ans[0].fetch_frames(F)
    print(f"layout={a.layout} contexts={a.contexts} config={a.config} chunk={a.chunk_msamples} own={a.own_buffers}: ms/step {min(res):.4f} (runs {[round(r, 4) for r in res]}), "
          f"bytes {off[F]}")


if __name__ == "__main__":
    main()
