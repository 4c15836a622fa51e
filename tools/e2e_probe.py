#!/usr/bin/env python3
"""Where the time of flacenc_encode_many goes (runs on the GPU box): per-stream wall, staging, GPU
and MD5 times for a few stream counts / batch sizes."""
import sys, os, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _pcm import synth_fast
from flac_codec_amd.encode import BatchEncoder, Options

per = synth_fast(5, 2, 24, 512 * 4096)
for n_streams, threads, batch, depth in [(64, 64, 1024, 2), (64, 16, 1024, 2), (64, 16, 256, 2), (64, 8, 256, 2), (64, 32, 256, 2), (16, 16, 256, 2)]:
    be = BatchEncoder(Options.best().batch_frames(batch).pipeline_depth(depth), threads=threads)
    streams = [per] * n_streams
    be.encode(streams, 48000, 24, 2, copy=False)
    ts = []
    for _ in range(7):
        t = time.perf_counter(); be.encode(streams, 48000, 24, 2, copy=False); ts.append(time.perf_counter() - t)
    j = be.last_jobs
    med = lambda k: round(statistics.median(x[k] for x in j), 1)
    print(f"streams {n_streams} threads {threads} batch {batch} depth {depth}: call ms {[round(x*1e3) for x in ts]} "
          f"-> median {n_streams*per.size/statistics.median(ts)/1e9:.2f} best {n_streams*per.size/min(ts)/1e9:.2f} Gsamples/s; "
          f"per stream (last call) elapsed {med('elapsed_ms')} pack {med('pack_ms')} gpu {med('gpu_ms')} md5 {med('md5_ms')} "
          f"max start {max(x['start_ms'] for x in j):.1f} max elapsed {max(x['elapsed_ms'] for x in j):.1f}")
