#!/usr/bin/env python3
"""Many streams, host PCM -> .flac: flacenc_encode_many (a writer per stream) against flacenc_encode_many_coalesced (shared
analysis batches through the pinned ring), MD5 included, bytes compared.
   python3 tools/small_streams_probe.py [--threads T] [--json PATH] [STREAMSxFRAMES ...]     (default: a sweep 1 .. 512 blocks)"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from flac_codec_amd.encode import BatchEncoder, Options  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--json", default=None)
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--depth", type=int, default=0)
ap.add_argument("--only-coalesced", action="store_true")
ap.add_argument("shapes", nargs="*", default=["8192x1", "4096x2", "2048x4", "1024x8", "512x16", "256x32", "128x64", "64x128", "64x512"])
a = ap.parse_args()
C, B = 2, bench.BLOCK
pcm = bench.make_pcm(1048, 8192 + 512, C, 24)
out = []
for shape in a.shapes:
    n, f = (int(v) for v in shape.split("x"))
    step = max(1, (8192 + 512 - f) // n)
    streams = [pcm[i * step * B * C: (i * step + f) * B * C] for i in range(n)]
    rec = {"streams": n, "frames_per_stream": f}
    res = {}
    for name, kw in (("one_writer_per_stream", {}), ("coalesced", {"coalesce": True})):
        if a.only_coalesced and not kw:
            continue
        o = Options.best()
        if kw and a.batch:
            o = o.batch_frames(a.batch)
        if kw and a.depth:
            o = o.pipeline_depth(a.depth)
        enc = BatchEncoder(o, threads=a.threads, **kw)
        h = enc.prepare(streams, 48000, 24, C)     # the job array a C caller holds; timed: the C entry point alone
        enc.run(h)
        got = [bytes(v) for v in enc.results(h, copy=False)]
        ts = []
        for _ in range(a.reps):
            t = time.perf_counter()
            enc.run(h)
            ts.append(time.perf_counter() - t)
        res[name] = got
        rec[name] = {"median_ms": round(statistics.median(ts) * 1e3, 2), "best_ms": round(min(ts) * 1e3, 2),
                     "Msamples/s": round(n * f * B * C / statistics.median(ts) / 1e6, 1)}
        print(f"{name:24s} {n:5d} x {f:4d} frames: median {statistics.median(ts)*1e3:8.2f} ms  best {min(ts)*1e3:8.2f} ms  "
              f"{n * f * B * C / statistics.median(ts) / 1e6:9.1f} Msamples/s", flush=True)
    rec["byte_identical"] = a.only_coalesced or res["coalesced"] == res["one_writer_per_stream"]
    print("byte-identical" if rec["byte_identical"] else "BYTES DIFFER", flush=True)
    out.append(rec)
if a.json:
    with open(a.json, "w") as fh:
        json.dump(out, fh, indent=1)
