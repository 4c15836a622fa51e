#!/usr/bin/env python3
"""Many SMALL streams, host PCM -> .flac: flacenc_encode_many (a writer per stream) against flacenc_encode_many_coalesced
(shared analysis batches).   python3 tools/small_streams_probe.py [streams] [frames per stream] [threads]"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from flac_codec_amd.encode import BatchEncoder, Options  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
f = int(sys.argv[2]) if len(sys.argv) > 2 else 32
th = int(sys.argv[3]) if len(sys.argv) > 3 else 32
C, B = 2, bench.BLOCK
pcm = bench.make_pcm(1048, 8192, C, 24)
step = max(1, (8192 - f) // n)
streams = [pcm[i * step * B * C: (i * step + f) * B * C] for i in range(n)]
res = {}
for name, kw in (("one_writer_per_stream", {}), ("coalesced", {"coalesce": True})):
    enc = BatchEncoder(Options.best(), threads=th, **kw)
    got = [bytes(v) for v in enc.encode(streams, 48000, 24, C, copy=False)]
    ts = []
    for _ in range(7):
        t = time.perf_counter()
        enc.encode(streams, 48000, 24, C, copy=False)
        ts.append(time.perf_counter() - t)
    res[name] = got
    print(f"{name:24s} {n} x {f} frames: median {statistics.median(ts)*1e3:8.2f} ms  best {min(ts)*1e3:8.2f} ms  "
          f"{n * f * B * C / statistics.median(ts) / 1e6:9.1f} Msamples/s")
assert res["coalesced"] == res["one_writer_per_stream"]
print("byte-identical")
