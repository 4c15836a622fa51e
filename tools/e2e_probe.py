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
CASES = [(64, 64, 1024, 2), (64, 16, 1024, 2), (64, 16, 256, 2), (64, 8, 256, 2), (64, 32, 256, 2), (16, 16, 256, 2)]
if len(sys.argv) > 1:   # "streams,threads,batch,depth ..."
    CASES = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for n_streams, threads, batch, depth in CASES:
    be = BatchEncoder(Options.best().batch_frames(batch).pipeline_depth(depth), threads=threads)
    # PROBE_DISTINCT=1: every stream its own array (1 GB of source PCM for 64 streams: the staging reads DRAM, as in the
    # bench), otherwise one array 64 times (the staging reads cache)
    streams = [per.copy() for _ in range(n_streams)] if os.environ.get('PROBE_DISTINCT') else [per] * n_streams
    be.encode(streams, 48000, 24, 2, copy=False)
    ts = []
    def throttled():
        try:
            return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat"))}
        except Exception:
            return {}
    th0, c0 = throttled(), time.process_time()
    for _ in range(7):
        t = time.perf_counter(); be.encode(streams, 48000, 24, 2, copy=False); ts.append(time.perf_counter() - t)
    th1, c1 = throttled(), time.process_time()
    if os.environ.get("PROBE_THREADS"):
        import glob, collections
        agg = collections.Counter()
        for st in glob.glob("/proc/self/task/*/stat"):
            try:
                f = open(st).read()
                comm = f[f.index("(") + 1:f.rindex(")")]
                rest = f[f.rindex(")") + 2:].split()
                agg[comm] += (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK")
            except Exception:
                pass
        print("  cpu seconds of the LIVE threads by name (whole process life):", dict(agg.most_common(8)))
    print(f"  cpu s per call {(c1 - c0) / 7:.3f}; throttled periods +{th1.get('nr_throttled', 0) - th0.get('nr_throttled', 0)}, "
          f"throttled ms +{(th1.get('throttled_usec', 0) - th0.get('throttled_usec', 0)) / 1e3:.0f}")
    j = be.last_jobs
    med = lambda k: round(statistics.median(x[k] for x in j), 1)
    print(f"streams {n_streams} threads {threads} batch {batch} depth {depth}: call ms {[round(x*1e3) for x in ts]} "
          f"-> median {n_streams*per.size/statistics.median(ts)/1e9:.2f} best {n_streams*per.size/min(ts)/1e9:.2f} Gsamples/s; "
          f"per stream (last call) elapsed {med('elapsed_ms')} pack {med('pack_ms')} gpu {med('gpu_ms')} md5 {med('md5_ms')} "
          f"max start {max(x['start_ms'] for x in j):.1f} max elapsed {max(x['elapsed_ms'] for x in j):.1f}")
