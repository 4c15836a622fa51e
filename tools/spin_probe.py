import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from _pcm import synth_fast
from flac_codec_amd.encode import BatchEncoder, Options
per = synth_fast(5, 2, 24, 2048 * 4096)
for shared in (False, True):
    be = BatchEncoder(Options.best().batch_frames(256).shared_md5(shared), threads=1)
    be.encode([per], 48000, 24, 2, copy=False)
    c0, t0 = time.process_time(), time.perf_counter()
    for _ in range(5):
        be.encode([per], 48000, 24, 2, copy=False)
    c1, t1 = time.process_time(), time.perf_counter()
    j = be.last_jobs[0]
    print(f"shared_md5={shared}: wall {(t1-t0)/5*1e3:.1f} ms  cpu {(c1-c0)/5*1e3:.1f} ms per stream; pack {j['pack_ms']:.1f} gpu {j['gpu_ms']:.1f} md5 {j['md5_ms']:.1f}")
