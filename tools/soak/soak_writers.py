import os, sys, time, hashlib, numpy as np
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from flac_codec_amd.encode import FlacSampleWriter, Options
from _pcm import synth_fast
inputs = [(synth_fast(20 + i, ch, bps, 44100 * 30 * ch // ch * 1), ch, bps, o) for i, (ch, bps, o) in enumerate(
    [(2, 16, Options.best), (2, 24, Options.best), (1, 16, Options.default), (2, 16, Options.fast), (6, 24, Options.best)])]
def job(k):
    pcm, ch, bps, o = inputs[k % len(inputs)]
    w = FlacSampleWriter(None, o(), 44100, bps, ch, pcm.size - pcm.size % ch)
    n = pcm.size - pcm.size % ch
    step = [n, 100003 * ch, 4096 * ch * 1024 + ch][k % 3]
    for s in range(0, n, step): w.write(pcm[s:min(n, s + step)])
    w.finalize(); data = w.getvalue(); w.close()
    return k % len(inputs), hashlib.sha256(data).hexdigest()
ref = {}
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
rounds = bad = 0
while time.time() < t_end:
    with ThreadPoolExecutor(24) as ex:
        for k, h in ex.map(job, range(96)):
            if ref.setdefault(k, h) != h: bad += 1
    rounds += 1
print(f"writer soak: {rounds} rounds x 96 streams on 24 threads, mismatches {bad}")
