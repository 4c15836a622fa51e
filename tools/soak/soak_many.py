"""flacenc_encode_many under random shapes: stream counts around its 64 open slots, ragged lengths (empty tails, single
frames, several batches), thread counts 1..48, three presets -- every finished stream must be the oracle's .flac.
`python3 tools/soak/soak_many.py [seconds]`"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _oracle as orc                      # noqa: E402
from _pcm import synth_fast                # noqa: E402
from flac_codec_amd.encode import BatchEncoder, Options   # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.Generator(np.random.PCG64(20261003))
t_end = time.time() + budget
rounds = streams_done = 0
cache = {}
while time.time() < t_end:
    preset = str(rng.choice(["fast", "default", "best"]))
    bps = int(rng.choice([16, 24]))
    block = 1152 if preset == "fast" else 4096
    n = int(rng.choice([3, 17, 63, 64, 65, 97, 130]))
    threads = int(rng.choice([1, 2, 3, 7, 16, 48]))
    bf = int(rng.choice([4, 16, 64]))
    lens = [int(rng.integers(1, 6 * bf * block // 4)) if rng.integers(4) else block * int(rng.integers(1, 9)) for _ in range(n)]
    streams = [synth_fast(int(rng.integers(1 << 20)) % 97 + 5000, 2, bps, ln) for ln in lens]
    opts = getattr(Options, preset)().batch_frames(bf)
    outs = BatchEncoder(opts, threads=threads).encode(streams, 48000, bps, 2)
    for s, o in zip(streams, outs):
        key = (preset, bps, s.size, int(s[:64].astype(np.int64).sum()), int(s[-64:].astype(np.int64).sum()))
        if key not in cache:
            rc, ref, _ = orc.encode_stream(orc.options(preset), 48000, bps, 2, s, total_known=True)
            assert rc == 0
            cache[key] = ref
        if o != cache[key]:
            print(f"MISMATCH preset {preset} bps {bps} streams {n} threads {threads} batch {bf} len {s.size}")
            sys.exit(1)
    rounds += 1
    streams_done += n
print(f"many-stream soak: {rounds} calls, {streams_done} streams, thread counts 1..48, mismatches 0")
