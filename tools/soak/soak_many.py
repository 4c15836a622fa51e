"""flacenc_encode_many (or, with `coalesce` as the second argument, flacenc_encode_many_coalesced) under random shapes: stream counts around its 64 open slots, ragged lengths (empty tails, single
frames, several batches), thread counts 1..48, three presets -- every finished stream must be the oracle's .flac.
`python3 tools/soak/soak_many.py [seconds] [coalesce]`"""
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
coalesce = len(sys.argv) > 2 and sys.argv[2] == "coalesce"
rng = np.random.Generator(np.random.PCG64(20261003))
t_end = time.time() + budget
rounds = streams_done = 0
cache = {}
while time.time() < t_end:
    preset = str(rng.choice(["fast", "default", "best"]))
    bps = int(rng.choice([16, 24]))
    block = 1152 if preset == "fast" else 4096
    n = int(rng.choice([3, 17, 63, 64, 65, 97, 130] + ([1, 300, 700] if coalesce else [])))
    threads = int(rng.choice([1, 2, 3, 7, 16, 48] + ([0] if coalesce else [])))
    bf = int(rng.choice([4, 16, 64]))
    lens = [int(rng.integers(1, 6 * bf * block // 4)) if rng.integers(4) else block * int(rng.integers(1, 9)) for _ in range(n)]
    streams = [synth_fast(int(rng.integers(1 << 20)) % 97 + 5000, 2, bps, ln) for ln in lens]
    opts = getattr(Options, preset)().batch_frames(bf * (16 if coalesce and rng.integers(2) else 1))
    if coalesce:
        opts = opts.pipeline_depth(int(rng.choice([1, 2, 6])))
    outs = BatchEncoder(opts, threads=threads, coalesce=coalesce).encode(streams, 48000, bps, 2)
    for s, o in zip(streams, outs):
        key = (preset, bps, s.size, int(s[:64].astype(np.int64).sum()), int(s[-64:].astype(np.int64).sum()))
        if key not in cache:
            rc, ref, _ = orc.encode_stream(orc.options(preset), 48000, bps, 2, s, total_known=True)
            assert rc == 0
            cache[key] = ref
        if o != cache[key]:
            ref = cache[key]
            first = next((i for i in range(min(len(o), len(ref))) if o[i] != ref[i]), min(len(o), len(ref)))
            idx = [k for k, (s2, o2) in enumerate(zip(streams, outs)) if s2 is s][0]
            print(f"MISMATCH preset {preset} bps {bps} streams {n} threads {threads} batch {opts._c.batch_frames} depth "
                  f"{opts._c.pipeline_depth} len {s.size} stream #{idx} of lens {lens}: {len(o)} bytes against {len(ref)}, first "
                  f"difference at byte {first} (MD5 field: bytes 26..41), round {rounds}")
            nbad = sum(1 for s2, o2 in zip(streams, outs) if o2 != cache.get((preset, bps, s2.size, int(s2[:64].astype(np.int64).sum()), int(s2[-64:].astype(np.int64).sum())), o2))
            print(f"  streams of this call known to differ: {nbad}")
            sys.exit(1)
    rounds += 1
    streams_done += n
print(f"many-stream soak ({'coalesced' if coalesce else 'a writer per stream'}): {rounds} calls, {streams_done} streams, thread counts 1..48, mismatches 0")
